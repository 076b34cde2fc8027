// mc_table_kernel<2, MM, EWM, false, false, WLT = true>: Wang-Landau with TableFlip proposals on the single-class lean layout
#include "mc_lean_multi.h"

int smolmc_launch_table_wl_2(smolmc_handle *h, const LeanParams &lp) {
    return launch_table_wl_nslot<2>(h, lp);
}
