// mc_table_multi_kernel<2, MM, EWM, false, WLT = true>: Wang-Landau with TableFlip proposals on the multi-class lean layout
#include "mc_lean_multi.h"

int smolmc_launch_multi_table_wl_2(smolmc_handle *h, const LeanParams &lp) {
    return launch_table_multi_wl_nslot<2>(h, lp);
}
