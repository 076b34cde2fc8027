// mc_table_multi_kernel<4, MM, EWM, false, false, BIAS = true>: TableFlip with an MCBias term on the multi-class lean layout
#include "mc_lean_multi.h"

int smolmc_launch_multi_table_bias_4(smolmc_handle *h, const LeanParams &lp) {
    return launch_table_multi_bias_nslot<4>(h, lp);
}
