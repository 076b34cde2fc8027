// mc_table_kernel<2, MM, EWM, false, BIAS = true>: TableFlip with an MCBias term on the single-class lean layout
#include "mc_lean.h"

int smolmc_launch_table_bias_2(smolmc_handle *h, const LeanParams &lp) {
    return launch_table_bias_nslot<2>(h, lp);
}
