// mc_table_kernel<4, MM, EWM, false, BIAS = true>: TableFlip with an MCBias term on the single-class lean layout
#include "mc_lean.h"

int smolmc_launch_table_bias_4(smolmc_handle *h, const LeanParams &lp) {
    return launch_table_bias_nslot<4>(h, lp);
}
