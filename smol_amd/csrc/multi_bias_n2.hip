// mc_lean_multi_kernel instantiations with an MCBias term, NSLOT = 2
#include "mc_lean_multi.h"

int smolmc_launch_multi_bias_2(smolmc_handle *h, const LeanParams &lp) {
    return launch_multi_bias_nslot<2>(h, lp);
}
