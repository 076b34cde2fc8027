// mc_lean_multi_kernel instantiations with an MCBias term, NSLOT = 8
#include "mc_lean_multi.h"

int smolmc_launch_multi_bias_8(smolmc_handle *h, const LeanParams &lp) {
    return launch_multi_bias_nslot<8>(h, lp);
}
