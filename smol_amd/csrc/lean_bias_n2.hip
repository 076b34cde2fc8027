// biased (MCBias) mc_lean_kernel instantiations for NSLOT = 2
#include "mc_lean.h"

int smolmc_launch_lean_bias_2(smolmc_handle *h, const LeanParams &lp) {
    return launch_lean_bias_nslot<2>(h, lp);
}
