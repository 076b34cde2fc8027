// biased (MCBias) mc_lean_kernel instantiations for NSLOT = 4
#include "mc_lean.h"

int smolmc_launch_lean_bias_4(smolmc_handle *h, const LeanParams &lp) {
    return launch_lean_bias_nslot<4>(h, lp);
}
