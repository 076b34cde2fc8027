// mc_lean_multi_kernel instantiations for NSLOT = 4
#include "mc_lean_multi.h"

int smolmc_launch_multi_4(smolmc_handle *h, const LeanParams &lp) {
    return launch_multi_nslot<4>(h, lp);
}
