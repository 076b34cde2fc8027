// mc_lean_kernel instantiations with several correlation functions per orbit (KF), NSLOT = 4
#include "mc_lean.h"

int smolmc_launch_lean_corr_4(smolmc_handle *h, const LeanParams &lp) {
    return launch_lean_corr_nslot<4>(h, lp);
}
