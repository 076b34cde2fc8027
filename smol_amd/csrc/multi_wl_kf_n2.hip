// mc_lean_multi_kernel Wang-Landau instantiations with several correlation functions per orbit (KFW), NSLOT = 2
#include "mc_lean_multi.h"

int smolmc_launch_multi_wl_kf_2(smolmc_handle *h, const LeanParams &lp) { return launch_multi_wl_nslot<2, false, 2>(h, lp); }
