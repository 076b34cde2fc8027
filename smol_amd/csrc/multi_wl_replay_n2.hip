// mc_lean_multi_kernel Wang-Landau replay instantiations (host-provided proposals, smolmc_replay), NSLOT = 2
#include "mc_lean_multi.h"

int smolmc_launch_multi_wl_replay_2(smolmc_handle *h, const LeanParams &lp) { return launch_multi_wl_nslot<2, true>(h, lp); }
