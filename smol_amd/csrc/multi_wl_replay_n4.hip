// mc_lean_multi_kernel Wang-Landau replay instantiations (host-provided proposals, smolmc_replay), NSLOT = 4
#include "mc_lean_multi.h"

int smolmc_launch_multi_wl_replay_4(smolmc_handle *h, const LeanParams &lp) { return launch_multi_wl_nslot<4, true>(h, lp); }
