// mc_lean_multi_kernel Wang-Landau replay instantiations (host-provided proposals, smolmc_replay), NSLOT = 8
#include "mc_lean_multi.h"

int smolmc_launch_multi_wl_replay_8(smolmc_handle *h, const LeanParams &lp) { return launch_multi_wl_nslot<8, true>(h, lp); }
