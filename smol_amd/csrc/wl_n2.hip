// mc_wl_kernel instantiations (lean Wang-Landau) for NSLOT = 2
#include "mc_wl.h"

int smolmc_launch_wl_2(smolmc_handle *h, const LeanParams &lp) { return launch_wl_nslot<2>(h, lp); }
int smolmc_launch_wl_replay_2(smolmc_handle *h, const LeanParams &lp) { return launch_wl_replay_nslot<2>(h, lp); }
