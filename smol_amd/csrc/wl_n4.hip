// mc_wl_kernel instantiations (lean Wang-Landau) for NSLOT = 4
#include "mc_wl.h"

int smolmc_launch_wl_4(smolmc_handle *h, const LeanParams &lp) { return launch_wl_nslot<4>(h, lp); }
int smolmc_launch_wl_replay_4(smolmc_handle *h, const LeanParams &lp) { return launch_wl_replay_nslot<4>(h, lp); }
