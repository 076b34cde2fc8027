// mc_lean_kernel replay instantiations (host-provided proposals, smolmc_replay) for NSLOT = 2
#include "mc_lean.h"

int smolmc_launch_lean_replay_2(smolmc_handle *h, const LeanParams &lp) { return launch_lean_replay_nslot<2>(h, lp); }
