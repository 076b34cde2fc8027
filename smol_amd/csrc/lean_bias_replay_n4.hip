// biased mc_lean_kernel replay instantiations for NSLOT = 4
#include "mc_lean.h"

int smolmc_launch_lean_bias_replay_4(smolmc_handle *h, const LeanParams &lp) { return launch_lean_bias_replay_nslot<4>(h, lp); }
