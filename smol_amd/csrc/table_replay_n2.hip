// mc_table_kernel replay instantiations (host-provided step records, smolmc_replay) for NSLOT = 2
#include "mc_lean.h"

int smolmc_launch_table_replay_2(smolmc_handle *h, const LeanParams &lp) { return launch_table_replay_nslot<2>(h, lp); }
