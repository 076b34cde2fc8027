// mc_table_multi_kernel replay instantiations for NSLOT = 4
#include "mc_lean_multi.h"

int smolmc_launch_multi_table_replay_4(smolmc_handle *h, const LeanParams &lp) { return launch_table_multi_replay_nslot<4>(h, lp); }
