// mc_lean_multi_kernel Wang-Landau instantiations (several site classes / update_period > 1), NSLOT = 8
#include "mc_lean_multi.h"

int smolmc_launch_multi_wl_8(smolmc_handle *h, const LeanParams &lp) { return launch_multi_wl_nslot<8>(h, lp); }
