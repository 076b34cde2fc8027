// mc_kernel instantiations for NSLOT = 2 (up to 128 clusters per site)
#include "mc_general.h"

int smolmc_launch_general_2(smolmc_handle *h, const KParams &kp, int replay) {
    return launch_general_nslot<2>(h, kp, replay);
}
