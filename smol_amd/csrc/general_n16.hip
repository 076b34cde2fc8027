// mc_kernel instantiations for NSLOT = 16 (up to 1024 clusters per site)
#include "mc_general.h"

int smolmc_launch_general_16(smolmc_handle *h, const KParams &kp, int replay) {
    return launch_general_nslot<16>(h, kp, replay);
}
