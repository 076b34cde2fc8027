// mc_kernel instantiations for NSLOT = 8 (up to 512 clusters per site)
#include "mc_general.h"

int smolmc_launch_general_8(smolmc_handle *h, const KParams &kp, int replay) {
    return launch_general_nslot<8>(h, kp, replay);
}
