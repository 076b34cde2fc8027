// mc_kernel instantiations for NSLOT = 4 (up to 256 clusters per site)
#include "mc_general.h"

int smolmc_launch_general_4(smolmc_handle *h, const KParams &kp, int replay) {
    return launch_general_nslot<4>(h, kp, replay);
}
