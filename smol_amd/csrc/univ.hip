// mc_univ_kernel instantiations (the universal backstop kernel, mc_univ.h)
#include "mc_univ.h"

int smolmc_launch_univ(smolmc_handle *h, const UParams &up, int replay) {
    const int wpb = h->univ_wpb;
    const size_t lds = (size_t)up.lds_per_wave * wpb;
    auto kern = up.occ_lds ? mc_univ_kernel<true> : mc_univ_kernel<false>;
    if (lds > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)((h->R + wpb - 1) / wpb);
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wpb), lds, h->stream, up, replay);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return 0;
}
