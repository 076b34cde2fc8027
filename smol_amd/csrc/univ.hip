// mc_univ_kernel instantiations (the universal backstop kernel, mc_univ.h)
#include "mc_univ.h"

int smolmc_launch_univ(smolmc_handle *h, const UParams &up, int replay) {
    const int wpb = h->univ_wpb;
    const size_t lds = (size_t)up.lds_shared + (size_t)up.lds_per_wave * wpb;
    const bool table = up.K.step_type == SMOLMC_STEP_TABLE_FLIP, k1 = up.all_k1 != 0;
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) cus = 256;
    const bool dense = (long long)h->R > 2ll * 4 * cus; // more than two walkers per SIMD
    const int sel = (up.occ_lds ? 8 : 0) | (k1 ? 4 : 0) | (table ? 2 : 0) | (dense ? 1 : 0);
    void (*kern)(const UParams, const int) = up.dict_lds ? smolmc_univ_kernel_dict(sel) : univ_select<false>(sel);
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
