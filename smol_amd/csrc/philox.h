// Philox4x32-10 (Salmon, Moraes, Dror, Shaw -- SC'11), host + device.
// The engine's counter-based random stream: key = walker seed (lo, hi),
// counter = (step_lo, step_hi, block, 0).  The CPU oracle
// (oracle/smolmc_oracle.c: orc_philox4x32) implements the identical function;
// tests/test_oracle_golden.py checks both against the Random123 known answers.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SMOLMC_HD __host__ __device__ __forceinline__
#else
#define SMOLMC_HD inline
#endif

struct philox_out {
    uint32_t w[4];
};

SMOLMC_HD void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0,
                            uint32_t k1) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32 on the device: the separate
    // v_mul_hi_u32 + v_mul_lo_u32 pair costs two quarter-rate instructions)
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0;
    c1 = lo1;
    c2 = n2;
    c3 = lo0;
}

SMOLMC_HD philox_out philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                   uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    philox_out o;
    o.w[0] = c0;
    o.w[1] = c1;
    o.w[2] = c2;
    o.w[3] = c3;
    return o;
}

// 53-bit uniform in [0,1) from two words (same construction as the oracle's u53)
SMOLMC_HD double philox_u53(uint32_t a, uint32_t b) {
    return (double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) * (1.0 / 9007199254740992.0);
}
