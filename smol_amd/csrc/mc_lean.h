// mc_lean.h -- lean Metropolis / Wang-Landau kernel, TableFlip kernel and their launch templates.
#pragma once
#include "smolmc_common.h"

// xor-butterfly inside 16-lane rows (DPP), then gfx950 permlane16/32 swaps: 22 VALU
// instructions, the total ends up in every lane.
template <int CTRL> __device__ __forceinline__ double dpp_xor_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, 0xf, 0xf, true);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_all(double v) {
    v = dpp_xor_add<0xB1>(v);  // quad_perm [1,0,3,2]
    v = dpp_xor_add<0x4E>(v);  // quad_perm [2,3,0,1]
    v = dpp_xor_add<0x141>(v); // row_half_mirror
    v = dpp_xor_add<0x140>(v); // row_mirror
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    return v;
}

// float32 sum over the wave for the accept pre-test, returned wave-uniform: 4 fused
// v_add_f32_dpp inside the 16-lane rows, then row_bcast:15 / row_bcast:31 in place (rows
// outside the row mask keep their value; only lane 63 needs the total) and v_readlane 63.
// The compiler does not fuse the partial-row-mask forms, hence the asm; the s_nops are the
// gfx9 DPP / readlane-SGPR wait states the hazard recogniser cannot see inside asm.
template <int CTRL> __device__ __forceinline__ float dpp_add_f32(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum_f32_uniform(float v) {
    v = dpp_add_f32<0xB1>(v);
    v = dpp_add_f32<0x4E>(v);
    v = dpp_add_f32<0x141>(v);
    v = dpp_add_f32<0x140>(v);
    float total;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_readlane_b32 %0, %1, 63\n\t"
                 "s_nop 1"
                 : "=s"(total), "+v"(v));
    return total;
}

// The same sum on the matrix pipe: two v_mfma_f64_16x16x4_f64 with B = ones
// (D[i][j] = sum_k A[i][k], lane l holds A[l & 15][l >> 4]; C/D: col = lane & 15,
// row = (lane >> 4) + 4 * reg) and three VALU adds in between.  Frees ~19 VALU issue
// slots per step but MEASURED SLOWER (9.87 ms vs 8.46 ms per 10^4 steps, same session):
// the two dependent f64 MFMAs lengthen the per-step dependency chain more than the VALU
// slots they free.  Kept behind -DSMOLMC_MFMA_REDUCE as a documented negative result.
typedef double smolmc_v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double wave_sum_mfma(double v) {
    const smolmc_v4d z = {0.0, 0.0, 0.0, 0.0};
    const smolmc_v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(v, 1.0, z, 0, 0, 0);
    const double t = (d[0] + d[1]) + (d[2] + d[3]);
    const smolmc_v4d d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(t, 1.0, z, 0, 0, 0);
    return d2[0];
}
#ifdef SMOLMC_MFMA_REDUCE
#define LEAN_WAVE_SUM wave_sum_mfma
#else
#define LEAN_WAVE_SUM wave_sum_all
#endif

// sum over the changeable sites k != s of q(k, occ_k) * G[s][k] (lane partial), eight sites
// per lane in flight so that the dependent latencies (index -> LDS species byte -> charge)
// of different sites overlap; the occupancy is read from LDS, so a tentatively applied
// first flip of a swap is seen without patching.
__device__ __forceinline__ double lean_ewald_partial(const LeanParams &P, const uint8_t *occ, int lane,
                                                     int s, int swa, int swm, int swb) {
    const double *g = P.ew_G + (size_t)s * P.ew_nact;
    const int W = P.ew_W, na = P.ew_nact;
    double out = 0;
    for (int j0 = lane; j0 < na; j0 += 64 * 8) {
        // branch-free: out-of-range lanes re-read the last element and are masked at the end
        // (conditional loads would put a full s_waitcnt between the eight loads)
        int k[8];
        double gk[8], q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int jj = min(j0 + 64 * u, na - 1);
            k[u] = P.ew_act[jj];
            gk[u] = g[jj];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            q[u] = P.ew_qs[(size_t)k[u] * W + (int)occ[lean_swz(k[u], swa, swm, swb)]];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            out = fma((j0 + 64 * u < na && k[u] != s) ? q[u] : 0.0, gk[u], out);
    }
    return out;
}

// potential-field update after an accepted flip of site s by charge dq: every other
// changeable site j gains dq * G[s][j] (G symmetric, row s is contiguous); the own entry
// is left as it was (phi excludes the self term): it is saved here and put back after the sweep.
// (PRE: the caller holds the E8 entries of the first 27 groups in registers, see field_sweep_gx_pre27)
template <int FOOT = 2, bool PRE = false>
__device__ __forceinline__ void field_apply(const LeanParams &P, double *phi, int lane, int s, double dq,
                                            const uint32_t (&e0)[27], const bool have_e0) {
    const int js = s - P.sbase;
    const double keep = phi[js];
    // (the compressed tables' pointers are re-read from the kernel arguments: see rare_params)
    const LeanParamsKernarg Q = rare_params();
    const unsigned char *gx = (const unsigned char *)Q->ew_gx;
    if (gx != nullptr) {
        const uint32_t *E8 = Q->ew_E8;
        const uint32_t s8[1] = {Q->ew_S8[js]};
        const double d[1] = {dq};
        if (PRE && have_e0) field_sweep_gx_pre27(phi, E8, gx, lane, P.ew_nact, s8, d, e0);
        else field_sweep_gx_multi<1, FOOT>(phi, E8, gx, lane, P.ew_nact, s8, d);
    } else {
        const double *g = P.ew_G + (size_t)s * P.ew_nact;
        field_sweep<false>(phi, g, g, lane, P.ew_nact, dq, 0.0);
    }
    phi[js] = keep; // (every lane stores the same value)
}

template <int FOOT = 2>
__device__ __forceinline__ void field_apply(const LeanParams &P, double *phi, int lane, int s, double dq) {
    const uint32_t none[27] = {};
    field_apply<FOOT, false>(P, phi, lane, s, dq, none, false);
}

// the flips of an accepted TableFlip step (lane f of vsite / vdq: site and charge change of flip f):
// with the translation-compressed kernel all of them in one pass over phi (up to four at a time),
// else one sweep per flip
// (inlined: out of line the kernel's register budget becomes the maximum over the call graph --
// 256 VGPRs and scratch -- instead of shrinking)
// (FEW: only the three- and four-flip sweeps are instantiated, shorter steps pad with zero charge
// changes -- for kernels whose code size matters more than the sweep of their rarer steps)
template <int FOOT = 1, bool FEW = false>
__device__ __forceinline__ void field_apply_flips(double *phi, int lane, int nfl, int vsite, double vdq) {
    const LeanParamsKernarg Q = rare_params();

    const unsigned char *gx = (const unsigned char *)Q->ew_gx;
    auto site_of = [&](int f) { return (int)rdlane((uint32_t)vsite, f); };
    auto dq_of = [&](int f) {
        return __hiloint2double((int)rdlane((uint32_t)__double2hiint(vdq), f), (int)rdlane((uint32_t)__double2loint(vdq), f));
    };
    const int sb = Q->sbase, na = Q->ew_nact;
    if (gx == nullptr) { // rows of the full site kernel, one sweep per flip
        const double *G = Q->ew_G;
        for (int f = 0; f < nfl; ++f) {
            const double dqf = dq_of(f);
            if (dqf == 0.0) continue;
            const double *g = G + (size_t)site_of(f) * na;
            field_sweep<false>(phi, g, g, lane, na, dqf, 0.0);
        }
        return;
    }
    const uint32_t *E8 = Q->ew_E8, *S8 = Q->ew_S8;
    for (int f0 = 0; f0 < nfl; f0 += 4) {
        const int n = min(4, nfl - f0);
        uint32_t s8[4];
        double dq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { // (flips beyond n repeat flip f0 with a zero charge change)
            const int f = k < n ? f0 + k : f0;
            s8[k] = S8[site_of(f) - sb];
            dq[k] = k < n ? dq_of(f) : 0.0;
        }
        if (FEW && n <= 3) {
            const uint32_t a[3] = {s8[0], s8[1], s8[2]};
            const double d[3] = {dq[0], dq[1], dq[2]};
            field_sweep_gx_multi<3, FOOT>(phi, E8, gx, lane, na, a, d);
        } else if (n == 1) {
            const uint32_t a[1] = {s8[0]};
            const double d[1] = {dq[0]};
            field_sweep_gx_multi<1, FOOT>(phi, E8, gx, lane, na, a, d);
        } else if (n == 2) {
            const uint32_t a[2] = {s8[0], s8[1]};
            const double d[2] = {dq[0], dq[1]};
            field_sweep_gx_multi<2, FOOT>(phi, E8, gx, lane, na, a, d);
        } else if (n == 3) {
            const uint32_t a[3] = {s8[0], s8[1], s8[2]};
            const double d[3] = {dq[0], dq[1], dq[2]};
            field_sweep_gx_multi<3, FOOT>(phi, E8, gx, lane, na, a, d);
        } else {
            field_sweep_gx_multi<4, FOOT>(phi, E8, gx, lane, na, s8, dq);
        }
    }
}

// both flips of a swap in one pass over phi (one read-modify-write per entry instead of two)
template <int FOOT = 1>
__device__ __forceinline__ void field_apply2(const LeanParams &P, double *phi, int lane, int s1, double dq1,
                                             int s2, double dq2) {
    const int j1 = s1 - P.sbase, j2 = s2 - P.sbase;
    const LeanParamsKernarg Q = rare_params();
    const unsigned char *gx = (const unsigned char *)Q->ew_gx;
    if (gx != nullptr) {
        // translation-compressed tables: G[s][s] = 0 there, so the multi-flip sweep leaves every entry right --
        // entry j1 gains dq1 * 0 + dq2 * G[s2][j1] -- and nothing is patched.  (Until round 5 the cross terms
        // were still read from the ROWS of the full site kernel here, two scalar loads from a 16 MB matrix per
        // accepted swap: an HBM round trip that every LDS wait behind it -- lgkmcnt is shared -- sat on.)
        const uint32_t *E8 = Q->ew_E8, *S8 = Q->ew_S8;
        const uint32_t s8[2] = {S8[j1], S8[j2]};
        const double d[2] = {dq1, dq2};
        field_sweep_gx_multi<2, FOOT>(phi, E8, gx, lane, P.ew_nact, s8, d);
        return;
    }
    const double *g1 = P.ew_G + (size_t)s1 * P.ew_nact, *g2 = P.ew_G + (size_t)s2 * P.ew_nact;
    // entry j1 must not see its own flip (but does see flip 2) and vice versa: both are patched
    // after the sweep from the values saved here
    const double keep1 = phi[j1], keep2 = phi[j2];
    const double c12 = g2[j1], c21 = g1[j2]; // cross terms G[s2][s1], G[s1][s2]
    field_sweep<true>(phi, g1, g2, lane, P.ew_nact, dq1, dq2);
    if (j1 != j2) {
        phi[j1] = fma(dq2, c12, keep1);
        phi[j2] = fma(dq1, c21, keep2);
    } else {
        phi[j1] = keep1;
    }
}

// -DSMOLMC_BOUNDS (make bounds -> libsmolmc_hip_bounds.so): trap when a gather address formed from an
// index row leaves the walker's occupancy in LDS.  The tables the rows are built from are checked on
// the host at create (validate_tables, engine.hip); this build checks the kernels' own packing /
// swizzle / unpack arithmetic on top of that.
__device__ __forceinline__ uint32_t bounded(uint32_t a, uint32_t lim) {
#ifdef SMOLMC_BOUNDS
    if (a >= lim) __builtin_trap();
#else
    (void)lim;
#endif
    return a;
}

// Index row of one site: ROW u16 entries per lane, fetched with raw buffer loads
// (resource descriptor + SGPR site offset + constant per-lane VGPR offset: no 64-bit VALU
// address arithmetic per step).  Words hold two u16 entries each.
template <int NW> struct RowWords { uint32_t w[NW]; };
template <int NW>
__device__ __forceinline__ RowWords<NW> load_row(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
    RowWords<NW> r;
    if constexpr (NW == 2) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
        r.w[0] = v[0]; r.w[1] = v[1];
    } else if constexpr (NW == 3) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b96(rs, voff, soff, 0);
        r.w[0] = v[0]; r.w[1] = v[1]; r.w[2] = v[2];
    } else if constexpr (NW == 4) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
        r.w[0] = v[0]; r.w[1] = v[1]; r.w[2] = v[2]; r.w[3] = v[3];
    } else if constexpr (NW == 6) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
        const auto u = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 16u, soff, 0);
        r.w[0] = v[0]; r.w[1] = v[1]; r.w[2] = v[2]; r.w[3] = v[3]; r.w[4] = u[0]; r.w[5] = u[1];
    } else {
        static_assert(NW % 4 == 0, "row width");
#pragma unroll
        for (int b = 0; b < NW / 4; ++b) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16u * b, soff, 0);
            r.w[4 * b] = v[0]; r.w[4 * b + 1] = v[1]; r.w[4 * b + 2] = v[2]; r.w[4 * b + 3] = v[3];
        }
    }
    return r;
}
template <int NW> __device__ __forceinline__ uint32_t row_entry(const RowWords<NW> &r, int q) {
    return (r.w[q >> 1] >> (16 * (q & 1))) & 0xffffu;
}
// one-wave-per-workgroup layout: 32-bit entries, used as LDS addresses as they are
template <bool WIDE, int NW> __device__ __forceinline__ uint32_t row_addr(const RowWords<NW> &r, int q) {
    if constexpr (WIDE) return r.w[q];
    else return row_entry<NW>(r, q);
}

// Wang-Landau flatness check (wanglandau.py:253-264), every check_period steps: kept out of
// line so that its temporaries do not add to the register pressure of the step loop.
// Returns the (possibly reduced) modification factor.
__device__ __noinline__ double wl_flatness_check(const double *wl_S, long long *wl_Hh, int L, double flat,
                                                 double div, double wl_m, int lane) {
    long cnt = 0;
    double sum = 0;
    for (int i = lane; i < L; i += 64)
        if (wl_S[i] > 0) { cnt++; sum += (double)wl_Hh[i]; }
    const double tcnt = wave_sum_all((double)cnt), tsum = wave_sum_all(sum);
    if (tcnt >= 2.0) {
        const double thr = flat * (tsum / tcnt);
        int bad = 0;
        for (int i = lane; i < L; i += 64)
            if (wl_S[i] > 0 && !((double)wl_Hh[i] > thr)) bad = 1;
        if (__ballot(bad) == 0ull) {
            for (int i = lane; i < L; i += 64) wl_Hh[i] = 0;
            wl_m = wl_m / div;
        }
    }
    return wl_m;
}

// flatness check (wanglandau.py:253-264) on the compact records: histogram = HBM base + LDS delta; on
// success the histogram is reset, the deltas move into the occurrences and m shrinks
__device__ __noinline__ double wl_multi_flatness_check(const double *S, uint32_t *cnt, double *occb, long long *hist_g,
                                                       long long *occ_g, int L, double flat, double div, double wl_m, int lane) {
    long n = 0;
    double sum = 0;
    for (int i = lane; i < L; i += 64)
        if (S[i] > 0) { n++; sum += (double)(hist_g[i] + (long long)cnt[i]); }
    const double tn = wave_sum_all((double)n), tsum = wave_sum_all(sum);
    if (tn >= 2.0) {
        const double thr = flat * (tsum / tn);
        int bad = 0;
        for (int i = lane; i < L; i += 64)
            if (S[i] > 0 && !((double)(hist_g[i] + (long long)cnt[i]) > thr)) bad = 1;
        if (__ballot(bad) == 0ull) {
            for (int i = lane; i < L; i += 64) {
                hist_g[i] = 0;
                occ_g[i] += (long long)cnt[i];
                if (occb) occb[i] += (double)cnt[i];
                cnt[i] = 0u;
            }
            wl_m = wl_m / div;
        }
    }
    return wl_m;
}

// LDS access by absolute address (SOLO layout: the occupancy starts at LDS address 0; going
// through the `extern __shared__` symbol would cost one "+ symbol" VALU add per access)
typedef __attribute__((address_space(3))) uint8_t lds_u8_t;
typedef __attribute__((address_space(3))) double lds_f64_t;
// (the integer -> LDS pointer casts only exist in the device pass: LDS pointers are 32 bits there)
#if defined(__HIP_DEVICE_COMPILE__)
#define SMOLMC_LDS_U8(a) (*(lds_u8_t *)(a))
#define SMOLMC_LDS_F64(a) (*(const lds_f64_t *)(a))
#define SMOLMC_LDS_F32(a) (*(const __attribute__((address_space(3))) float *)(a))
#else
#define SMOLMC_LDS_F32(a) (*(const float *)(uintptr_t)(a))
#define SMOLMC_LDS_U8(a) (*(uint8_t *)(uintptr_t)(a))
#define SMOLMC_LDS_F64(a) (*(const double *)(uintptr_t)(a))
#endif
template <bool ABS> __device__ __forceinline__ uint8_t occ_ld(const uint8_t *occ, uint32_t a) {
    if constexpr (ABS) return SMOLMC_LDS_U8(a);
    else return occ[a];
}
// a uniform value the compiler must treat as freshly defined here (no hoisting of what is derived from it)
__device__ __forceinline__ uint32_t opaque_u32(uint32_t v) {
    asm volatile("" : "+s"(v));
    return v;
}
template <bool ABS> __device__ __forceinline__ void occ_st(uint8_t *occ, uint32_t a, uint8_t v) {
    if constexpr (ABS) SMOLMC_LDS_U8(a) = v;
    else occ[a] = v;
}

// SOLO: one wave per workgroup with the walker's occupancy at LDS address 0 and the block-shared
// tables behind it.  The index rows then hold 32-bit LDS addresses that go into ds_read_u8 as
// they are -- no per-gather "wave base + unpack u16" instruction (8 VALU per swap step) -- and
// the site / candidate addresses need no base either.  Costs one copy of the tables per wave,
// so it is chosen when 16 waves per CU still fit (engine.hip).
// KF > 0: correlation features with up to KF correlation functions per orbit (evaluator.pyx:211-265
// for any number of species).  The accept decision is taken from ONE folded table per slot
// (E = sum_k coef_k ct_k, so a proposal costs what it costs in interaction mode); the KF
// correlation-function tables of the slot sit behind it in LDS and are read only on ACCEPTED
// steps, at the table index the decision already computed, into KF accumulators per slot.
// REPLAY: the proposals (site1, code1, site2, code2) and acceptance uniforms of every step come from
// the host in the reference's draw order (smolmc_replay) instead of the engine's
// Philox stream: the reference-order golden trajectories then drive this kernel's own evaluation --
// index rows, gathers, delta tables, swap antisymmetry, float32 pre-test, potential field -- and
// not only the general kernel's.  Separate instantiations (lean_replay_n*.hip).
template <int NSLOT, int MM, int STEP, bool HAS_MU, int EWM, bool WL, bool BIAS = false, bool SOLO = false,
          int KF = 0, int OCC = 0, bool REPLAY = false>
// OCC: waves per SIMD the register allocation is held to (0 = the compiler's choice, which is 4
// for the headline instantiation at 113 VGPRs).  OCC = 6 (80 VGPRs, a few spills) is launched
// when there are more walkers than 4 waves per SIMD can hold, see launch_lean_me.
__global__ void __attribute__((amdgpu_waves_per_eu(OCC ? OCC : 1, OCC ? OCC : 8)))
__launch_bounds__(256) mc_lean_kernel(const LeanParams P) {
    static_assert(KF == 0 || (!WL && !BIAS && !SOLO), "correlation-function tables: plain Metropolis layouts only");
    static_assert(!REPLAY || (!WL && OCC == 0), "replay: Metropolis variants (Wang-Landau: mc_wl_kernel)");
    constexpr int NACC = KF ? KF : 1;
    // EWM: 0 = no Ewald term, 1 = compact Ewald with per-proposal row sums, 2 = potential field in
    // LDS.  A template parameter (not the runtime flag ew_field): both variants' pointers and code
    // otherwise stay live across the step loop and the kernel spills SGPRs.
    constexpr bool HAS_EW = EWM != 0, ew_field = EWM == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((uint32_t)(uintptr_t)smem != 0u) __builtin_trap(); // (no static LDS in this kernel: absolute LDS addresses below)
    const int lane = threadIdx.x & 63;
    const int wave = SOLO ? 0 : threadIdx.x >> 6;
    const int nwaves = SOLO ? 1 : blockDim.x >> 6;
    const int r = uni(blockIdx.x * nwaves + wave);
    const size_t per_wave = (size_t)P.Nlds + 64 * 8 + (WL ? (size_t)P.wl.L * 24 : 0) +
                            ((HAS_EW && ew_field) ? 64 + (size_t)P.ew_nact * 8 : 0);
    double *s_dt = SOLO ? (double *)(smem + ((per_wave + 15) & ~(size_t)15)) : (double *)smem;
    const uint32_t dt_off = SOLO ? (uint32_t)((per_wave + 15) & ~(size_t)15) : 0u; // table base, folded into the slot offsets
    double *s_mu = s_dt + P.dt_len;               // 8 doubles
    double *s_q = s_mu + 8, *s_dg = s_mu + 16; // field mode: charge / diagonal term per code
    unsigned char *wbase = SOLO ? smem : (unsigned char *)(s_mu + 24) + (size_t)wave * per_wave;
    uint8_t *occ = wbase;                         // indexed by SWIZZLED site address
    // Metropolis: scratch for the feature reduction; Wang-Landau: the CURRENT features
    // (wanglandau.py:216-218 needs them every step for the per-bin running mean)
    double *s_feat = (double *)(wbase + P.Nlds);
    double *wl_S = s_feat + 64;                   // WL: entropy [L]
    long long *wl_Hh = (long long *)(wl_S + (WL ? P.wl.L : 0)); // WL: histogram [L]
    long long *wl_Oc = wl_Hh + (WL ? P.wl.L : 0);                // WL: occurrences [L]
    double *phi = (double *)(wbase + P.Nlds + 64 * 8 + 64);     // Ewald potential field [ew_nact]
    const int swa = P.swz_a, swm = P.swz_m, swb = P.swz_b;
    for (int i = threadIdx.x; i < P.dt_len; i += blockDim.x) s_dt[i] = P.dt[i];
#ifdef SMOLMC_EXP_F32TAB
    // EXPERIMENT (the round-2 review's "one bounded experiment" on the headline kernel): a float32
    // shadow of the delta tables for the decision, float64 entries read on accepted steps only.
    // Solo layout, swap steps, no Ewald / bias / correlation functions; the host adds the shadow's
    // LDS when SMOLMC_EXP_F32TAB is set in the environment (engine.hip).
    constexpr bool F32TAB = SOLO && STEP == SMOLMC_STEP_SWAP && !HAS_EW && !WL && !BIAS && KF == 0 && !REPLAY;
    const uint32_t sh_off = dt_off + (uint32_t)(P.dt_len + 24) * 8u; // shadow entry of the double at LDS address a: sh_off + (a - dt_off) / 2
    if (F32TAB) {
        float *s_dt32 = (float *)(smem + sh_off);
        for (int i = threadIdx.x; i < P.dt_len; i += blockDim.x) s_dt32[i] = (float)P.dt[i];
    }
#else
    constexpr bool F32TAB = false;
#endif
    if (HAS_MU && threadIdx.x < 8) s_mu[threadIdx.x] = threadIdx.x < P.ncodes ? P.mu_row[threadIdx.x] : 0.0;
    if (HAS_EW && ew_field && threadIdx.x < 8) {
        s_q[threadIdx.x] = P.ew_qrow[threadIdx.x];
        s_dg[threadIdx.x] = P.ew_dgrow[threadIdx.x];
    }
    const bool live = r < P.R;
    if (live) {
        // the swizzle only touches address bits >= 2: move whole dwords
        const uint32_t *src = (const uint32_t *)(P.occ + (size_t)r * P.Npad);
        for (int i = lane; i < P.Npad / 4; i += 64)
            *(uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb)) = src[i];
        s_feat[lane] = (WL && lane < P.F) ? P.features[(size_t)r * P.F + lane] : 0.0;
        if (HAS_EW && ew_field)
            for (int j = lane; j < P.ew_nact; j += 64) phi[j] = P.ew_phi[(size_t)r * P.ew_nact + j];
        if (WL)
            for (int i = lane; i < P.wl.L; i += 64) {
                wl_S[i] = P.wl.entropy[(size_t)r * P.wl.L + i];
                wl_Hh[i] = P.wl.hist[(size_t)r * P.wl.L + i];
                wl_Oc[i] = P.wl.occur[(size_t)r * P.wl.L + i];
            }
    }
    __syncthreads();
    if (!live) return;

    // single flips with the potential field and the compressed site kernel: the E8 entries of the
    // first 27 groups of field entries, lane constants of the launch (field_sweep_gx_pre27)
#ifdef SMOLMC_NO_EWALD_E8REG // A/B switch
    constexpr bool EPRE = false;
#else
    constexpr bool EPRE = HAS_EW && STEP == SMOLMC_STEP_FLIP && !REPLAY;
#endif
    uint32_t ereg[27] = {};
    bool epre_on = false;
    if (EPRE) {
        const LeanParamsKernarg Q = rare_params();
        const uint32_t *E8 = Q->ew_E8;
        epre_on = ew_field && Q->ew_gx != nullptr && (Q->ew_nact >> 6) >= 27;
        if (epre_on) {
#pragma unroll
            for (int u = 0; u < 27; ++u) ereg[u] = E8[64 * u + lane];
        }
    }

    // per-lane slot constants (registers for the whole launch)
    uint32_t doff8[NSLOT], st8[NSLOT][MM], sfeat[NSLOT];
    double wgt[NSLOT], acc[NSLOT], sfs[NSLOT];
    double accK[NSLOT][NACC]; // KF: per correlation function (acc then only carries the enthalpy)
    uint32_t kslot[NSLOT];    // KF: number of correlation functions of the slot's orbit
#pragma unroll
    for (int it = 0; it < NSLOT; ++it) {
        const LeanSlot sl = P.slots[it * 64 + lane];
        doff8[it] = sl.doff8 + dt_off;
        sfeat[it] = sl.feat;      // only used by the Wang-Landau variant
        sfs[it] = sl.live ? sl.fs : 0.0;
        kslot[it] = sl.live;
#pragma unroll
        for (int m = 0; m < MM; ++m) st8[it][m] = sl.stride8[m];
        wgt[it] = sl.w;
        acc[it] = 0.0;
#pragma unroll
        for (int k = 0; k < NACC; ++k) accK[it][k] = 0.0;
    }
    // Wang-Landau keeps the CURRENT feature vector in s_feat and adds every accepted step's
    // per-slot deltas to it with LDS atomics.  Lanes of one orbit hit the same cell (up to ~40
    // lanes per address on the headline model) and the LDS serialises them -- ~430 cycles per
    // ds_add_f64, a quarter of a step at one wave per SIMD, and every later LDS operation queues
    // behind it.  The 64 doubles of s_feat therefore hold WLK shadow copies of the vector
    // (WLK = min(8, 64 / F), packed back to back); lane l adds into copy l % WLK, which divides the
    // multiplicity per address by WLK; a reader sums the copies (wl_cur_feat).
    const int wl_stride = WL ? P.F : 64;                     // copies packed back to back
    const int wl_k = WL ? max(1, min(8, 63 / max(wl_stride, 1))) : 1; // (cell 63 stays zero: the address of "no copy")
    uint32_t sfeat_wl[NSLOT];
#pragma unroll
    for (int it = 0; it < NSLOT; ++it) sfeat_wl[it] = sfeat[it] + (uint32_t)((lane % wl_k) * wl_stride);
    // per-lane read addresses of the copies (doubles): copy k of the lane's feature, or the zero cell
    int wl_rd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) wl_rd[k] = (WL && k < wl_k && lane < wl_stride) ? lane + k * wl_stride : 63;
    auto wl_cur_feat = [&]() -> double { // lane f < F: current feature f (other lanes: unused)
        const int f = lane < wl_stride ? lane : 0;
        double t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = k < wl_k ? s_feat[f + (k < wl_k ? k : 0) * wl_stride] : 0.0;
        return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    };
    // feature read-out: sum over lanes and slots of fs * accumulator into s_feat[feature]
    auto reduce_features = [&](double *dst) {
#pragma unroll
        for (int it = 0; it < NSLOT; ++it) {
            if (KF == 0) {
                __hip_atomic_fetch_add(&dst[sfeat[it]], sfs[it] * acc[it], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WAVEFRONT);
            } else {
#pragma unroll
                for (int k = 0; k < NACC; ++k) {
                    const bool on = (uint32_t)k < kslot[it];
                    __hip_atomic_fetch_add(&dst[sfeat[it] + (on ? k : 0)], on ? sfs[it] * accK[it][k] : 0.0,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
        }
    };
#ifdef SMOLMC_EXP_CLOCK // experiment: shader clock during the launch (s_memtime vs the 100 MHz s_memrealtime)
    const long long ck0 = clock64(), wk0 = wall_clock64();
#endif
    double H = P.enthalpy[r];
    const double nbeta = WL ? 0.0 : -P.beta[r];
    double wl_m = WL ? P.wl.m[r] : 0.0;
    // bin coordinate of the current enthalpy, carried from the post-step of the previous step
    double wl_bq = WL ? floordiv_exact(H - P.wl.vmin, P.wl.bin) : 0.0;
    const double wl_inv_bin = WL ? 1.0 / P.wl.bin : 0.0;
    long long wl_counter = WL ? P.wl.counter[r] : 0;
    // counter modulo the check period, carried instead of recomputed (a 64-bit
    // modulo by a runtime divisor every step costs more than the step's arithmetic)
    long long wl_rem_check = WL ? (P.wl.check ? wl_counter % P.wl.check : 1) : 0; // (check period 0: never reaches it)
    unsigned long long step = P.nsteps[r];
    uint32_t nacc_add = 0; // accepted steps of this launch (32-bit counter; < 2^31 steps per launch)
    const uint32_t key0 = (uint32_t)P.seeds[r], key1 = (uint32_t)(P.seeds[r] >> 32);
    const uint32_t nact = (uint32_t)P.nact, nt8 = P.nt8, snt8 = P.snt8;
    const int sbase = P.sbase;
    double acc_mu = 0.0, acc_ew = 0.0;
    // MCBias (separate instantiations, lean_bias_n*.hip: even a never-taken runtime branch costs
    // the unbiased kernel 10 %): biased walkers always take the exact decision path
    const int btype = BIAS ? P.bias_type : 0;
    // running sums of the quadratic biases: the net charge (SquareChargeBias) / A_k . n - b_k of every hyperplane
    // (SquareHyperplaneBias, bias.py:290-366: up to SMOLMC_MAX_BIAS_ROWS rows, on the lean kernels since round 5)
    const int brows = (btype && btype != SMOLMC_BIAS_FUGACITY) ? P.bias_rows : 0;
    double bias_acc = 0.0, chg[SMOLMC_MAX_BIAS_ROWS];
#pragma unroll
    for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k) chg[k] = k < brows ? P.charge[(size_t)r * SMOLMC_MAX_BIAS_ROWS + k] : 0.0;
    // Metropolis without Ewald: the accept decision is pre-tested on a float32 wave sum of
    // the lane partials against thresholds widened by a rigorous error bound (P.fast_eps);
    // only the rare undecided step pays for the float64 reduction, so decisions are exactly
    // those of the float64 rule.  The enthalpy is then rebuilt from the per-slot feature
    // accumulators (sum_slots w * acc, the same sum in another order) and is
    // reduced when it is read (sample rows, end of launch).
    constexpr bool FAST = !WL && !HAS_EW;
    float thr_lo = 0.0f, thr_hi = 0.0f;
    const double inv_nbeta = FAST ? 1.0 / nbeta : 0.0;
    uint32_t nacc_before = 0; // accept counter before the current step: last_acc is read lazily
    int nsite = 0, naddr = 0;
    // trace at launch start; features of a sample = base + sum over lanes of fs * acc
    double *featp = P.features + (size_t)r * P.F;
    const double base_feat = lane < P.F ? featp[lane] : 0.0;
    // (no sampling: a countdown that cannot reach zero within the < 2^31 steps of a launch)
    uint32_t smp_countdown = P.smp.every ? (uint32_t)P.smp.every : 0xffffffffu;
    long long smp_index = 0;
    // random batch: lane l holds block (l & 3) of step batch_base + (l >> 2)
    uint32_t W0 = 0;
    int cand[4] = {0, 0, 0, 0}, canda[4] = {0, 0, 0, 0}; // candidate sites / their LDS addresses
    double vGc = 0.0; // Ewald field mode: prefetched cross terms of the first-round candidates
    double logu = 0.0; // log of the acceptance uniform of step (step & ~63) + lane
    unsigned long long batch64_base = ~0ull;
    unsigned long long batch_base = ~0ull;
    constexpr int ROW = NSLOT * MM; // entries per lane per site (u16, SOLO: u32)
    constexpr int NW = SOLO ? ROW : ROW / 2;          // dwords per lane per site
    constexpr uint32_t ENT = SOLO ? 4u : 2u;
    constexpr uint32_t SITE_BYTES = 64u * ROW * ENT;
    const __amdgpu_buffer_rsrc_t idx_rs = __builtin_amdgcn_make_buffer_rsrc(
        SOLO ? (void *)P.idx32 : (void *)P.idx, 0, 0x7fffffff, 0x00020000);
    const uint32_t lane_voff = (uint32_t)lane * (ROW * ENT);

    // software pipeline: the site of step k comes from W(k-1, 0, 1), so the index row of
    // the NEXT step is always known one step ahead and is fetched while this step runs.
    int s1, a1;
    RowWords<NW> row1;
    // replay: record i of this walker = (site1, code1, site2, code2), -1 = no flip; the site of an
    // empty step is any valid one (its "flip" keeps the species)
    uint32_t ridx = 0;
    double lu_rp = 0.0;
    int rp_bad = 0;
    auto rp_site = [&](const uint32_t i) -> int {
        const int v = uni(P.rp_steps[((size_t)r * (uint32_t)P.steps + i) * 4]);
        return v >= 0 ? v : sbase;
    };
    {
        if (REPLAY) {
            s1 = rp_site(0u);
        } else {
            const unsigned long long sp = step - 1ull;
            const uint32_t w = (uint32_t)uni((int)philox4x32_10((uint32_t)sp, (uint32_t)(sp >> 32), 0u, 0u,
                                                                key0, key1).w[1]);
            s1 = sbase + (int)__umulhi(w, nact);
        }
        a1 = lean_swz(s1, swa, swm, swb);
        row1 = load_row<NW>(idx_rs, lane_voff, (uint32_t)s1 * SITE_BYTES);
    }

    // Loop skeleton: the steps run in chunks that end at the next random-batch boundary, the
    // next sample row or the end of the launch, whichever comes first, so that the step loop
    // itself carries one down-counter and the two lane indices (a per-step batch test, sample
    // countdown, 64-bit step increment and launch counter cost ~10 SALU instructions per step).
    // Wang-Landau: the per-bin feature sums are the one global-memory update of a step.  Two things
    // about it matter at one wave per SIMD (measured 5.5 -> 4.x ms per 5000 steps):
    //  * vmcnt counts in order, so the atomic must be YOUNGER than the step's row fetches -- issued
    //    in the post-step it sits ahead of the next step's fetch of the partner's row, whose wait
    //    then covers the atomic's round trip to L2 as well.  A step therefore only notes its cell
    //    and value (wl_pend, wl_pend_val); the next step issues the atomic right behind its row
    //    fetch, the last one is flushed after the loop;
    //  * it must be issued on EVERY path and from ALL lanes (lanes >= F add 0.0 to the cells
    //    behind, the array is padded by 64): behind a branch or an exec mask the compiler's
    //    s_waitcnt insertion can no longer count it and falls back to vmcnt(0) at the row waits.
    //    The occurrence counts live in LDS beside the histogram for the same reason.
    double *wl_pend = WL ? P.wl.meanf + (size_t)r * P.wl.L * P.F : nullptr;
    // (the value: the shadow copies as they were READ in the post-step, summed only where the atomic
    // is issued in the next step -- the post-step then never waits for its own LDS atomics)
    double wl_pc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    auto wl_pend_sum = [&]() -> double {
        return ((wl_pc[0] + wl_pc[1]) + (wl_pc[2] + wl_pc[3])) + ((wl_pc[4] + wl_pc[5]) + (wl_pc[6] + wl_pc[7]));
    };
#ifdef SMOLMC_EXP_PHASES // experiment: shader cycles per phase of a step (walker 0 prints the averages)
    long long lph[5] = {0, 0, 0, 0, 0};
    long long lph_t = clock64();
#endif
    uint32_t steps_left = (uint32_t)P.steps; // the host splits launches at 2^30 steps
    while (steps_left != 0u) {
        // -------- random words (generated 16 steps at a time) --------
        const unsigned long long base = step & ~15ull;
        if (!REPLAY && base != batch_base) {
            batch_base = base;
            const unsigned long long st = base + (unsigned)(lane >> 2);
            const philox_out o = philox4x32_10((uint32_t)st, (uint32_t)(st >> 32), (uint32_t)(lane & 3),
                                               0u, key0, key1);
            W0 = o.w[0];
            // site of the step AFTER the lane's step and its swizzled LDS address, lane-parallel
            // (one v_readlane each per step instead of the scalar mulhi + swizzle chain)
            nsite = sbase + (int)__umulhi(o.w[1], nact);
            naddr = lean_swz(nsite, swa, swm, swb);
            // metropolis.py:46-48 compares the exponent with log(rng.random()): the float64 log
            // (and the float32 thresholds derived from it) of the acceptance uniforms of SIXTY-FOUR
            // steps at once, lane l <-> step (step & ~63) + l, from one more Philox block per lane
            // (block 0 of that step; the 16-step batches recompute the same words for the
            // proposals) -- a log per 16-step batch would use 16 of its 64 lanes
            if ((step & ~63ull) != batch64_base) {
                batch64_base = step & ~63ull;
                const unsigned long long s64 = batch64_base + (unsigned)lane;
                const philox_out a = philox4x32_10((uint32_t)s64, (uint32_t)(s64 >> 32), 0u, 0u, key0, key1);
                logu = log(philox_u53(a.w[2], a.w[3]));
                if (FAST) {
                    // accept <=> -beta dH > log u <=> dH < log(u) / -beta =: thr  (dH <= 0 always
                    // passes since thr >= 0); certain on either side of thr -+ eps
                    const double thr = logu * inv_nbeta; // (a rounding of the band centre: covered by eps)
                    const double eps = (F32TAB ? 2.0 : 1.0) * P.fast_eps + 1e-6 * fabs(thr); // (shadow tables: + sum|w| max|dt| 2^-23, a sixteenth of fast_eps)
                    thr_lo = P.fast_eps > 0.0 ? (float)(thr - eps) : -INFINITY;
                    thr_hi = P.fast_eps > 0.0 ? (float)(thr + eps) : INFINITY;
                }
            }
            if (STEP == SMOLMC_STEP_SWAP) {
                cand[0] = sbase + (int)__umulhi(o.w[0], nact);
                cand[1] = sbase + (int)__umulhi(o.w[1], nact);
                cand[2] = sbase + (int)__umulhi(o.w[2], nact);
                cand[3] = sbase + (int)__umulhi(o.w[3], nact);
#pragma unroll
                for (int j = 0; j < 4; ++j) canda[j] = lean_swz(cand[j], swa, swm, swb);
                if (HAS_EW && ew_field) {
                    // cross term G[candidate][site of the lane's step] of the first-round
                    // candidates of all 16 steps in one gather, issued a batch ahead (the site
                    // kernel lives in L2 / Infinity Cache; a dependent load per step would sit on
                    // the critical path).  The site of step k comes from the block of step k-1.
                    const int prev = __shfl(nsite, (lane & ~3) - 4);
                    const int site_l = lane < 4 ? s1 : prev;
                    vGc = P.ew_G[(size_t)cand[0] * P.ew_nact + (site_l - sbase)];
                }
            }
        }
        uint32_t chunk = 16u - (uint32_t)(step & 15ull);
        chunk = min(chunk, steps_left);
        chunk = min(chunk, smp_countdown);
        steps_left -= chunk;
        smp_countdown -= chunk;
        const unsigned long long chunk_end = step + chunk;
        int l4 = (int)(step & 15ull) * 4;
        int l64 = (int)(step & 63ull); // lane of this step's acceptance uniform / thresholds
        do {
        // site of the next step (depends only on random words; its index row is fetched below)
#ifndef SMOLMC_NO_SETPRIO
        __builtin_amdgcn_s_setprio(1); // wave priority rises through the step, see the decision below
#endif
#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); lph[0] += tn - lph_t; lph_t = tn; }
#endif
        int s1n, a1n;
        int rq1 = 0, rq2 = -1, rq3 = 0; // replay: code1, site2, code2 of this step's record
        bool rp_empty = false;
        if (REPLAY) {
            const int *rec = P.rp_steps + ((size_t)r * (uint32_t)P.steps + ridx) * 4;
            const int q0 = uni(rec[0]);
            rq1 = uni(rec[1]); rq2 = uni(rec[2]); rq3 = uni(rec[3]);
            rp_empty = q0 < 0;
            double u = uni_d(P.rp_u[(size_t)r * (uint32_t)P.steps + ridx]);
            if (u != u) u = 0.0; // NaN: the reference accepted without drawing a number
            lu_rp = log(u);
            s1n = ridx + 1u < (uint32_t)P.steps ? rp_site(ridx + 1u) : s1;
            a1n = lean_swz(s1n, swa, swm, swb);
            if (FAST) { // the thresholds of the float32 pre-test, per step here (see the 64-step batches)
                const double thr = lu_rp * inv_nbeta, eps = P.fast_eps + 1e-6 * fabs(thr);
                thr_lo = P.fast_eps > 0.0 ? (float)(thr - eps) : -INFINITY;
                thr_hi = P.fast_eps > 0.0 ? (float)(thr + eps) : INFINITY;
            }
        } else {
            s1n = (int)rdlane((uint32_t)nsite, l4);
            a1n = (int)rdlane((uint32_t)naddr, l4);
        }
        // LDS address of site 1 as a VGPR, made once per step (the compiler would re-make the
        // SGPR -> VGPR move in every block that touches the site)
        uint32_t va1 = (uint32_t)a1;
        asm volatile("" : "+v"(va1));
        const int o1 = uni((int)occ_ld<SOLO>(occ, va1));
        int nfl, s2, a2, n1, n2 = 0, o2 = 0; // (swap: s2 / a2 / o2 are set by every proposal outcome)
        if (STEP != SMOLMC_STEP_SWAP) { s2 = s1; a2 = a1; }
        int fb = -1; // swap: lane of a first-round candidate hit (prefetched Ewald cross term)
        if (REPLAY) {
            // the recorded proposal.  A swap kernel evaluates swaps in their antisymmetric form
            // (n2 == o1, o2 == n1 below): anything else in the record is flagged, not evaluated wrongly
            if (STEP == SMOLMC_STEP_FLIP) {
                nfl = rp_empty ? 0 : 1;
                n1 = rp_empty ? o1 : rq1;
                rp_bad |= (rq2 >= 0) ? 1 : 0;
            } else if (!rp_empty && rq2 >= 0) {
                nfl = 2;
                s2 = rq2;
                a2 = lean_swz(s2, swa, swm, swb);
                o2 = uni((int)occ_ld<SOLO>(occ, (uint32_t)a2));
                n2 = rq3;
                n1 = rq1;
                rp_bad |= (n2 != o1 || n1 != o2) ? 1 : 0;
                n2 = o1; n1 = o2;
            } else {
                nfl = 0; s2 = s1; a2 = a1; o2 = o1; n2 = o1; n1 = o1;
                rp_bad |= (!rp_empty || rq2 >= 0) ? 1 : 0;
            }
        } else if (STEP == SMOLMC_STEP_FLIP) {
            // Flip.propose_step (mcusher.py:154-170), default encoding 0..nc-1
            const uint32_t kk = __umulhi(rdlane(W0, l4 + 1), (uint32_t)(P.ncodes - 1));
            n1 = (int)kk + ((int)kk >= o1 ? 1 : 0);
            nfl = 1;
        } else {
            // Swap.propose_step (mcusher.py:176-200) by rejection over the candidate sequence.
            // Every exit sets the outputs itself and leaves through one branch (a shared
            // "found >= 0" epilogue costs the common first-candidate hit four compare-and-branch
            // pairs and a select chain).
            nfl = 2;
            n2 = o1;
            // candidate J of the first round: lanes l4+1 .. l4+3 test their word J
#define SMOLMC_CAND_MASK(J)                                                                        \
    const int v##J = (int)occ_ld<SOLO>(occ, (uint32_t)canda[J]);                                   \
    const unsigned long long m##J = __ballot(v##J != o1) & (0xEull << l4);
#define SMOLMC_CAND_TAKE(J)                                                                        \
    {                                                                                              \
        const int b = __ffsll((long long)m##J) - 1;                                                \
        s2 = (int)rdlane((uint32_t)cand[J], b);                                                    \
        a2 = (int)rdlane((uint32_t)canda[J], b);                                                   \
        o2 = (int)rdlane((uint32_t)v##J, b);                                                       \
        if (J == 0) fb = b;                                                                        \
    }
            SMOLMC_CAND_MASK(0)
            if (m0) SMOLMC_CAND_TAKE(0)
            else {
                SMOLMC_CAND_MASK(1)
                if (m1) SMOLMC_CAND_TAKE(1)
                else {
                    SMOLMC_CAND_MASK(2)
                    if (m2) SMOLMC_CAND_TAKE(2)
                    else {
                        SMOLMC_CAND_MASK(3)
                        if (m3) SMOLMC_CAND_TAKE(3)
                        else {
                            bool hit = false;
                            const unsigned long long cur = chunk_end - chunk; // this step (rare path)
                            for (uint32_t q = 0;; ++q) {
                                const philox_out o = philox4x32_10((uint32_t)cur, (uint32_t)(cur >> 32),
                                                                   4u + 64u * q + (uint32_t)lane, 0u, key0, key1);
                                int selsite = -1, selv = 0;
#pragma unroll
                                for (int j = 3; j >= 0; --j) {
                                    const int cs = sbase + (int)__umulhi(o.w[j], nact);
                                    const int v = (int)occ[lean_swz(cs, swa, swm, swb)];
                                    if (v != o1) { selsite = cs; selv = v; }
                                }
                                const unsigned long long m = __ballot(selsite >= 0);
                                if (m) {
                                    const int b = __ffsll((long long)m) - 1;
                                    s2 = (int)rdlane((uint32_t)selsite, b);
                                    a2 = lean_swz(s2, swa, swm, swb);
                                    o2 = (int)rdlane((uint32_t)selv, b);
                                    hit = true;
                                    break;
                                }
                                if ((q & 63u) == 0) { // swap_options.size == 0 -> empty step
                                    int any = 0;
                                    for (uint32_t a = lane; a < nact; a += 64)
                                        any |= ((int)occ[lean_swz(sbase + (int)a, swa, swm, swb)] != o1);
                                    if (__ballot(any) == 0ull) break;
                                }
                            }
                            if (!hit) { nfl = 0; s2 = s1; a2 = a1; o2 = o1; } // empty step: no-op 'flips'
                        }
                    }
                }
            }
#undef SMOLMC_CAND_MASK
#undef SMOLMC_CAND_TAKE
            n1 = o2;
        }

        // data-dependent row of site 2: issued before flip 1 is evaluated (s2 == s1 for the
        // rare empty step, the loaded row is then unused)
#ifndef SMOLMC_NO_SETPRIO
        __builtin_amdgcn_s_setprio(2);
#endif
        RowWords<NW> row2 = row1;
        if (STEP == SMOLMC_STEP_SWAP) {
#ifdef SMOLMC_EXP_ROW2 // timing experiment only: row of an early-known site (wrong results)
            row2 = load_row<NW>(idx_rs, lane_voff, (uint32_t)s1n * SITE_BYTES);
#else
            row2 = load_row<NW>(idx_rs, lane_voff, (uint32_t)s2 * SITE_BYTES);
#endif
        }
        if (WL) {
            unsafeAtomicAdd(wl_pend + lane, wl_pend_sum());
        }

#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); lph[1] += tn - lph_t; lph_t = tn; }
#endif
        // -------- enthalpy delta ---------------------------------------------------
        // Swap: the second flip is the reverse species change of the first
        // (n2 == o1, o2 == n1), and the delta tables are antisymmetric in (old, new), so both
        // flips read the SAME (old, new) block -- its offset is added to the slot offsets once --
        // and the step's delta per slot is the difference of the two reads: one table offset
        // add, one subtraction and one FMA per slot instead of two adds and two FMAs
        // (SMOLMC_NO_SWAP_DIFF: the flip-by-flip form, kept for A/B runs).
#ifdef SMOLMC_NO_SWAP_DIFF
        constexpr bool DIFF = false;
#else
        constexpr bool DIFF = STEP == SMOLMC_STEP_SWAP;
#endif
        double e = 0.0, d1[NSLOT], d2[NSLOT];
#ifdef SMOLMC_EXP_F32TAB
        float t1f[NSLOT], ef32 = 0.0f;
#endif
        uint32_t dp[NSLOT];
        uint32_t ad1[NSLOT], ad2[NSLOT]; // KF: LDS addresses of the two decision reads
        {
            const uint32_t pair1 = (uint32_t)o1 * snt8 + (uint32_t)n1 * nt8; // uniform
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) {
                uint32_t a = doff8[it];
                if (DIFF) a = dp[it] = doff8[it] + pair1;
#pragma unroll
                for (int m = 0; m < MM; ++m) a += __umul24(st8[it][m], (uint32_t)occ_ld<SOLO>(occ, bounded(row_addr<SOLO, NW>(row1, it * MM + m), (uint32_t)P.Nlds)));
                if (DIFF && F32TAB) {
#ifdef SMOLMC_EXP_F32TAB
                    ad1[it] = a;
                    t1f[it] = SMOLMC_LDS_F32(sh_off + ((a - dt_off) >> 1));
#endif
                } else if (DIFF) {
                    d1[it] = SMOLMC_LDS_F64(a);
                    if (KF) ad1[it] = a;
                } else {
                    d1[it] = SMOLMC_LDS_F64(a + pair1); // (absolute LDS address: a includes the table base)
                    if (KF) ad1[it] = a + pair1;
                    e = fma(wgt[it], d1[it], e);
                }
            }
        }
        // index row of the NEXT step's site, straight into row1: its last use (the gathers above)
        // has been issued, so no second register set and no copy at the end of the step
        row1 = load_row<NW>(idx_rs, lane_voff, (uint32_t)s1n * SITE_BYTES);
        double ew_part = 0.0, ew_uni = 0.0; // lane-partial / uniform parts of the Ewald delta
        double dq1 = 0.0, dq2 = 0.0;
        if (HAS_EW) {
            if (ew_field) { // no global loads: charges / diagonal terms per code from LDS
                dq1 = s_q[n1] - s_q[o1];
                ew_uni = 2.0 * dq1 * phi[s1 - sbase] + (s_dg[n1] - s_dg[o1]);
            } else {
                const int W = P.ew_W;
                dq1 = P.ew_qs[(size_t)s1 * W + n1] - P.ew_qs[(size_t)s1 * W + o1];
                ew_part = 2.0 * dq1 * lean_ewald_partial(P, occ, lane, s1, swa, swm, swb);
                ew_uni = 2.0 * dq1 * P.ew_frozen[s1] +
                         (P.ew_dg[(size_t)s1 * W + n1] - P.ew_dg[(size_t)s1 * W + o1]);
            }
        }
        if (STEP == SMOLMC_STEP_SWAP) {
            // the second flip sees the first (expansion.py:217-229): apply it tentatively in
            // LDS (undone below on rejection) instead of patching every gathered value
#ifndef SMOLMC_EXP_NOTENT // timing experiment only when defined (wrong results)
            occ_st<SOLO>(occ, va1, (uint8_t)n1); // every lane stores the same byte: no exec juggling
#endif
            const uint32_t pair2 = (uint32_t)o2 * snt8 + (uint32_t)n2 * nt8;
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) {
                uint32_t a = DIFF ? dp[it] : doff8[it];
#pragma unroll
                for (int m = 0; m < MM; ++m) a += __umul24(st8[it][m], (uint32_t)occ_ld<SOLO>(occ, bounded(row_addr<SOLO, NW>(row2, it * MM + m), (uint32_t)P.Nlds)));
                if (DIFF && F32TAB) {
#ifdef SMOLMC_EXP_F32TAB
                    ad2[it] = a;
                    ef32 = fmaf((float)wgt[it], t1f[it] - SMOLMC_LDS_F32(sh_off + ((a - dt_off) >> 1)), ef32);
#endif
                } else if (DIFF) {
                    d1[it] -= SMOLMC_LDS_F64(a); // D[(o2,n2)] = -D[(o1,n1)]: the step's delta of this slot
                    if (KF) ad2[it] = a;
                    e = fma(wgt[it], d1[it], e);
                } else {
                    d2[it] = SMOLMC_LDS_F64(a + pair2);
                    if (KF) ad2[it] = a + pair2;
                    e = fma(wgt[it], d2[it], e);
                }
            }
            if (HAS_EW) {
                if (ew_field) { // the second flip sees the first through the cross term
                    dq2 = s_q[n2] - s_q[o2];
                    const double cross =
                        fb >= 0 ? __hiloint2double((int)rdlane((uint32_t)__double2hiint(vGc), fb),
                                                   (int)rdlane((uint32_t)__double2loint(vGc), fb))
                                : P.ew_G[(size_t)s2 * P.ew_nact + (s1 - sbase)];
                    ew_uni += 2.0 * dq2 * (phi[s2 - sbase] + dq1 * cross) + (s_dg[n2] - s_dg[o2]);
                } else {
                    const int W = P.ew_W;
                    dq2 = P.ew_qs[(size_t)s2 * W + n2] - P.ew_qs[(size_t)s2 * W + o2];
                    ew_part += 2.0 * dq2 * lean_ewald_partial(P, occ, lane, s2, swa, swm, swb);
                    ew_uni += 2.0 * dq2 * P.ew_frozen[s2] +
                              (P.ew_dg[(size_t)s2 * W + n2] - P.ew_dg[(size_t)s2 * W + o2]);
                }
            }
        }
#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); lph[2] += tn - lph_t; lph_t = tn; }
#endif
        double dMu = 0.0;
        if (HAS_MU && nfl >= 1) {
            dMu = s_mu[n1] - s_mu[o1];
            if (nfl == 2) dMu += s_mu[n2] - s_mu[o2];
        }
        double dH = 0.0, dEw = 0.0;
        double wl_nbq = 0.0; // WL: bin coordinate of the proposed enthalpy
        // compute_bias_change against the original occupancy (kernel/base.py:307-311; bias.py)
        double dB = 0.0, dQ[SMOLMC_MAX_BIAS_ROWS] = {0.0, 0.0, 0.0, 0.0};
        if (btype && nfl >= 1) {
            if (btype == SMOLMC_BIAS_FUGACITY) {
                dB = P.bias_pair[o1 * 8 + n1];
                if (nfl == 2) dB += P.bias_pair[o2 * 8 + n2];
            } else {
                double sq_new = 0.0, sq_old = 0.0;
#pragma unroll
                for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k)
                    if (k < brows) {
                        const double *bp = P.bias_pair + k * P.bias_row_stride;
                        double x = bp[o1 * 8 + n1];
                        if (nfl == 2) x += bp[o2 * 8 + n2];
                        dQ[k] = x;
                        const double cn = chg[k] + x;
                        sq_old += chg[k] * chg[k];
                        sq_new += cn * cn;
                    }
                dB = -P.bias_pen * sq_new - (-P.bias_pen * sq_old);
            }
        }
#ifdef SMOLMC_EXP_F32TAB
        bool have_d1 = false;
#endif
        // exact float64 decision (metropolis.py:31-49 / wanglandau.py:186-202)
        auto exact_decision = [&]() -> bool {
#ifdef SMOLMC_EXP_F32TAB
            if (F32TAB) { // the float64 deltas, now that they are needed
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) {
                    d1[it] = SMOLMC_LDS_F64(ad1[it]) - SMOLMC_LDS_F64(ad2[it]);
                    e = fma(wgt[it], d1[it], e);
                }
                have_d1 = true;
            }
#endif
            dH = LEAN_WAVE_SUM(e);
            if (HAS_EW) {
                dEw = (ew_field ? 0.0 : LEAN_WAVE_SUM(ew_part)) + ew_uni;
                dH += P.ew_coef * dEw;
            }
            if (HAS_MU) dH -= dMu;
            // -------- accept (metropolis.py:31-49) --------
            const double lu = REPLAY ? lu_rp
                                     : __hiloint2double((int)rdlane((uint32_t)__double2hiint(logu), l64),
                                                        (int)rdlane((uint32_t)__double2loint(logu), l64));
            // (the ballots make the wave-uniform decision visibly uniform to the compiler:
            // scalar branch, uniform counters in SGPRs)
            if (!WL) {
                const double exponent = nbeta * dH + 0.0 + dB; // metropolis.py:41-44
                return __ballot((exponent >= 0.0) || (exponent > lu)) != 0ull;
            } else {
                // WangLandau._accept_step (wanglandau.py:186-202)
                const double new_h = H + dH;
                if (new_h < P.wl.vmin || new_h >= P.wl.vmax) {
                    return false;
                } else {
                    const int b = (int)wl_bq;
                    wl_nbq = floordiv_exact_inv(new_h - P.wl.vmin, P.wl.bin, wl_inv_bin);
                    const int nb = (int)wl_nbq;
                    const double exponent = wl_S[b] - wl_S[nb] + 0.0;
                    return __ballot((exponent >= 0.0) || (exponent > lu)) != 0ull;
                }
            }
        };
        // -------- update (kernel/base.py:327-343) --------
        // SELACC: the per-slot accumulators are updated once, after the decision paths have
        // merged, as acc = fma(sel, delta, acc) with the wave-uniform sel = 1.0 / 0.0 -- updating
        // them inside the accept branch makes the register allocator keep two copies and shuffle
        // them with v_mov_b64 on every path (~5 VALU per step)
#ifdef SMOLMC_NO_SELACC
        constexpr bool SELACC = false;
#else
        constexpr bool SELACC = DIFF && FAST && !BIAS && !F32TAB; // (measured: no gain for the flip / Ewald variants)
#endif
        uint32_t sel_hi = 0u; // high word of sel
        auto on_accept = [&]() {
#ifdef SMOLMC_EXP_F32TAB
            if (F32TAB && !have_d1) {
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) d1[it] = SMOLMC_LDS_F64(ad1[it]) - SMOLMC_LDS_F64(ad2[it]);
            }
#endif
            if (KF) {
                // the K correlation-function tables of each slot (global memory, L2-resident),
                // read at the index the decision already computed: byte ad - doff8 inside table k
                // of the slot's group, which starts at doff8 * MAX_KF (LeanParams::dtk); slots
                // with fewer functions add a zero
                const unsigned char *kb = (const unsigned char *)P.dtk;
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) {
                    const uint32_t g1 = doff8[it] * (uint32_t)(SMOLMC_LEAN_MAX_KF - 1) + ad1[it];
                    const uint32_t g2 = doff8[it] * (uint32_t)(SMOLMC_LEAN_MAX_KF - 1) + ad2[it];
#pragma unroll
                    for (int k = 0; k < NACC; ++k) {
                        const bool on = (uint32_t)k < kslot[it];
                        const uint32_t off = (uint32_t)k * P.ktab8;
                        double v = on ? *(const double *)(kb + (g1 + off)) : 0.0;
                        if (STEP == SMOLMC_STEP_SWAP) {
                            const double v2 = on ? *(const double *)(kb + (g2 + off)) : 0.0;
                            v = DIFF ? v - v2 : v + v2;
                        }
                        accK[it][k] += v;
                    }
                }
            }
            if (SELACC) {
                sel_hi = 0x3ff00000u;
            } else if (!WL) {
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) acc[it] += d1[it];
                if (STEP == SMOLMC_STEP_SWAP && !DIFF) {
#pragma unroll
                    for (int it = 0; it < NSLOT; ++it) acc[it] += d2[it];
                }
            } else {
                // _do_accept_step (wanglandau.py:204-220): current features += delta features
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) {
                    const double dd = (STEP == SMOLMC_STEP_SWAP && !DIFF) ? d1[it] + d2[it] : d1[it];
                    __hip_atomic_fetch_add(&s_feat[sfeat_wl[it]], sfs[it] * dd, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            if (STEP == SMOLMC_STEP_FLIP) occ_st<SOLO>(occ, va1, (uint8_t)n1);
#ifdef SMOLMC_EXP_NOTENT
            if (STEP == SMOLMC_STEP_SWAP) occ_st<SOLO>(occ, (uint32_t)a1, (uint8_t)n1);
#endif
            if (STEP == SMOLMC_STEP_SWAP) occ_st<SOLO>(occ, (uint32_t)a2, (uint8_t)n2); // (n2 == o1 == occ[a1] when empty)
            if (HAS_EW && ew_field) {
                if (STEP == SMOLMC_STEP_SWAP) {
                    if (dq1 != 0.0 || dq2 != 0.0) field_apply2(P, phi, lane, s1, dq1, s2, dq2);
                } else if (dq1 != 0.0) {
#ifndef SMOLMC_EXP_NOFIELD // timing experiment only when defined (wrong results)
                    field_apply<2, EPRE>(P, phi, lane, s1, dq1, ereg, epre_on);
#endif
                }
            }
            acc_mu += dMu;
            acc_ew += dEw;
            bias_acc += dB;
#pragma unroll
            for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k) chg[k] += dQ[k];
            if (!FAST) H += dH;
            if (!SELACC) nacc_add++;
        };
        auto on_reject = [&]() {
            if (STEP == SMOLMC_STEP_SWAP) {
#ifndef SMOLMC_EXP_NOTENT
            occ_st<SOLO>(occ, va1, (uint8_t)o1); // undo the tentative first flip
#endif
            }
        };
        // Each outcome of the float32 pre-test runs its update directly (a merged
        // "decided / accepted" pair of flags costs the common path extra compare-and-branch steps).
        nacc_before = nacc_add;
        bool accepted = false; // (read after this point by the Wang-Landau post-step only)
        // Wave priority rises through the step: 0 in the loop skeleton (random batches), 1 during
        // the proposal, 2 from the row fetch / gathers, 3 for the decision chain (reduction, accept
        // test, updates) -- the wave furthest into its step issues first and reaches its next
        // memory instructions sooner.  Headline: 5.17 -> 4.66 ms (two levels: 5.03; three: 4.87 /
        // 4.73 depending on where level 1 starts; one constant level or bare scheduling barriers at
        // the same places: slower than nothing; the decision at the LOWEST level: 5.10); 0-3 % on
        // the other variants, -2 % at one wave per SIMD.
#ifndef SMOLMC_NO_SETPRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        if (FAST && !BIAS) {
#ifdef SMOLMC_EXP_F32TAB
            const float ef = F32TAB ? ((HAS_MU && lane == 0) ? ef32 - (float)dMu : ef32) : (float)((HAS_MU && lane == 0) ? e - dMu : e);
#else
            const float ef = (float)((HAS_MU && lane == 0) ? e - dMu : e);
#endif
            const float S = wave_sum_f32_uniform(ef);
            const unsigned long long bit = 1ull << (REPLAY ? 0 : l64); // (replay: the thresholds are uniform)
            const bool ca = (__ballot(S < thr_lo) & bit) != 0ull;
            const bool cr = (__ballot(S > thr_hi) & bit) != 0ull;
            if (ca) on_accept();
            else if (cr) on_reject();
            else if (exact_decision()) on_accept();
            else on_reject();
        } else {
            accepted = exact_decision();
            if (accepted) on_accept();
            else on_reject();
        }
        if (SELACC) {
            const uint32_t sh = (uint32_t)uni((int)sel_hi);
            const double sel = __hiloint2double((int)sh, 0);
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) acc[it] = fma(sel, d1[it], acc[it]);
            nacc_add += sh >> 29; // 0x3ff00000 >> 29 == 1
        }
        s1 = s1n;
        a1 = a1n;
#ifndef SMOLMC_NO_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if (REPLAY) { // accept flag and running enthalpy of every step (what smolmc_replay returns)
            double lane_e = 0.0;
            if (FAST) {
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) lane_e = fma(wgt[it], acc[it], lane_e);
            }
            const double Hnow = FAST ? H + (wave_sum_all(lane_e) - acc_mu) : H;
            if (lane == 0) {
                const size_t k = (size_t)r * (uint32_t)P.steps + ridx;
                P.rp_acc[k] = (uint8_t)(nacc_add != nacc_before);
                P.rp_H[k] = Hnow;
            }
            ridx++;
        }

#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); lph[3] += tn - lph_t; lph_t = tn; }
#endif
        if (WL) {
            // WangLandau._do_post_step (wanglandau.py:222-266)
            // the bin only moves on accepted steps, to the one computed by the accept test
            const double bq = accepted ? wl_nbq : wl_bq;
            wl_bq = bq;
            // the lean kernel is only dispatched for update_period == 1 (everything else takes
            // mc_kernel): occurrences and per-bin feature SUMS by fire-and-forget atomics -- the
            // running mean of wanglandau.py:235-239 is sum / occurrences, formed when read
            if (bq >= 0.0 && bq < (double)P.wl.L) {
                const int b = (int)bq;
                wl_counter++;
                if (++wl_rem_check == P.wl.check) wl_rem_check = 0;
                wl_pend = P.wl.meanf + ((size_t)r * P.wl.L + b) * P.F;
#pragma unroll
                for (int k = 0; k < 8; ++k) wl_pc[k] = s_feat[wl_rd[k]];
                if (lane == 0) {
                    // LDS atomics without return value: a read-modify-write would put one more
                    // LDS round trip on the step's dependency chain (one wave per SIMD here)
                    __hip_atomic_fetch_add(&wl_S[b], wl_m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add(&wl_Hh[b], 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add(&wl_Oc[b], 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            if (wl_rem_check == 0) {
                const LeanParamsKernarg Q = rare_params();
                wl_m = wl_flatness_check(wl_S, wl_Hh, Q->wl.L, Q->wl.flat, Q->wl.div, wl_m, lane);
            }
        }

#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); lph[4] += tn - lph_t; lph_t = tn; }
#endif
        l4 += 4;
        l64 += 1;
        } while (--chunk != 0u);
        step = chunk_end;

        if (smp_countdown == 0) { // record one thinned sample of this walker
            // (sampling parameters re-read from the kernel arguments: see rare_params)
            const LeanParamsKernarg Q = rare_params();
            const int qF = Q->F, qFce = Q->Fce;
            double *const q_feat = Q->smp.feat;
            smp_countdown = (uint32_t)Q->smp.every;
            const size_t row = (size_t)smp_index * Q->R + r;
            smp_index++;
            if (WL) {
                const double fcur = wl_cur_feat();
                if (lane < qF) q_feat[row * qF + lane] = fcur;
            } else {
                s_feat[lane] = 0.0;
                reduce_features(s_feat);
                if (lane < qFce) q_feat[row * qF + lane] = base_feat + s_feat[lane];
            }
            if (!WL && HAS_EW && lane == qFce) q_feat[row * qF + lane] = base_feat + acc_ew;
            if (!WL && HAS_MU && lane == qFce + (HAS_EW ? 1 : 0))
                q_feat[row * qF + lane] = base_feat + acc_mu;
            double lane_e = 0.0;
            if (FAST) {
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) lane_e = fma(wgt[it], acc[it], lane_e);
            }
            const double Hnow = FAST ? H + (wave_sum_all(lane_e) - acc_mu) : H;
            if (lane == 0) {
                Q->smp.H[row] = Hnow;
                Q->smp.acc[row] = (uint8_t)(nacc_add != nacc_before);
                if (BIAS && Q->smp_bias_off) (Q->smp.H + Q->smp_bias_off)[row] = Q->bias[r] + bias_acc; // trace.bias (kernel/base.py:307-311)
            }
            if (Q->smp.occ) {
                const int qNpad = Q->Npad;
                uint32_t *dst = (uint32_t *)(Q->smp.occ + row * qNpad);
                for (int i = lane; i < qNpad / 4; i += 64)
                    dst[i] = *(const uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb));
            }
        }
    }

#ifdef SMOLMC_EXP_CLOCK
    if ((r == 0 || r == 2049) && lane == 0) {
        const long long dc = clock64() - ck0, dw = wall_clock64() - wk0;
        printf("walker %d: %lld shader cycles in %lld ticks of 10 ns -> %.1f MHz, %.1f cycles per step\n", r, dc, dw,
               (double)dc / (double)dw * 100.0, (double)dc / (double)P.steps);
    }
#endif
#ifdef SMOLMC_EXP_PHASES
    if (r == 0 && lane == 0)
        printf("lean phases (cycles per step): skeleton %.0f | proposal %.0f | gathers+tables %.0f | decision+update %.0f | post-step %.0f\n",
               (double)lph[0] / (double)P.steps, (double)lph[1] / (double)P.steps, (double)lph[2] / (double)P.steps,
               (double)lph[3] / (double)P.steps, (double)lph[4] / (double)P.steps);
#endif
    // ---- write back ---------------------------------------------------------------
    if (HAS_EW && ew_field)
        for (int j = lane; j < P.ew_nact; j += 64) P.ew_phi[(size_t)r * P.ew_nact + j] = phi[j];
    {
        uint32_t *dst = (uint32_t *)(P.occ + (size_t)r * P.Npad);
        for (int i = lane; i < P.Npad / 4; i += 64)
            dst[i] = *(const uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb));
    }
    if (WL) {
        const double fcur = wl_cur_feat();
        unsafeAtomicAdd(wl_pend + lane, wl_pend_sum()); // the last step's per-bin sums
        if (lane < P.F) featp[lane] = fcur;
        for (int i = lane; i < P.wl.L; i += 64) {
            P.wl.entropy[(size_t)r * P.wl.L + i] = wl_S[i];
            P.wl.hist[(size_t)r * P.wl.L + i] = wl_Hh[i];
            P.wl.occur[(size_t)r * P.wl.L + i] = wl_Oc[i];
        }
        if (lane == 0) {
            P.wl.m[r] = wl_m;
            P.wl.counter[r] = wl_counter;
        }
    } else {
        s_feat[lane] = 0.0;
        reduce_features(s_feat);
        if (lane < P.Fce) featp[lane] = base_feat + s_feat[lane];
    }
    if (FAST) {
        double lane_e = 0.0;
#pragma unroll
        for (int it = 0; it < NSLOT; ++it) lane_e = fma(wgt[it], acc[it], lane_e);
        H += wave_sum_all(lane_e) - acc_mu;
    }
    if (btype && lane == 0) {
        P.bias[r] += bias_acc;
#pragma unroll
        for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k)
            if (k < brows) P.charge[(size_t)r * SMOLMC_MAX_BIAS_ROWS + k] = chg[k];
    }
    if (REPLAY && lane == 0 && rp_bad) atomicOr(P.rp_err, 1);
    if (lane == 0) {
        if (!WL && HAS_EW) featp[P.Fce] += acc_ew;
        if (!WL && HAS_MU) featp[P.Fce + (HAS_EW ? 1 : 0)] += acc_mu;
        P.enthalpy[r] = H;
        P.nsteps[r] = step;
        P.nacc[r] += nacc_add;
        if (P.steps) P.last_acc[r] = (uint8_t)(nacc_add != nacc_before);
    }
}


// Philox out of line for the TableFlip kernel: its step loop is tens of KB of code (the instruction
// cache is 64 KB per two CUs) and holds many calls, none of them on the per-step path.
__device__ __noinline__ philox_out philox_call(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t k0, uint32_t k1) {
    return philox4x32_10(c0, c1, c2, 0u, k0, k1);
}

// sum over species c of ln n_c! - ln (n_c + u_c)!  (mcusher.py:694-709) with lane c holding the
// count n_c and its change u_c: every lane walks its own |u_c| table entries ln(k) (host libm; LDS
// copy when it fits, else HBM) with the first four reads in flight together -- a scalar loop over
// species and k costs one dependent (scalar-cache or LDS) round trip per entry -- and the lane
// partials are then added in species order.  (Deferring the sum to the point where the exponent
// is formed, to hide the table latency behind the site selection, measured slower.)
__device__ __forceinline__ double table_log_count_ratio(const double *lnt, int u, int n0, int nc) {
    const int au = u < 0 ? -u : u;
    const int sgn = u < 0 ? -1 : 1;           // entries n0+1 .. n0+u (u > 0) or n0 .. n0+u+1 (u < 0)
    const int first = u < 0 ? n0 : n0 + 1;
    double t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = k < au ? lnt[first + sgn * k] : 0.0;
    double part = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) part += t[k];
    for (int k = 4; k < au; ++k) part += lnt[first + sgn * k];
    part = u < 0 ? part : -part;
    double tot = 0.0;
    for (int c = 0; c < nc; ++c)
        tot += __hiloint2double((int)rdlane((uint32_t)__double2hiint(part), c),
                                (int)rdlane((uint32_t)__double2loint(part), c));
    return tot;
}

// (defined in mc_lean_multi.h, which the translation units of the Wang-Landau table kernels include)
__device__ __noinline__ void wl_multi_row_swap(double *grows, double *crow, int old_bin, int new_bin, int F, int lane, int sum_mode);

// ----------------------------------------------------------------------------
// TableFlip kernel (charge-neutral semigrand steps, smol/moca/kernel/mcusher.py:397-711)
// for lean-eligible models: one site class, one contiguous active sublattice, interaction
// features, optional mu row and compact Ewald.  A step is either a canonical Swap (with
// probability swap_weight, or when no table direction is feasible) or a flip-table
// direction u: -u[c] random sites of every depleted species are picked without
// replacement (rejection over the candidate stream, 256 candidates per wave round) and
// randomly re-assigned to the enriched species; the a-priori factor
// log(p_next/p_now) + sum ln n_now! - ln n_next! enters the Metropolis exponent.
// Flips of a step are evaluated sequentially against the LDS occupancy with each flip
// applied tentatively (expansion.py:217-229) and undone on rejection.
// ----------------------------------------------------------------------------
// EWM: 0 = no Ewald term, 1 = compact Ewald with per-proposal row sums, 2 = potential field in LDS.
// A template parameter, not a runtime flag: the unused variants' pointers and code otherwise stay
// live across the step loop (the kernel spills SGPRs as it is).
// REPLAY: the steps come from the host as records of SMOLMC_STEP_ROW ints (smolmc_replay: the
// reference's own proposals in its Generator's order); the kernel derives the table direction of a
// record from its count changes (_get_flip_id, mcusher.py:641-654), evaluates the a-priori factor with
// the code of the native path (or takes the given one) and reports accept flag, enthalpy and factor
// per step.
// BIAS (round 6; table_bias_n*.hip): an MCBias term in the exponent (metropolis.py:43-44) -- the reference composes any
// usher with any bias (kernel/base.py:192-239) and until now TableFlip with a bias ran on the universal kernel.  The
// pair tables of the biased lean kernels (bias_pair[row][old * 8 + new]) are read lane-parallel, lane f = flip f; the
// flips of a table step touch distinct sites, so "the last flip of a site counts" (bias.py:75-93) is every flip.
// WLT (round 6; table_wl_n*.hip): the Wang-Landau kernel (kernel/wanglandau.py:175-266, any update_period) with TableFlip
// proposals -- the reference composes any usher with any kernel and until now this pair ran on the universal kernel.
// The accept rule is S[bin] - S[new bin] + the a-priori factor of the step (wanglandau.py:197-198); the per-walker
// state (entropies, counted steps, the log of finished runs) is mc_lean_multi_kernel's (mc_lean_multi.h: WLK) behind
// the potential field in LDS; the current feature vector lives in the lanes of one register and the enthalpy is
// carried as the reference carries it (:216-218).
template <int NSLOT, int MM, int EWM, bool REPLAY = false, bool BIAS = false, bool WLT = false>
__global__ void __launch_bounds__(512) mc_table_kernel(const LeanParams P) { // (four or eight walkers per workgroup, two waves per SIMD)
    static_assert(!(WLT && BIAS), "Cannot apply bias to Wang-Landau simulation (wanglandau.py:127-128)");
    static_assert(!(WLT && REPLAY), "Wang-Landau TableFlip replays take the universal kernel");
    constexpr bool has_ew = EWM != 0, ew_field = EWM == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    // (launch slot -> walker: see update_walker_order, engine.hip)
    const int slot = uni(blockIdx.x * nwaves + wave);
    const int r = (P.order != nullptr && slot < P.R) ? uni(P.order[slot]) : slot;
    double *s_dt = (double *)smem;
    double *s_mu = s_dt + P.dt_len; // 8 doubles
    const size_t per_wave = (size_t)P.Nlds + 64 * 8 + 64 + (ew_field ? (size_t)P.ew_nact * 8 : 0) +
                            (WLT ? wl_multi_wave_bytes(P.wl.L, P.F, P.wl.sum_mode) : 0);
    double *s_q = s_mu + 8, *s_dg = s_mu + 16; // field mode: charge / diagonal term per code
    // block-shared copies of the flip table (<= 8 vectors x 8 codes), its weights and ln(k)
    double *s_tfw = s_mu + 24;
    int *s_tf = (int *)(s_tfw + 16);
    double *s_ln = s_tfw + 16 + 32;
    unsigned char *wbase = (unsigned char *)(s_ln + P.tf_ln_len) + (size_t)wave * per_wave;
    uint8_t *occ = wbase;
    double *s_feat = (double *)(wbase + P.Nlds);
    int *s_cnt = (int *)(s_feat + 64); // species counts of the walker [<= 8]
    double *phi = (double *)(wbase + P.Nlds + 64 * 8 + 64); // Ewald potential field [ew_nact]
    // WLT: S f64 [L] | counted steps u32 [L] | update_period 1 (sums): log of SMOLMC_WLM_LOG finished runs [F] -- else
    // (running means): occurrences at launch start f64 [L] and SMOLMC_WL_ROWS cached rows [F] (wl_multi_wave_bytes)
    const int wl_sum_mode = WLT ? P.wl.sum_mode : 1;
    double *wl_S = phi + (ew_field ? P.ew_nact : 0);
    uint32_t *wl_cnt = (uint32_t *)(wl_S + (WLT ? P.wl.L : 0));
    double *wl_occb = (double *)((unsigned char *)wl_cnt + (WLT ? (((size_t)P.wl.L * 4 + 7) & ~(size_t)7) : 0));
    double *s_rows = wl_occb + ((WLT && !wl_sum_mode) ? P.wl.L : 0);
    const int swa = P.swz_a, swm = P.swz_m, swb = P.swz_b;
    const bool has_mu = P.mu_row != nullptr;
    for (int i = threadIdx.x; i < P.dt_len; i += blockDim.x) s_dt[i] = P.dt[i];
    if (threadIdx.x < 8) s_mu[threadIdx.x] = (has_mu && threadIdx.x < P.ncodes) ? P.mu_row[threadIdx.x] : 0.0;
    if (ew_field && threadIdx.x < 8) {
        s_q[threadIdx.x] = P.ew_qrow[threadIdx.x];
        s_dg[threadIdx.x] = P.ew_dgrow[threadIdx.x];
    }
    for (int i = threadIdx.x; i < 2 * P.tf_n; i += blockDim.x) s_tfw[i] = P.tf_w[i];
    for (int i = threadIdx.x; i < P.tf_n * P.ncodes; i += blockDim.x) s_tf[i] = P.tf_table[i];
    for (int i = threadIdx.x; i < P.tf_ln_len; i += blockDim.x) s_ln[i] = P.tf_ln[i];
    const bool live = r < P.R;
    if (live) {
        const uint32_t *src = (const uint32_t *)(P.occ + (size_t)r * P.Npad);
        for (int i = lane; i < P.Npad / 4; i += 64)
            *(uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb)) = src[i];
        s_feat[lane] = 0.0;
        if (lane < 16) s_cnt[lane] = 0;
        if (ew_field)
            for (int j = lane; j < P.ew_nact; j += 64) phi[j] = P.ew_phi[(size_t)r * P.ew_nact + j];
        if (WLT) {
            for (int i = lane; i < P.wl.L; i += 64) {
                wl_S[i] = P.wl.entropy[(size_t)r * P.wl.L + i];
                wl_cnt[i] = 0u;
                if (!wl_sum_mode) wl_occb[i] = (double)P.wl.occur[(size_t)r * P.wl.L + i];
            }
            if (!wl_sum_mode)
                for (int i = lane; i < SMOLMC_WL_ROWS * P.F; i += 64) s_rows[i] = 0.0;
        }
    }
    __syncthreads();
    if (!live) return;
    const int nc = P.ncodes, sbase = P.sbase;
    const uint32_t nact = (uint32_t)P.nact, nt8 = P.nt8, snt8 = P.snt8;
    for (int a = lane; a < (int)nact; a += 64)
        atomicAdd(&s_cnt[(int)occ[lean_swz(sbase + a, swa, swm, swb)]], 1);
    // lane-indexed working copies: lane c (< ncodes) holds the count of species c and column c
    // of every flip vector; feasibility of a direction is then one compare + one ballot
    int vcnt = lane < P.ncodes ? s_cnt[lane] : 0;
    int vtf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) vtf[i] = (i < P.tf_n && lane < P.ncodes) ? s_tf[i * P.ncodes + lane] : 0;
    // bit idx of the result: direction idx (2 i = +vector i, 2 i + 1 = -vector i) keeps every
    // count inside [0, n_active] when applied to the counts vc (flip_weights_mask, math.py:832-867)
    // (the rarely executed pieces -- feasibility masks, weight sums, a-priori factors: only after an
    // accepted table step -- re-read their parameters from the kernel-argument segment, see
    // rare_params: everything read from P stays in SGPRs across the whole step loop otherwise)
    // (one batch of scalar loads per recomputation: every separate one costs a full wait)
    auto feasible = [&](const int vc, const int tfn, const int na) -> unsigned {
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < tfn) {
                const int vp = vc + vtf[i], vm = vc - vtf[i];
                if (__ballot(vp < 0 || vp > na) == 0ull) m |= 1u << (2 * i);
                if (__ballot(vm < 0 || vm > na) == 0ull) m |= 2u << (2 * i);
            }
        return m;
    };
    // lane c: the largest count change of species c in any direction.  Counts that stay at least
    // that far from both limits make EVERY direction feasible -- the usual case away from the
    // composition limits, where the feasibility mask and the weight sums then need no recomputation
    int vmaxu = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) vmaxu = max(vmaxu, vtf[i] < 0 ? -vtf[i] : vtf[i]);
    auto all_feasible = [&](const int vc) -> bool {
        return __ballot(lane < nc && (vc - vmaxu < 0 || vc + vmaxu > (int)nact)) == 0ull;
    };
    // lane idx (< 2 tf_n): the enriched species of direction idx in the order the assignment draws
    // them (species ascending, u_c entries each; mcusher.py:627-631) as packed nibbles -- up to
    // SMOLMC_MAX_STEP_FLIPS = 8 of them
    uint32_t venr = 0;
    if (lane < 2 * P.tf_n) {
        const int sgn = (lane & 1) ? -1 : 1;
        int k = 0;
        for (int c = 0; c < P.ncodes; ++c) {
            const int u = sgn * s_tf[(lane >> 1) * P.ncodes + c];
            for (int z = 0; z < u && k < 8; ++z, ++k) venr |= (uint32_t)c << (4 * k);
        }
    }
    // ... and the depleted species in the order the site scan wants them (species ascending, -u_c
    // entries each), with their number (= the number of flips of the direction)
    uint32_t vdep = 0;
    int vncol = 0;
    if (lane < 2 * P.tf_n) {
        const int sgn = (lane & 1) ? -1 : 1;
        for (int c = 0; c < P.ncodes; ++c) {
            const int u = sgn * s_tf[(lane >> 1) * P.ncodes + c];
            for (int z = 0; z < -u; ++z, ++vncol)
                if (vncol < 8) vdep |= (uint32_t)c << (4 * vncol);
        }
    }
    const uint32_t lane4 = (uint32_t)(lane & 7) * 4u;
    const double vw = lane < 2 * P.tf_n ? s_tfw[lane] : 0.0; // lane idx: weight of direction idx
    auto weight_of = [&](const int idx) -> double {
        return __hiloint2double((int)rdlane((uint32_t)__double2hiint(vw), idx),
                                (int)rdlane((uint32_t)__double2loint(vw), idx));
    };
    auto masked_sum = [&](const unsigned m, const int tfn) -> double { // sum of the weights of the set directions
        double sw = 0.0;
        const int n2 = 2 * tfn;
        for (int idx = 0; idx < n2; ++idx)
            if ((m >> idx) & 1u) sw += weight_of(idx);
        return sw;
    };

    uint32_t doff8[NSLOT], st8[NSLOT][MM], sfeat[NSLOT];
    double wgt[NSLOT], acc[NSLOT], sfs[NSLOT];
#pragma unroll
    for (int it = 0; it < NSLOT; ++it) {
        const LeanSlot sl = P.slots[it * 64 + lane];
        doff8[it] = sl.doff8;
        sfeat[it] = sl.feat;
        sfs[it] = sl.live ? sl.fs : 0.0;
#pragma unroll
        for (int m = 0; m < MM; ++m) st8[it][m] = sl.stride8[m];
        wgt[it] = sl.w;
        acc[it] = 0.0;
    }
    double H = P.enthalpy[r];
    const double nbeta = -P.beta[r];
    unsigned long long step = P.nsteps[r];
    uint32_t nacc_add = 0; // accepted steps of this launch (< 2^30 steps per launch)
    // MCBias: running bias and, for the square biases, the running A_k . n - b_k of every hyperplane (one for
    // SquareChargeBias: the net charge)
    const int tb_type = BIAS ? P.bias_type : 0;
    const int tb_rows = (tb_type && tb_type != SMOLMC_BIAS_FUGACITY) ? P.bias_rows : 0;
    double tb_acc = 0.0, tb_chg[SMOLMC_MAX_BIAS_ROWS];
#pragma unroll
    for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k) tb_chg[k] = (BIAS && k < tb_rows) ? P.charge[(size_t)r * SMOLMC_MAX_BIAS_ROWS + k] : 0.0;
    const uint32_t key0_ = (uint32_t)P.seeds[r], key1_ = (uint32_t)(P.seeds[r] >> 32);
    // The ten Philox round keys (key + i * Weyl constant) are loop invariant and the compiler
    // parks all twenty of them in SGPRs across the step loop, which then spills; the keys are
    // made opaque at every call so that the round keys are re-derived there (20 scalar adds per
    // Philox call, a few calls per 16 steps).
#define key0 opaque_u32(key0_)
#define key1 opaque_u32(key1_)
    double acc_mu = 0.0, acc_ew = 0.0;
    int last_acc = 1;
    double *featp = P.features + (size_t)r * P.F;
    const double base_feat = lane < P.F ? featp[lane] : 0.0;
    // ---- Wang-Landau state (WLT; as mc_lean_multi_kernel's WLK with update_period 1) ----
    double fcur = base_feat;  // lane f < F: the walker's current feature vector (_current_features); H is _current_enthalpy
    double wl_m = WLT ? P.wl.m[r] : 0.0;
    int wb = 0;               // current bin (walkers start inside the window: smolmc_set_state)
    if (WLT) wb = min(max(uni((int)floordiv_exact(H - P.wl.vmin, P.wl.bin)), 0), P.wl.L - 1);
    const long long wl_counter0 = WLT ? P.wl.counter[r] : 0;
    const uint32_t wl_check = WLT ? (uint32_t)P.wl.check : 0u, wl_upd = WLT ? (uint32_t)P.wl.update : 1u;
    // (check period 0 = no device-side check: the remainder starts at 1 and cannot wrap to 0 inside a launch of < 2^30 steps)
    uint32_t wl_rem_check = (WLT && wl_check) ? (uint32_t)uni((int)(wl_counter0 % (long long)wl_check)) : 1u;
    uint32_t wl_rem_upd = WLT ? (uint32_t)uni((int)(wl_counter0 % (long long)wl_upd)) : 0u;
    uint32_t wl_run_n = 0;    // sums: post-steps of the current (bin, features) state not yet in its row
    int vtag = -1;            // sums: lane i < SMOLMC_WLM_LOG: the bin of log entry i; means: lane i < SMOLMC_WL_ROWS: the bin cached in row i
    int wl_nlog = 0;
    // running means (update_period > 1): the cached row of a bin (direct-mapped; the row a slot held goes back to HBM)
    auto wl_row_of = [&](const int bin) -> double * {
        const int slot = bin & (SMOLMC_WL_ROWS - 1);
        const int tag = (int)rdlane((uint32_t)vtag, slot);
        double *crow = s_rows + (uint32_t)slot * (uint32_t)P.F;
        if (tag != bin) {
            const LeanParamsKernarg Q = rare_params();
            wl_multi_row_swap(Q->wl.meanf + (size_t)r * Q->wl.L * Q->F, crow, tag, bin, Q->F, lane, 0);
            vtag = lane == slot ? bin : vtag;
        }
        return crow;
    };
    // shadow copies of the feature cells for the accepted steps' deltas (see mc_wl.h): lane l adds into copy l % wl_k,
    // a reader sums the copies; cell 63 is never written (the address of "no copy")
    const int wl_F = WLT ? P.F : 1;
    const int wl_k = max(1, min(8, 63 / max(wl_F, 1)));
    const uint32_t wl_shadow = (uint32_t)((lane % wl_k) * wl_F);
    const int wl_rd0 = lane < wl_F ? lane : 63, wl_rdstep = lane < wl_F ? wl_F : 0; // copy k of feature `lane`: cell wl_rd0 + k wl_rdstep
    const bool wl_zero_lane = lane < wl_k * wl_F;
    // finished runs {bin, run_n * features} are logged in LDS and go to the rows of per-bin feature sums in HBM in one
    // burst of fire-and-forget atomics (64 / F entries per instruction; the rows of a walker are touched by its own wave only)
    const int wl_epi = max(1, 64 / max(wl_F, 1)), wl_lane_e = lane / max(wl_F, 1), wl_lane_f = lane - wl_lane_e * wl_F;
    auto wl_log_flush = [&]() {
        const LeanParamsKernarg Q = rare_params();
        double *grows = Q->wl.meanf + (size_t)r * Q->wl.L * Q->F;
        const int qF = Q->F;
        for (int base = 0; base < wl_nlog; base += wl_epi) {
            const int e = base + wl_lane_e;
            const int bin = __shfl(vtag, e & 63); // (uniform control flow: ds_bpermute reads switched-off lanes otherwise)
            if (wl_lane_e < wl_epi && e < wl_nlog)
                unsafeAtomicAdd(grows + (size_t)bin * qF + wl_lane_f, s_rows[(uint32_t)base * (uint32_t)qF + lane]);
        }
        wl_nlog = 0;
    };
    auto wl_flush_run = [&]() {
        if (wl_run_n != 0u) {
            if (lane < wl_F) s_rows[(uint32_t)wl_nlog * (uint32_t)wl_F + lane] = (double)wl_run_n * fcur;
            vtag = lane == wl_nlog ? wb : vtag;
            wl_run_n = 0u;
            if (++wl_nlog == SMOLMC_WLM_LOG) wl_log_flush();
        }
    };
    // 32-bit loop state (the host splits launches at 2^30 steps): the kernel is short of SGPRs
    uint32_t smp_countdown = P.smp.every ? (uint32_t)P.smp.every : 0xffffffffu, smp_index = 0; // (no sampling: never reaches zero)
    uint32_t W0 = 0, W1 = 0, W2 = 0, W3 = 0;
    uint32_t batch_base = ~0u; // low word of the batch's first step: consecutive steps change it exactly when the batch changes
    uint32_t w_site_carry = 0;
    constexpr int ROW = NSLOT * MM;
    constexpr int NW = ROW / 2;
    constexpr uint32_t SITE_BYTES = 64u * ROW * 2u;
    const __amdgpu_buffer_rsrc_t idx_rs =
        __builtin_amdgcn_make_buffer_rsrc((void *)P.idx, 0, 0x7fffffff, 0x00020000);
    const uint32_t lane_voff = (uint32_t)lane * (ROW * 2u);
    // (two kernel arguments every step needs, parked in vector registers: the kernel is short of
    // SGPRs and the compiler otherwise re-reads them from the kernel-argument segment -- a scalar
    // load and a full wait -- in the middle of every step)
    uint32_t ew_na_v = has_ew ? (uint32_t)P.ew_nact : 0u;
    double ew_coef_v = has_ew ? P.ew_coef : 0.0;
    asm volatile("" : "+v"(ew_na_v), "+v"(ew_coef_v));
    // cached between accepted table steps: feasible directions at the current counts, their weight
    // sum, and (lane dir of vlp, bit dir of lp_valid) the log a-priori factor of direction dir
    bool head_valid = false;
    unsigned feas_now = 0;
    unsigned long long lp_valid = 0ull;
    double sumw = 0.0, vlp = 0.0, vcum = 0.0;
    int last_feas = -1;
    // One flip vector (the usual charge-neutral exchange): the counts live on a line, count = start + kpos * u,
    // and the factor of a direction is a function of kpos alone -- that of the reverse direction one position
    // further its negative: f(k, -u) = -f(k - 1, +u) (the count terms swap sides, p_next / p_now inverts).
    // vlp then holds F(k) = f(k, +u) for the positions around kpos, lane k & 63 (bit k & 63 of lp_valid), and an
    // accepted table step invalidates nothing: a hot walker wanders back and forth over a few positions and
    // recomputed both factors (a table read from L2 + a wave reduction each) after every accepted table step.
    // Several vectors: lane d / bit d hold direction d at the current counts, dropped when the counts change.
#ifdef SMOLMC_NO_LP_LINE // A/B switch
    const bool lp_line = false;
#else
    const bool lp_line = P.tf_n == 1;
#endif
    int kpos = 0;
    // Proposal batch: the proposals of 64 consecutive steps computed at once, lane l <-> step
    // (step & ~63) + l, all on the vector unit (the step-at-a-time proposal below is a chain of
    // dependent scalar instructions -- ballot, find-first, readlane, compare -- of one wave:
    // ~4000 cycles per step where the batch costs ~40 vector instructions per step).  A proposal
    // is a function of the random words of its step, of the feasibility mask of the directions
    // (species counts) and of the species at the sites its scan EXAMINES (site 1 and the
    // candidates up to the partner for a swap; the site candidates up to the last pick for a
    // table step).  Those sites are kept (16 x u16 per lane); after every accepted step the
    // lanes that examined a changed site, and all lanes when the feasibility mask changed, are
    // marked stale, and a stale step -- as every step the batch does not cover: more than four
    // flips, more than 32 site candidates, no swap partner among the first 12 candidates --
    // is proposed by the step-at-a-time code, which is the definition (oracle: propose_table_flip).
    uint32_t q_base = ~0u;           // low word of the batch's first step
    uint32_t q_meta = 0;             // bit 0 covered | bit 1 swap | bits 2-4 flips | bits 5-8 direction
    uint32_t q_s01 = 0, q_s23 = 0;   // sites of the flips (u16 each)
    uint32_t q_pack = 0;             // old species of flip f: nibble f; new species: nibble 4 + f
    uint32_t q_w1 = 0;               // W(step, 0, 1): the site word of the NEXT step
    uint32_t q_c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) q_c[i] = 0xffffffffu;
    double q_logu = 0.0;             // log of the acceptance uniform
    unsigned long long q_stale = ~0ull;
    auto compute_head = [&]() {
        const LeanParamsKernarg Q = rare_params();
        const int tfn = Q->tf_n, na = Q->nact;
        feas_now = feasible(vcnt, tfn, na);
        sumw = masked_sum(feas_now, tfn);
        head_valid = true;
        lp_valid = 0ull;
        // running sums of the feasible weights, lane idx <-> direction idx, added in the
        // order choose_section_from_partition adds them: the per-step choice below is then
        // one compare + ballot instead of a loop of readlanes and float64 adds
        double c = 0.0;
        last_feas = -1;
        const int n2 = 2 * tfn;
        for (int idx = 0; idx < n2; ++idx)
            if ((feas_now >> idx) & 1u) {
                c += weight_of(idx);
                if (lane == idx) vcum = c;
                last_feas = idx;
            }
    };
    auto propose_batch = [&](const unsigned long long b0) { // b0: first step of the block
        const LeanParamsKernarg Q = rare_params();
        uint32_t carry; // lane 0's site word W(b0 - 1, 0, 1)
        if (q_base == (uint32_t)b0 - 64u) {
            carry = rdlane(q_w1, 63);
        } else {
            const unsigned long long sp = b0 - 1ull;
            carry = (uint32_t)uni((int)philox_call((uint32_t)sp, (uint32_t)(sp >> 32), 0u, key0, key1).w[1]);
        }
        q_base = (uint32_t)b0;
        q_stale = 0ull;
        const unsigned long long st = b0 + (unsigned)lane;
        const uint32_t c0 = (uint32_t)st, c1 = (uint32_t)(st >> 32);
        const philox_out o0 = philox_call(c0, c1, 0u, key0, key1);
        q_logu = log(philox_u53(o0.w[2], o0.w[3]));
        q_w1 = o0.w[1];
        uint32_t wsite = (uint32_t)__shfl((int)o0.w[1], (lane + 63) & 63);
        wsite = lane == 0 ? carry : wsite;
        q_meta = 0u;
        if (!(sumw > 0.0)) return; // no feasible direction: every step a swap -- left to the step-at-a-time code
        const bool is_swap = (double)o0.w[0] * (1.0 / 4294967296.0) < Q->tf_sw;
        const philox_out o1 = philox_call(c0, c1, 1u, key0, key1);
        const philox_out o2 = philox_call(c0, c1, 2u, key0, key1);
        // direction, its species lists (choose_section_from_partition, math.py:870-893)
        const double target = (double)o1.w[0] * (1.0 / 4294967296.0) * sumw;
        int d = -1;
        {
            const int n2 = 2 * Q->tf_n;
            for (int idx = 0; idx < n2; ++idx) {
                const double c = __hiloint2double((int)rdlane((uint32_t)__double2hiint(vcum), idx),
                                                  (int)rdlane((uint32_t)__double2loint(vcum), idx));
                if (((feas_now >> idx) & 1u) && d < 0 && target < c) d = idx;
            }
            if (d < 0) d = last_feas;
        }
        const uint32_t dep = (uint32_t)__shfl((int)vdep, d), enr = (uint32_t)__shfl((int)venr, d);
        const int ncol = __shfl(vncol, d);
#pragma unroll
        for (int i = 0; i < 16; ++i) q_c[i] = 0xffffffffu;
        uint32_t meta = 0u, pack = 0u;
        int col0 = -1, col1 = -1, col2 = -1, col3 = -1;
        if (is_swap) {
            // Swap.propose_step (mcusher.py:176-200): the first 12 candidates c_t = W(step, 1 + t % 3, t / 3)
            const philox_out o3 = philox_call(c0, c1, 3u, key0, key1);
            const int s1 = sbase + (int)__umulhi(wsite, nact);
            const int sp1 = (int)occ[lean_swz(s1, swa, swm, swb)];
            q_c[0] = 0xffff0000u | (uint32_t)s1;
            int found = -1, fo = 0;
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                const uint32_t w = (t % 3 == 0 ? o1 : t % 3 == 1 ? o2 : o3).w[t / 3];
                const int cs = sbase + (int)__umulhi(w, nact);
                const int v = (int)occ[lean_swz(cs, swa, swm, swb)];
                if (found < 0) {
                    const int slot = 1 + t;
                    q_c[slot >> 1] = (slot & 1) ? ((q_c[slot >> 1] & 0x0000ffffu) | ((uint32_t)cs << 16))
                                                : ((q_c[slot >> 1] & 0xffff0000u) | (uint32_t)cs);
                    if (v != sp1) { found = cs; fo = v; }
                }
            }
            if (found >= 0) {
                col0 = s1;
                col1 = found;
                pack = (uint32_t)sp1 | ((uint32_t)fo << 4) | ((uint32_t)fo << 16) | ((uint32_t)sp1 << 20);
                meta = 1u | 2u | (2u << 2);
            }
        }
        // table steps: the sites of the depleted species from the candidate stream
        // c_t = W(step, 4 + t / 4, t % 4); one pass, the wanted species is that of the next pick
        int k = 0;
        bool scanning = !is_swap && ncol >= 1 && ncol <= 4;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (__ballot(scanning) != 0ull) {
                const philox_out o = philox_call(c0, c1, 4u + (uint32_t)b, key0, key1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = 4 * b + j;
                    const int cs = sbase + (int)__umulhi(o.w[j], nact);
                    const int v = (int)occ[lean_swz(cs, swa, swm, swb)];
                    if (scanning) {
                        q_c[t >> 1] = (t & 1) ? ((q_c[t >> 1] & 0x0000ffffu) | ((uint32_t)cs << 16))
                                              : ((q_c[t >> 1] & 0xffff0000u) | (uint32_t)cs);
                        const int want = (int)((dep >> (4 * k)) & 15u);
                        const bool dup = cs == col0 || cs == col1 || cs == col2 || cs == col3;
                        if (v == want && !dup) {
                            col0 = k == 0 ? cs : col0;
                            col1 = k == 1 ? cs : col1;
                            col2 = k == 2 ? cs : col2;
                            col3 = k == 3 ? cs : col3;
                            pack |= (uint32_t)v << (4 * k);
                            k++;
                            scanning = k < ncol;
                        }
                    }
                }
            }
        }
        if (!is_swap && ncol >= 1 && ncol <= 4 && k == ncol) {
            // the random assignment to the enriched species (mcusher.py:627-631): the k-th draw
            // W(step, 2, k) takes the rr-th pick still available
            uint32_t avail = (1u << ncol) - 1u, left = (uint32_t)ncol;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                if (kk < ncol) {
                    const uint32_t rr = __umulhi(o2.w[kk], left);
                    uint32_t m = avail;
#pragma unroll
                    for (int z = 0; z < 3; ++z) m = z < (int)rr ? (m & (m - 1u)) : m;
                    const int pj = __ffs((int)m) - 1;
                    avail &= ~(1u << pj);
                    left--;
                    pack |= ((enr >> (4 * kk)) & 15u) << (16 + 4 * pj);
                }
            meta = 1u | ((uint32_t)ncol << 2) | ((uint32_t)d << 5);
        }
        q_meta = meta;
        q_s01 = ((uint32_t)col0 & 0xffffu) | ((uint32_t)col1 << 16);
        q_s23 = ((uint32_t)col2 & 0xffffu) | ((uint32_t)col3 << 16);
        q_pack = pack;
    };
    // the 16-step word batch of the step-at-a-time proposal (lane l = block l & 3 of step base + (l >> 2))
    auto word_batch = [&]() {
        const unsigned long long base = step & ~15ull;
        if ((uint32_t)base != batch_base) {
            if (batch_base == (uint32_t)base - 16u) {
                w_site_carry = rdlane(W1, 60);
            } else {
                const unsigned long long sp = base - 1ull;
                w_site_carry = (uint32_t)uni((int)philox_call((uint32_t)sp, (uint32_t)(sp >> 32), 0u, key0, key1).w[1]);
            }
            batch_base = (uint32_t)base;
            const unsigned long long st = base + (unsigned)(lane >> 2);
            const philox_out o = philox_call((uint32_t)st, (uint32_t)(st >> 32), (uint32_t)(lane & 3), key0, key1);
            W0 = o.w[0]; W1 = o.w[1]; W2 = o.w[2]; W3 = o.w[3];
        }
    };

#ifdef SMOLMC_EXP_PHASES // experiment: shader cycles per phase of a step (walker 0 prints the averages)
    long long ph_acc[6] = {0, 0, 0, 0, 0, 0}, ph_cov[3] = {0, 0, 0}, ph_bat = 0, ph_prop[4] = {0, 0, 0, 0};
    long long ph_t = clock64();
#endif
#ifdef SMOLMC_EXP_VGPREF // experiment, MEASURED SLOWER (not in the build): the cross terms of the NEXT step fetched a step ahead, at
    // the end of this step (=1) or before its decision (=2).  Config 5 hot / cold ladder 10.23 / 6.47 ms per sweep -> 11.0 / 7.4
    // either way: two more live VGPRs across the sweep (spills 4 -> 8) and three readlanes per step cost more than the read
    // that -DSMOLMC_EXP_NOVG prices at 8 %.  Kept as a documented negative result, like wave_sum_mfma.
    double vG_pref = 0.0;
    uint32_t vG_tag = ~0u; // low word of the step vG_pref belongs to
#endif
    for (uint32_t steps_left = (uint32_t)P.steps; steps_left != 0u; --steps_left, ++step) {
        // feasibility mask, weight sums of the directions: recomputed (here only: one copy of the
        // code) after the species counts changed; the batch's directions assume the old mask
        if (__builtin_expect(!head_valid, 0)) {
            const unsigned feas_old = feas_now;
            compute_head();
            if (feas_now != feas_old) q_stale = ~0ull;
        }
#ifdef SMOLMC_EXP_PHASES
        const long long tb0 = clock64();
#endif
        if (!REPLAY && __builtin_expect((uint32_t)(step & ~63ull) != q_base, 0)) propose_batch(step & ~63ull); // (first step of a launch)
        const int l6 = (int)(step & 63ull);
        const int l4 = (int)(step & 15ull) * 4;
        const uint32_t q_m = rdlane(q_meta, l6);
#ifdef SMOLMC_NO_TABLE_BATCH // A/B switch: every step through the step-at-a-time proposal
        const bool covered = false;
#else
        const bool covered = !REPLAY && (q_m & 1u) != 0u && ((q_stale >> l6) & 1ull) == 0ull;
#endif
        const uint32_t a01 = rdlane(q_s01, l6), a23 = rdlane(q_s23, l6), pk = rdlane(q_pack, l6);
        double lu = __hiloint2double((int)rdlane((uint32_t)__double2hiint(q_logu), l6),
                                     (int)rdlane((uint32_t)__double2loint(q_logu), l6));
#ifdef SMOLMC_EXP_PHASES
        ph_bat += clock64() - tb0;
#endif
#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); ph_acc[0] += tn - ph_t; ph_t = tn; }
#endif
#ifndef SMOLMC_NO_SETPRIO
        __builtin_amdgcn_s_setprio(1); // wave priority rises through the step (see mc_lean_kernel)
#endif

        // flips of this step live lane-indexed: lane f holds flip f
        int vsite = 0, vnew = 0, vold = 0;
        int nfl = 0, dir = -1;
        // Index rows of the first four flips and the Ewald cross terms G[s_i][s_j] of all flip
        // pairs (lane 8 i + j holds pair j < i): fetched by the proposal as soon as the SITES
        // are known -- for a table step right after the picks, so that the random assignment
        // hides the fetch -- on both proposal paths alike (see mc_lean_kernel on loads that exist
        // on some paths only).
        RowWords<NW> rows[4];
        double vG = 0.0;
        // G[si][sj] for the lanes that hold a flip pair: a random 8-byte read from the rows of the site kernel (24 MB
        // for config 5) every step.  Timing without it (-DSMOLMC_EXP_NOVG, wrong results): -8 % per sweep; the same
        // value through the translation-compressed tables (E8 / S8, then gx: two dependent L2 reads) measured level on
        // the hot ladder and 4 % slower on the cold one -- the dependent chain is what costs, not where it ends.
        auto cross_G = [&](const uint32_t si, const uint32_t sj, const bool want) -> double {
            double g = 0.0;
#ifndef SMOLMC_EXP_NOVG // timing experiment only when defined (wrong results)
            if (want) g = P.ew_G[(size_t)(si * ew_na_v) + (sj - (uint32_t)sbase)];
#endif
            return g;
        };
        auto fetch_rows = [&]() {
#pragma unroll
            for (int f = 0; f < 4; ++f)
                rows[f] = load_row<NW>(idx_rs, lane_voff, rdlane((uint32_t)vsite, f < nfl ? f : 0) * SITE_BYTES);
            if (has_ew && ew_field) {
                const int pi = lane >> 3, pj = lane & 7;
                const int si = __shfl(vsite, pi), sj = __shfl(vsite, pj);
                vG = cross_G((uint32_t)si, (uint32_t)sj, pj < pi && pi < nfl);
            }
        };

#ifdef SMOLMC_EXP_VGPREF
        auto prefetch_next_G = [&]() {
        // the batch knows the sites of the next step; if that step stays covered (this step's marking is done) it takes
        // the value from here instead of waiting for the read in its own evaluation
        if (!REPLAY && has_ew && ew_field && l6 < 63) {
            const uint32_t qn = rdlane(q_meta, l6 + 1);
            const int nfn = (int)((qn >> 2) & 7u);
            if ((qn & 1u) && !((q_stale >> (l6 + 1)) & 1ull) && nfn >= 2) {
                const uint32_t b01 = rdlane(q_s01, l6 + 1), b23 = rdlane(q_s23, l6 + 1);
                const uint32_t t0 = b01 & 0xffffu, t1 = b01 >> 16, t2 = nfn > 2 ? b23 & 0xffffu : t0, t3 = nfn > 3 ? b23 >> 16 : t0;
                const int pi = lane >> 3, pj = lane & 7;
                const uint32_t si = pi == 1 ? t1 : pi == 2 ? t2 : t3, sj = pj == 0 ? t0 : pj == 1 ? t1 : t2;
                vG_pref = cross_G(si, sj, pj < pi && pi < nfn);
                vG_tag = (uint32_t)step + 1u;
            }
        }
        };
#endif
        int vu = 0; // table step: lane c holds the change of the count of species c
        double log_priori = 0.0;
        // count changes (lane c: species c) of table direction d
        auto set_direction = [&](const int d) {
            const int usg = (d & 1) ? -1 : 1;
            // column values of the chosen vector, lane-indexed (register array -> select chain)
#pragma unroll
            for (int i = 0; i < 8; ++i) vu = (i == (d >> 1)) ? vtf[i] : vu;
            vu *= usg; // lane c: change of the count of species c
        };
        // compute_log_priori_factor (mcusher.py:656-711), cached per direction (lane d of vlp)
        auto priori_of = [&](const int d) {
            // where the factor of direction d lives: lane d, or on the line the position of the +u step it is
            // (the negative of) -- see lp_line
            const int slot = lp_line ? ((kpos - (d & 1)) & 63) : d;
            const bool flip_sign = lp_line && (d & 1);
            if (__builtin_expect(!((lp_valid >> slot) & 1ull), 0)) {
                const LeanParamsKernarg Q = rare_params();
                const int tfn = Q->tf_n, na = Q->nact, lnlen = Q->tf_ln_len;
                const double tsw = Q->tf_sw;
                const double *lng = Q->tf_ln;
                // (all directions feasible now and after the step: the same sum, no mask to form)
                const double sum_next = (feas_now == (1u << (2 * tfn)) - 1u && all_feasible(vcnt + vu))
                                            ? sumw : masked_sum(feasible(vcnt + vu, tfn, na), tfn);
                double lf = 0.0;
                // equal weights and equal feasible sums: p_next / p_now is exactly 1 (the common
                // case away from the composition limits), no division / log needed
                const double w_now = weight_of(d), w_back = weight_of(d ^ 1);
                if (!(w_now == w_back && sum_next == sumw)) {
                    const double p_now = (1.0 - tsw) * w_now / sumw;
                    const double p_next = (1.0 - tsw) * w_back / sum_next;
                    lf = log(p_next / p_now);
                }
                lf += table_log_count_ratio(lnlen ? s_ln : lng, vu, vcnt, nc);
                lf = uni_d(lf);
                if (lane == slot) vlp = flip_sign ? -lf : lf;
                lp_valid |= 1ull << slot;
            }
            log_priori = __hiloint2double((int)rdlane((uint32_t)__double2hiint(vlp), slot),
                                          (int)rdlane((uint32_t)__double2loint(vlp), slot));
            if (flip_sign) log_priori = -log_priori;
        };
        if (covered) {
            // the batch's proposal of this step (see propose_batch)
            nfl = (int)((q_m >> 2) & 7u);
            vsite = lane == 0 ? (int)(a01 & 0xffffu) : lane == 1 ? (int)(a01 >> 16)
                  : lane == 2 ? (int)(a23 & 0xffffu) : (int)(a23 >> 16);
            vold = (int)((pk >> lane4) & 15u);
            vnew = (int)(((pk >> 16) >> lane4) & 15u);
            if (!(q_m & 2u)) {
                dir = (int)((q_m >> 5) & 15u);
                set_direction(dir);
            }
            {
                // rows and Ewald cross terms as in fetch_rows, the sites being scalars here
                const uint32_t s0 = a01 & 0xffffu, s1 = a01 >> 16, s2 = nfl > 2 ? a23 & 0xffffu : s0, s3 = nfl > 3 ? a23 >> 16 : s0;
                rows[0] = load_row<NW>(idx_rs, lane_voff, s0 * SITE_BYTES);
                rows[1] = load_row<NW>(idx_rs, lane_voff, s1 * SITE_BYTES);
                rows[2] = load_row<NW>(idx_rs, lane_voff, s2 * SITE_BYTES);
                rows[3] = load_row<NW>(idx_rs, lane_voff, s3 * SITE_BYTES);
                if (has_ew && ew_field) {
                    const int pi = lane >> 3, pj = lane & 7;
                    const uint32_t si = pi == 1 ? s1 : pi == 2 ? s2 : s3, sj = pj == 0 ? s0 : pj == 1 ? s1 : s2;
#ifdef SMOLMC_EXP_VGPREF
                    if (vG_tag == (uint32_t)step) vG = vG_pref;
                    else
#endif
                    vG = cross_G(si, sj, pj < pi && pi < nfl);
                }
            }
#ifdef SMOLMC_EXP_PHASES
            ph_cov[0]++;
#endif
        } else if (REPLAY) {
            // the recorded step: lane f <-> flip f
            const LeanParamsKernarg Q = rare_params();
            const size_t krec = (size_t)r * (uint32_t)Q->steps + ((uint32_t)Q->steps - steps_left);
            const int *rec = Q->rp_steps + krec * SMOLMC_STEP_ROW;
            const int v = lane < SMOLMC_STEP_ROW ? rec[lane] : -1;
            while (nfl < SMOLMC_MAX_STEP_FLIPS && (int)rdlane((uint32_t)v, 2 * nfl) >= 0) nfl++;
            const int ra = __shfl(v, 2 * (lane & 7)), rb = __shfl(v, 2 * (lane & 7) + 1); // (uniform control flow)
            vsite = lane < nfl ? ra : sbase;
            vnew = lane < nfl ? rb : 0;
            // sites of the active sublattice, distinct inside a step (what the usher returns, mcusher.py:612-634)
            bool bad = lane < nfl && (vsite < sbase || vsite >= sbase + (int)nact || vnew < 0 || vnew >= nc);
            for (int f = 0; f < nfl; ++f) bad |= lane < nfl && lane != f && vsite == (int)rdlane((uint32_t)vsite, f);
            int rbad = __ballot(bad) != 0ull ? 2 : 0;
            if (rbad) { nfl = 0; vsite = sbase; vnew = 0; }
            vold = lane < nfl ? (int)occ[lean_swz(vsite, swa, swm, swb)] : 0;
            for (int f = 0; f < nfl; ++f)
                vu += (lane == (int)rdlane((uint32_t)vnew, f)) - (lane == (int)rdlane((uint32_t)vold, f));
            if (__ballot(lane < nc && vu != 0) != 0ull) { // _get_flip_id (mcusher.py:641-654)
                const int tfn = Q->tf_n;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (i < tfn && dir < 0) {
                        if (__ballot(lane < nc && vtf[i] != vu) == 0ull) dir = 2 * i;
                        else if (__ballot(lane < nc && -vtf[i] != vu) == 0ull) dir = 2 * i + 1;
                    }
                if (dir < 0) { rbad |= 1; nfl = 0; vu = 0; } // "Step ... is not in flip table." (:673-674)
            }
            if (rbad && lane == 0) atomicOr(Q->rp_err, rbad);
            double u = uni_d(Q->rp_u[krec]);
            if (u != u) u = 0.0; // NaN: the reference accepted without drawing
            lu = log(u);
            fetch_rows();
        } else {
        word_batch();
        const uint32_t w_site = l4 == 0 ? w_site_carry : rdlane(W1, l4 - 4);
        bool do_swap = (double)rdlane(W0, l4) * (1.0 / 4294967296.0) < P.tf_sw;
#ifdef SMOLMC_EXP_PHASES
        if (q_m & 1u) ph_cov[1]++; else if (do_swap) ph_cov[2]++;
#endif
        if (!do_swap) { // flip_weights_mask (math.py:832-867) at the current counts
            // the species counts only change on accepted table steps: the feasibility mask, its
            // weight sum and the a-priori factor of every direction are kept until then
            if (!(sumw > 0.0)) do_swap = true;
        }
        if (do_swap) {
            // Swap.propose_step (mcusher.py:176-200)
            const int s1 = sbase + (int)__umulhi(w_site, nact);
            const int o1 = uni((int)occ[lean_swz(s1, swa, swm, swb)]);
            int found = -1, fo = 0;
            const uint32_t ws[4] = {W0, W1, W2, W3};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (found < 0) {
                    const int cs = sbase + (int)__umulhi(ws[j], nact);
                    const int v = (int)occ[lean_swz(cs, swa, swm, swb)];
                    const unsigned long long m = __ballot(v != o1) & (0xEull << l4);
                    if (m) {
                        const int b = __ffsll((long long)m) - 1;
                        found = (int)rdlane((uint32_t)cs, b);
                        fo = (int)rdlane((uint32_t)v, b);
                    }
                }
            }
            if (found < 0) {
                for (uint32_t q = 0;; ++q) {
                    const philox_out o = philox_call((uint32_t)step, (uint32_t)(step >> 32),
                                                       4u + 64u * q + (uint32_t)lane, key0, key1);
                    int selsite = -1, selv = 0;
#pragma unroll
                    for (int j = 3; j >= 0; --j) {
                        const int cs = sbase + (int)__umulhi(o.w[j], nact);
                        const int v = (int)occ[lean_swz(cs, swa, swm, swb)];
                        if (v != o1) { selsite = cs; selv = v; }
                    }
                    const unsigned long long m = __ballot(selsite >= 0);
                    if (m) {
                        const int b = __ffsll((long long)m) - 1;
                        found = (int)rdlane((uint32_t)selsite, b);
                        fo = (int)rdlane((uint32_t)selv, b);
                        break;
                    }
                    if ((q & 63u) == 0) {
                        int any = 0;
                        for (uint32_t a = lane; a < nact; a += 64)
                            any |= ((int)occ[lean_swz(sbase + (int)a, swa, swm, swb)] != o1);
                        if (__ballot(any) == 0ull) break;
                    }
                }
            }
            if (found >= 0) {
                nfl = 2;
                vsite = lane == 0 ? s1 : found;
                vnew = lane == 0 ? fo : o1;
                vold = lane == 0 ? o1 : fo;
            }
            fetch_rows();
        } else {
            // choose_section_from_partition (math.py:870-893) with W(step, 1, 0)
            const double target = (double)rdlane(W0, l4 + 1) * (1.0 / 4294967296.0) * sumw;
            {
                const uint32_t hit = (uint32_t)__ballot(target < vcum) & feas_now; // first feasible idx with target < running sum
                dir = hit ? __ffs((int)hit) - 1 : last_feas;
            }
            set_direction(dir);
#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); ph_acc[1] += tn - ph_t; ph_t = tn; }
#endif
            // pick the sites of the depleted species from the candidate stream
            // c_t = W(step, 4 + t / 4, t % 4): 256 candidates per wave round, lane l holds
            // t = 256 round + 4 l + j.  The scan is scalar: per species four ballots (one per j),
            // each pick = first set bit at or after the running stream position.
            int vcol = 0, vcsp = 0, ncol = 0; // collected sites and their (depleted) species, lane-indexed
            uint32_t tpos = 0;        // next stream position inside the current round
            uint32_t round = 0;
            int cs[4] = {0, 0, 0, 0}, cv[4] = {0, 0, 0, 0};
            bool have_round = false;
            // ---- first the 64 leading candidates of the stream, one per lane -------------------
            // (one Philox call and one gather; per depleted species a ballot gives the candidates of
            // that species at or after the running stream position, every pick is its lowest set
            // bit.  A site the stream names twice or a stream that needs more than 64 candidates
            // sends the step to the full scan below.)
            bool fast_done = false;
#ifndef SMOLMC_NO_TABLE_FAST
            {
                const philox_out o = philox_call((uint32_t)step, (uint32_t)(step >> 32), 4u + ((uint32_t)lane >> 2), key0, key1);
                const uint32_t wsel = (lane & 3) == 0 ? o.w[0] : (lane & 3) == 1 ? o.w[1] : (lane & 3) == 2 ? o.w[2] : o.w[3];
                const int cb_site = sbase + (int)__umulhi(wsel, nact);
                const int cvl = (int)occ[lean_swz(cb_site, swa, swm, swb)];
                uint32_t fpos = 0; // next stream position (kept across species)
                bool ok = true;
                uint32_t dep = (uint32_t)__ballot(vu < 0) & ((1u << nc) - 1u); // depleted species
                while (dep != 0u && ok) {
                    const int c = __ffs((int)dep) - 1;
                    dep &= dep - 1u;
                    int need = -(int)rdlane((uint32_t)vu, c);
                    unsigned long long m = __ballot(cvl == c); // bit t: candidate t has species c
                    m = fpos < 64u ? (m >> fpos) << fpos : 0ull;
                    if (__popcll(m) < need || ncol + need > 8) { ok = false; break; }
                    do {
                        const int t = __ffsll((long long)m) - 1;
                        m &= m - 1ull;
                        const int site = (int)rdlane((uint32_t)cb_site, t);
                        if ((__ballot(vcol == site) & ((1ull << ncol) - 1ull)) != 0ull) { ok = false; break; }
                        vcol = lane == ncol ? site : vcol;
                        vcsp = lane == ncol ? c : vcsp;
                        ncol++;
                        fpos = (uint32_t)t + 1u;
                    } while (--need > 0);
                }
                if (ok) fast_done = true;
                else { vcol = 0; vcsp = 0; ncol = 0; } // stream exhausted or repeated site: full scan
            }
#endif
            for (int c = 0; c < nc && !fast_done; ++c) {
                int need = -(int)rdlane((uint32_t)vu, c);
                unsigned long long B[4] = {0ull, 0ull, 0ull, 0ull};
                bool have_masks = false;
                while (need > 0) {
                    if (!have_round) {
                        const philox_out o = philox_call((uint32_t)step, (uint32_t)(step >> 32),
                                                           4u + 64u * round + (uint32_t)lane, key0, key1);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            cs[j] = sbase + (int)__umulhi(o.w[j], nact);
                            cv[j] = (int)occ[lean_swz(cs[j], swa, swm, swb)];
                        }
                        have_round = true;
                        have_masks = false;
                        tpos = 0;
                    }
                    if (!have_masks) {
                        // candidates of species c at stream positions >= tpos (the position is
                        // kept across species); picks then go in stream order, so taking the
                        // minimum and clearing its bit needs no further position masks
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t l0 = tpos > (uint32_t)j ? (tpos - (uint32_t)j + 3u) >> 2 : 0u; // first lane with 4 l + j >= tpos
                            const unsigned long long b = __ballot(cv[j] == c);
                            B[j] = l0 < 64u ? (b >> l0) << l0 : 0ull;
                        }
                        have_masks = true;
                    }
                    // first remaining candidate (branch-free: ffs of an empty mask gives 0, i.e.
                    // position 0xfffffffc + j, larger than any real one)
                    uint32_t best = 0xffffffffu;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        best = min(best, 4u * (uint32_t)(__ffsll((long long)B[j]) - 1) + (uint32_t)j);
                    if (best >= 0xfffffff0u) { round++; have_round = false; continue; }
                    tpos = best + 1u;
                    const int bl = (int)(best >> 2), bj = (int)(best & 3u);
#pragma unroll
                    for (int j = 0; j < 4; ++j) B[j] &= ~((unsigned long long)(bj == j) << bl);
                    const int picked = (int)rdlane((uint32_t)(bj == 0 ? cs[0] : bj == 1 ? cs[1] : bj == 2 ? cs[2] : cs[3]), bl);
                    // a site already collected in this step is skipped (choice without replacement)
                    if (__ballot(lane < ncol && vcol == picked) != 0ull) continue;
                    if (lane == ncol) { vcol = picked; vcsp = c; }
                    ncol++;
                    need--;
                }
            }
#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); ph_acc[2] += tn - ph_t; ph_t = tn; }
#endif
            // The flips of the step are the picks in PICK order (site and old species are known from
            // the scan); their rows are fetched now.  The random assignment to the enriched species
            // (:627-631) then only has to say which pick gets which species: the oracle removes the
            // rr-th entry of the remaining list, i.e. the rr-th pick still available -- a scalar
            // bit mask -- and evaluates the flips in the order it draws them; any order of the same
            // flips gives the same step (the deltas telescope), up to the rounding of the sums.
            vsite = vcol;
            vold = vcsp;
            nfl = ncol;
            fetch_rows();
            {
                // k-th draw: the rr-th pick still available gets the k-th enriched species of the
                // direction (venr); all on the scalar unit, the new species of pick p collected as
                // nibble p of one word
                const uint32_t enr = rdlane(venr, dir);
                uint32_t avail = (1u << ncol) - 1u, newpack = 0u;
                uint32_t left = (uint32_t)ncol;
                for (int k = 0; k < ncol; ++k) {
                    const int wl = l4 + 2 + (k >> 2);
                    const int wj = k & 3;
                    const uint32_t word = rdlane(wj == 0 ? W0 : wj == 1 ? W1 : wj == 2 ? W2 : W3, wl);
                    const int rr = (int)__umulhi(word, left);
                    uint32_t m = avail;
                    for (int z = 0; z < rr; ++z) m &= m - 1u; // drop the rr lowest available picks
                    const int pj = __ffs((int)m) - 1;
                    avail &= ~(1u << pj);
                    left--;
                    newpack |= ((enr >> (4 * k)) & 15u) << (4 * pj);
                }
                vnew = (int)((newpack >> lane4) & 15u);
            }
        }
        } // (step-at-a-time proposal)
#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); ph_prop[covered ? 0 : 1] += tn - ph_t; ph_prop[2] -= tn; }
#endif
        if (REPLAY) {
            const LeanParamsKernarg Q = rare_params();
            const size_t krec = (size_t)r * (uint32_t)Q->steps + ((uint32_t)Q->steps - steps_left);
            const double given = Q->rp_lp ? uni_d(Q->rp_lp[krec]) : __builtin_nan("");
            if (given == given) log_priori = nfl ? given : 0.0;
            else if (dir >= 0) priori_of(dir);
        } else if (dir >= 0) priori_of(dir);
#ifdef SMOLMC_EXP_PHASES
        ph_prop[2] += clock64();
#endif

#ifndef SMOLMC_NO_SETPRIO
        __builtin_amdgcn_s_setprio(2);
#endif
#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); ph_acc[3] += tn - ph_t; ph_t = tn; }
#endif
        // -------- sequential evaluation of the flips of this step -----------------------
        double e = 0.0, pend[NSLOT], ew_part = 0.0, ew_uni = 0.0, dMu = 0.0;
        double vdq = 0.0; // lane f holds the charge change of flip f (potential-field mode)
#pragma unroll
        for (int it = 0; it < NSLOT; ++it) pend[it] = 0.0;
        auto eval_flip = [&](const int f, const RowWords<NW> &row) {
            const int s = (int)rdlane((uint32_t)vsite, f), nw = (int)rdlane((uint32_t)vnew, f);
            const int od = (int)rdlane((uint32_t)vold, f);
            const uint32_t pair = (uint32_t)od * snt8 + (uint32_t)nw * nt8;
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) {
                uint32_t a = doff8[it];
#pragma unroll
                for (int m = 0; m < MM; ++m) a += __umul24(st8[it][m], (uint32_t)occ[bounded(row_entry<NW>(row, it * MM + m), (uint32_t)P.Nlds)]);
                const double d = *(const double *)((const unsigned char *)s_dt + (a + pair));
                e = fma(wgt[it], d, e);
                pend[it] += d;
            }
            if (has_ew) {
                if (ew_field) {
                    // flip f sees the earlier flips of the step through the cross terms
                    const double dq = s_q[nw] - s_q[od];
                    double pot = phi[s - sbase];
                    for (int m = 0; m < f; ++m) {
                        const double dqm = __hiloint2double((int)rdlane((uint32_t)__double2hiint(vdq), m),
                                                            (int)rdlane((uint32_t)__double2loint(vdq), m));
                        const double gfm = __hiloint2double((int)rdlane((uint32_t)__double2hiint(vG), 8 * f + m),
                                                            (int)rdlane((uint32_t)__double2loint(vG), 8 * f + m));
                        pot = fma(dqm, gfm, pot);
                    }
                    ew_uni += 2.0 * dq * pot + (s_dg[nw] - s_dg[od]);
                    if (lane == f) vdq = dq;
                } else {
                    const int W = P.ew_W;
                    const double dq = P.ew_qs[(size_t)s * W + nw] - P.ew_qs[(size_t)s * W + od];
                    ew_part += 2.0 * dq * lean_ewald_partial(P, occ, lane, s, swa, swm, swb);
                    ew_uni += 2.0 * dq * P.ew_frozen[s] +
                              (P.ew_dg[(size_t)s * W + nw] - P.ew_dg[(size_t)s * W + od]);
                }
            }
            if (has_mu) dMu += s_mu[nw] - s_mu[od];
            occ[lean_swz(s, swa, swm, swb)] = (uint8_t)nw; // tentative (every lane, same byte)
        };
        // The first four flips in two phases: (A) every flip's occupancy gathers, its potential read
        // and its tentative write are ISSUED back to back -- the LDS executes in order, so flip
        // f + 1 sees flip f without anybody waiting --, (B) table reads and arithmetic.  One
        // gather round trip and one table round trip per step instead of one of each per flip
        // (the step is latency-bound: at most two waves share a SIMD here).  The compact Ewald
        // form without field (EWM 1) sums over the tentative occupancy per flip and keeps the
        // flip-by-flip order.
        constexpr bool TWO_PHASE = !has_ew || ew_field;
        // (straight-line code per flip count: with a branch per flip inside, the compiler's wait
        // insertion drains the LDS counter at every flip and nothing overlaps)
        auto two_phase = [&](auto nflips) {
            constexpr int NF = decltype(nflips)::value;
            int fsite[NF], fnw[NF], fod[NF];
            uint32_t g[NF][NSLOT * MM];
            double fpot[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                fsite[f] = (int)rdlane((uint32_t)vsite, f);
                fnw[f] = (int)rdlane((uint32_t)vnew, f);
                fod[f] = (int)rdlane((uint32_t)vold, f);
#pragma unroll
                for (int q = 0; q < NSLOT * MM; ++q) g[f][q] = (uint32_t)occ[bounded(row_entry<NW>(rows[f], q), (uint32_t)P.Nlds)];
                fpot[f] = ew_field ? phi[fsite[f] - sbase] : 0.0;
                occ[lean_swz(fsite[f], swa, swm, swb)] = (uint8_t)fnw[f]; // tentative (every lane, same byte)
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int nw = fnw[f], od = fod[f];
                const uint32_t pair = (uint32_t)od * snt8 + (uint32_t)nw * nt8;
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) {
                    uint32_t a = doff8[it];
#pragma unroll
                    for (int m = 0; m < MM; ++m) a += __umul24(st8[it][m], g[f][it * MM + m]);
                    const double d = *(const double *)((const unsigned char *)s_dt + (a + pair));
                    e = fma(wgt[it], d, e);
                    pend[it] += d;
                }
                if (ew_field) { // flip f sees the earlier flips of the step through the cross terms
                    const double dq = s_q[nw] - s_q[od];
                    double pot = fpot[f];
#pragma unroll
                    for (int m = 0; m < f; ++m) {
                        const double dqm = __hiloint2double((int)rdlane((uint32_t)__double2hiint(vdq), m),
                                                            (int)rdlane((uint32_t)__double2loint(vdq), m));
                        const double gfm = __hiloint2double((int)rdlane((uint32_t)__double2hiint(vG), 8 * f + m),
                                                            (int)rdlane((uint32_t)__double2loint(vG), 8 * f + m));
                        pot = fma(dqm, gfm, pot);
                    }
                    ew_uni += 2.0 * dq * pot + (s_dg[nw] - s_dg[od]);
                    if (lane == f) vdq = dq;
                }
                if (has_mu) dMu += s_mu[nw] - s_mu[od];
            }
        };
        // (two instantiations: the code of the step loop competes for the instruction cache; steps of
        // one flip or of more than three are evaluated flip by flip)
        if (TWO_PHASE && nfl == 3) two_phase(std::integral_constant<int, 3>{});
        else if (TWO_PHASE && nfl == 2) two_phase(std::integral_constant<int, 2>{});
        else
            for (int f = 0; f < nfl; ++f)
                eval_flip(f, load_row<NW>(idx_rs, lane_voff, rdlane((uint32_t)vsite, f) * SITE_BYTES));
#ifndef SMOLMC_NO_SETPRIO
        __builtin_amdgcn_s_setprio(3);
#endif
#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); ph_acc[4] += tn - ph_t; ph_t = tn; }
#endif
#if defined(SMOLMC_EXP_VGPREF) && SMOLMC_EXP_VGPREF == 2
        prefetch_next_G(); // (before the decision: more cover, but an accepted step's sweep queues behind it)
#endif
        double dH = wave_sum_all(e);
        double dEw = 0.0;
        if (has_ew) {
            dEw = (ew_field ? 0.0 : wave_sum_all(ew_part)) + ew_uni;
            dH += ew_coef_v * dEw;
        }
        if (has_mu) dH -= dMu;
        // compute_bias_change of the step against the occupancy before it (kernel/base.py:307-311, bias.py:75-93)
        double dB = 0.0, dQ[SMOLMC_MAX_BIAS_ROWS] = {0.0, 0.0, 0.0, 0.0};
        if (BIAS && tb_type && nfl >= 1) {
            const LeanParamsKernarg Q = rare_params();
            const uint32_t pidx = lane < nfl ? (uint32_t)(vold * 8 + vnew) : 0u; // lane f: the species pair of flip f
            if (tb_type == SMOLMC_BIAS_FUGACITY) {
                const double x = Q->bias_pair[pidx];
                for (int f = 0; f < nfl; ++f) // (in the order of the flips, as the reference adds them)
                    dB += __hiloint2double((int)rdlane((uint32_t)__double2hiint(x), f), (int)rdlane((uint32_t)__double2loint(x), f));
            } else {
                double sq_new = 0.0, sq_old = 0.0;
                const double pen = Q->bias_pen;
                const int stride = Q->bias_row_stride;
#pragma unroll
                for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k)
                    if (k < tb_rows) {
                        const double x = Q->bias_pair[(size_t)k * stride + pidx];
                        double xs = 0.0;
                        for (int f = 0; f < nfl; ++f)
                            xs += __hiloint2double((int)rdlane((uint32_t)__double2hiint(x), f), (int)rdlane((uint32_t)__double2loint(x), f));
                        dQ[k] = xs;
                        const double cn = tb_chg[k] + xs;
                        sq_old += tb_chg[k] * tb_chg[k];
                        sq_new += cn * cn;
                    }
                dB = -pen * sq_new - (-pen * sq_old);
            }
        }
        bool accepted;
        int wnb = wb;
        if (WLT) { // WangLandau._accept_step (wanglandau.py:186-202): exact float64 delta, exact floor division
            const LeanParamsKernarg Q = rare_params();
            const double new_h = H + dH, vmin = Q->wl.vmin;
            accepted = false;
            if (__ballot(!(new_h < vmin || new_h >= Q->wl.vmax)) != 0ull) {
                wnb = uni((int)floordiv_exact(new_h - vmin, Q->wl.bin));
                const double ex = wl_S[wb] - wl_S[wnb] + log_priori; // (:197-198)
                accepted = __ballot((ex >= 0.0) || (ex > lu)) != 0ull;
            }
        } else {
            const double exponent = nbeta * dH + log_priori + dB; // metropolis.py:41-44
            accepted = __ballot((exponent >= 0.0) || (exponent > lu)) != 0ull;
        }
        if (accepted) {
            if (BIAS) {
                tb_acc += dB;
#pragma unroll
                for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k) tb_chg[k] += dQ[k];
            }
            if (WLT) {
                // the state (bin, features) ends here: its post-steps go to the bin's row; then features and bin follow
                // the step (_do_accept_step, wanglandau.py:204-220; the enthalpy below with the Metropolis kernels')
                if (wl_sum_mode) wl_flush_run();
#pragma unroll
                for (int it = 0; it < NSLOT; ++it)
                    __hip_atomic_fetch_add(&s_feat[sfeat[it] + wl_shadow], sfs[it] * pend[it], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WAVEFRONT);
                double df = 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) df += s_feat[k < wl_k ? wl_rd0 + k * wl_rdstep : 63];
                if (wl_zero_lane) s_feat[lane] = 0.0;
                if (has_ew) df += lane == P.Fce ? dEw : 0.0;
                if (has_mu) df += lane == P.Fce + (has_ew ? 1 : 0) ? dMu : 0.0;
                fcur += df;
                wb = wnb;
            }
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) acc[it] += pend[it];
            if (dir >= 0) {
                // species counts follow the accepted table direction; the mask of the feasible
                // directions is recomputed unless all were feasible and still are
                const bool was_all = all_feasible(vcnt);
                vcnt += vu;
#ifndef SMOLMC_EXP_NOPRIORI // timing experiment only when defined (wrong results)
                if (lp_line) {
                    // one position along the line; the two lanes whose positions leave the window of 62 around
                    // kpos are dropped (their residues now belong to positions on the other side)
                    kpos += (dir & 1) ? -1 : 1;
                    lp_valid &= ~((1ull << ((kpos + 32) & 63)) | (1ull << ((kpos + 33) & 63)));
                } else {
                    lp_valid = 0ull;
                }
#endif
                if (!(was_all && all_feasible(vcnt))) head_valid = false;
            }
#ifndef SMOLMC_EXP_NOSTALE // timing experiment only when defined (wrong results)
            if (!REPLAY) {
                // batch lanes whose scan examined a site that has just changed are stale (all 32
                // kept sites against every flipped site; unused slots hold 0xffff, no site)
                const bool hit = batch_lane_examined(q_c, vsite, nfl);
#ifdef SMOLMC_EXP_STALEIGN // timing experiment only when defined (wrong results): the marking is computed and dropped
                if (__ballot(hit) == 0x123456789abcull) q_stale = ~0ull;
#else
                q_stale |= __ballot(hit);
#endif
            }
#endif
#ifndef SMOLMC_EXP_NOFIELD // timing experiment only when defined (wrong results)
            if (ew_field) field_apply_flips<1, true>(phi, lane, nfl, vsite, vdq);
#endif
            acc_mu += dMu;
            acc_ew += dEw;
            H += dH;
            nacc_add++;
        } else {
            // undo the tentative flips: lane f restores the site of flip f (the sites of a step
            // are distinct, so the order of the stores does not matter)
            if (lane < nfl) occ[lean_swz(vsite, swa, swm, swb)] = (uint8_t)vold;
        }
#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); ph_acc[5] += tn - ph_t; ph_t = tn; }
#endif
        last_acc = accepted ? 1 : 0;
        if (WLT) { // WangLandau._do_post_step (wanglandau.py:222-266), accepted or not
            if (wl_sum_mode) {
                wl_run_n++;
            } else { // running mean with total = occurrences[bin] as they are now (:233-239)
                double *crow = wl_row_of(wb);
                const double total = wl_occb[wb] + (double)wl_cnt[wb];
                const double inv = 1.0 / (total + 1.0);
                if (lane < P.F) crow[lane] = inv * (fcur + total * crow[lane]);
            }
            if (++wl_rem_upd == wl_upd) { // entropy, histogram, occurrences every update_period steps (:241-245)
                wl_rem_upd = 0u;
                if (lane == 0) {
                    __hip_atomic_fetch_add(&wl_S[wb], wl_m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add(&wl_cnt[wb], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            if (++wl_rem_check == wl_check) wl_rem_check = 0u;
            if (wl_rem_check == 0u) {
                const LeanParamsKernarg Q = rare_params();
                const size_t o = (size_t)r * Q->wl.L;
                wl_m = wl_multi_flatness_check(wl_S, wl_cnt, wl_sum_mode ? nullptr : wl_occb, Q->wl.hist + o, Q->wl.occur + o,
                                               Q->wl.L, Q->wl.flat, Q->wl.div, wl_m, lane);
            }
        }
#ifndef SMOLMC_NO_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if (REPLAY && lane == 0) { // what smolmc_replay returns per step
            const LeanParamsKernarg Q = rare_params();
            const size_t krec = (size_t)r * (uint32_t)Q->steps + ((uint32_t)Q->steps - steps_left);
            Q->rp_acc[krec] = (uint8_t)last_acc;
            Q->rp_H[krec] = H;
            if (Q->rp_lp_out) Q->rp_lp_out[krec] = log_priori;
        }
#if defined(SMOLMC_EXP_VGPREF) && SMOLMC_EXP_VGPREF != 2
        prefetch_next_G(); // (at the end of the step)
#endif

        if (--smp_countdown == 0) {
            const LeanParamsKernarg Q = rare_params(); // (sampling parameters: see rare_params)
            const int qF = Q->F, qFce = Q->Fce;
            double *const q_feat = Q->smp.feat;
            smp_countdown = (uint32_t)Q->smp.every;
            const size_t rowi = (size_t)smp_index * Q->R + r;
            smp_index++;
            if (WLT) {
                if (lane < qF) q_feat[rowi * qF + lane] = fcur;
            } else {
            s_feat[lane] = 0.0;
#pragma unroll
            for (int it = 0; it < NSLOT; ++it)
                __hip_atomic_fetch_add(&s_feat[sfeat[it]], sfs[it] * acc[it], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (lane < qFce) q_feat[rowi * qF + lane] = base_feat + s_feat[lane];
            if (has_ew && lane == qFce) q_feat[rowi * qF + lane] = base_feat + acc_ew;
            if (has_mu && lane == qFce + (has_ew ? 1 : 0)) q_feat[rowi * qF + lane] = base_feat + acc_mu;
            }
            if (lane == 0) {
                Q->smp.H[rowi] = H;
                Q->smp.acc[rowi] = (uint8_t)last_acc;
                if (BIAS && Q->smp_bias_off) (Q->smp.H + Q->smp_bias_off)[rowi] = Q->bias[r] + tb_acc; // trace.bias
            }
            if (Q->smp.occ) {
                const int qNpad = Q->Npad;
                uint32_t *dst = (uint32_t *)(Q->smp.occ + rowi * qNpad);
                for (int i = lane; i < qNpad / 4; i += 64)
                    dst[i] = *(const uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb));
            }
        }
    }

#ifdef SMOLMC_EXP_PHASES
    if ((r == 0 || r == P.R / 2 || r == P.R - 1) && lane == 0)
        printf("phases (cycles per step): skeleton %.0f | head %.0f | picks %.0f | assign/swap %.0f | eval %.0f | decide %.0f\n",
               (double)ph_acc[0] / (double)P.steps, (double)ph_acc[1] / (double)P.steps, (double)ph_acc[2] / (double)P.steps,
               (double)ph_acc[3] / (double)P.steps, (double)ph_acc[4] / (double)P.steps, (double)ph_acc[5] / (double)P.steps);
    if ((r == 0 || r == P.R / 2 || r == P.R - 1) && lane == 0)
        printf("batch: covered %.3f of the steps, stale %.3f, swaps left out %.3f | batch %.0f cycles per step\n",
               (double)ph_cov[0] / (double)P.steps, (double)ph_cov[1] / (double)P.steps, (double)ph_cov[2] / (double)P.steps,
               (double)ph_bat / (double)P.steps);
    if ((r == 0 || r == P.R / 2 || r == P.R - 1) && lane == 0)
        printf("proposal: %.0f cycles per covered step, %.0f per step-at-a-time step; a-priori factor %.0f cycles per step\n",
               (double)ph_prop[0] / (double)(ph_cov[0] ? ph_cov[0] : 1), (double)ph_prop[1] / (double)(P.steps - ph_cov[0] ? P.steps - ph_cov[0] : 1),
               (double)ph_prop[2] / (double)P.steps);
#endif
    if (ew_field)
        for (int j = lane; j < P.ew_nact; j += 64) P.ew_phi[(size_t)r * P.ew_nact + j] = phi[j];
    {
        uint32_t *dst = (uint32_t *)(P.occ + (size_t)r * P.Npad);
        for (int i = lane; i < P.Npad / 4; i += 64)
            dst[i] = *(const uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb));
    }
    if (WLT) {
        if (wl_sum_mode) {
            wl_flush_run(); // the unfinished run of the current state
            wl_log_flush();
        } else {
            for (int slot = 0; slot < SMOLMC_WL_ROWS; ++slot) { // cached rows (running means) back to HBM
                const int tag = (int)rdlane((uint32_t)vtag, slot);
                if (tag >= 0 && lane < P.F)
                    P.wl.meanf[((size_t)r * P.wl.L + tag) * P.F + lane] = s_rows[(uint32_t)slot * (uint32_t)P.F + lane];
            }
        }
        for (int i = lane; i < P.wl.L; i += 64) {
            const size_t o = (size_t)r * P.wl.L + i;
            P.wl.entropy[o] = wl_S[i];
            P.wl.hist[o] += (long long)wl_cnt[i];
            P.wl.occur[o] += (long long)wl_cnt[i];
        }
        if (lane < P.F) featp[lane] = fcur;
        if (lane == 0) {
            P.wl.m[r] = wl_m;
            P.wl.counter[r] = wl_counter0 + (long long)(uint32_t)P.steps;
        }
    } else {
    s_feat[lane] = 0.0;
#pragma unroll
    for (int it = 0; it < NSLOT; ++it)
        __hip_atomic_fetch_add(&s_feat[sfeat[it]], sfs[it] * acc[it], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (lane < P.Fce) featp[lane] = base_feat + s_feat[lane];
    }
    if (lane == 0) {
        if (has_ew && !WLT) featp[P.Fce] += acc_ew;
        if (has_mu && !WLT) featp[P.Fce + (has_ew ? 1 : 0)] += acc_mu;
        P.enthalpy[r] = H;
        P.nsteps[r] = step;
        P.nacc[r] += nacc_add;
        P.last_acc[r] = (uint8_t)last_acc;
        if (BIAS && tb_type) {
            P.bias[r] += tb_acc;
#pragma unroll
            for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k)
                if (k < tb_rows) P.charge[(size_t)r * SMOLMC_MAX_BIAS_ROWS + k] = tb_chg[k];
        }
    }
}


#undef key0
#undef key1
template <int NSLOT, int MM, int STEP, bool MU, int EW, bool WL, bool BIAS = false, bool SOLO = false, int KF = 0,
          int OCC = 0, bool REPLAY = false>
static int launch_lean_inst(smolmc_handle *h, const LeanParams &lp) {
    const unsigned grid = SOLO ? (unsigned)h->R : (unsigned)((h->R + 3) / 4);
    auto kern = mc_lean_kernel<NSLOT, MM, STEP, MU, EW, WL, BIAS, SOLO, KF, OCC, REPLAY>;
    if (h->lean_lds > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h->lean_lds));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SOLO ? 64 : 256), h->lean_lds, h->stream, lp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return 0;
}
template <int NSLOT, int MM, int STEP>
static int launch_lean_me(smolmc_handle *h, const LeanParams &lp) {
    const bool mu = lp.mu_row != nullptr, ew = lp.ew_G != nullptr;
    if (h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU)
        return launch_lean_inst<NSLOT, MM, STEP, false, 0, true>(h, lp);
    if (ew)
        return mu ? (lp.ew_field ? launch_lean_inst<NSLOT, MM, STEP, true, 2, false>(h, lp) : launch_lean_inst<NSLOT, MM, STEP, true, 1, false>(h, lp))
                  : (lp.ew_field ? launch_lean_inst<NSLOT, MM, STEP, false, 2, false>(h, lp) : launch_lean_inst<NSLOT, MM, STEP, false, 1, false>(h, lp));
    if (h->lean_solo && h->lean_occ == 6)
        return mu ? launch_lean_inst<NSLOT, MM, STEP, true, 0, false, false, true, 0, 6>(h, lp)
                  : launch_lean_inst<NSLOT, MM, STEP, false, 0, false, false, true, 0, 6>(h, lp);
    if (h->lean_solo) // Metropolis without Ewald: one wave per workgroup (see mc_lean_kernel)
        return mu ? launch_lean_inst<NSLOT, MM, STEP, true, 0, false, false, true>(h, lp)
                  : launch_lean_inst<NSLOT, MM, STEP, false, 0, false, false, true>(h, lp);
    return mu ? launch_lean_inst<NSLOT, MM, STEP, true, 0, false>(h, lp)
              : launch_lean_inst<NSLOT, MM, STEP, false, 0, false>(h, lp);
}
template <int NSLOT, int MM>
static int launch_lean_nm(smolmc_handle *h, const LeanParams &lp) {
    if (h->cfg.step_type == SMOLMC_STEP_SWAP) return launch_lean_me<NSLOT, MM, SMOLMC_STEP_SWAP>(h, lp);
    return launch_lean_me<NSLOT, MM, SMOLMC_STEP_FLIP>(h, lp);
}
template <int NSLOT, int MM, int EWM, bool REPLAY = false, bool BIAS = false, bool WLT = false>
static int launch_table_ewm(smolmc_handle *h, const LeanParams &lp) {
    const int wpb = h->lean_wpb;
    const size_t lds = wpb == 8 ? h->lean_lds_wpb8 : h->lean_lds;
    const unsigned grid = (unsigned)((h->R + wpb - 1) / wpb);
    auto kern = mc_table_kernel<NSLOT, MM, EWM, REPLAY, BIAS, WLT>;
    if (lds > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wpb), lds, h->stream, lp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return 0;
}


template <int NSLOT, int MM, bool REPLAY = false, bool BIAS = false, bool WLT = false>
static int launch_table_inst(smolmc_handle *h, const LeanParams &lp) {
    if (lp.ew_G == nullptr) return launch_table_ewm<NSLOT, MM, 0, REPLAY, BIAS, WLT>(h, lp);
    if constexpr (WLT) return launch_table_ewm<NSLOT, MM, 2, REPLAY, BIAS, WLT>(h, lp); // (Wang-Landau: the Ewald term from the field only)
    else return lp.ew_field ? launch_table_ewm<NSLOT, MM, 2, REPLAY, BIAS>(h, lp) : launch_table_ewm<NSLOT, MM, 1, REPLAY, BIAS>(h, lp);
}
// (instantiated in table_wl_n*.hip only)
template <int NSLOT> static int launch_table_wl_nslot(smolmc_handle *h, const LeanParams &lp) {
    return h->lean_mm == 2 ? launch_table_inst<NSLOT, 2, false, false, true>(h, lp) : launch_table_inst<NSLOT, 3, false, false, true>(h, lp);
}
// (instantiated in table_bias_n*.hip only)
template <int NSLOT> static int launch_table_bias_nslot(smolmc_handle *h, const LeanParams &lp) {
    return h->lean_mm == 2 ? launch_table_inst<NSLOT, 2, false, true>(h, lp) : launch_table_inst<NSLOT, 3, false, true>(h, lp);
}
// (instantiated in table_replay_n*.hip only)
template <int NSLOT> static int launch_table_replay_nslot(smolmc_handle *h, const LeanParams &lp) {
    return h->lean_mm == 2 ? launch_table_inst<NSLOT, 2, true>(h, lp) : launch_table_inst<NSLOT, 3, true>(h, lp);
}

// biased Metropolis variants (instantiated in lean_bias_n*.hip only)
template <int NSLOT, int MM, int STEP>
static int launch_lean_bias_me(smolmc_handle *h, const LeanParams &lp) {
    const bool mu = lp.mu_row != nullptr, ew = lp.ew_G != nullptr;
    if (ew)
        return mu ? (lp.ew_field ? launch_lean_inst<NSLOT, MM, STEP, true, 2, false, true>(h, lp) : launch_lean_inst<NSLOT, MM, STEP, true, 1, false, true>(h, lp))
                  : (lp.ew_field ? launch_lean_inst<NSLOT, MM, STEP, false, 2, false, true>(h, lp) : launch_lean_inst<NSLOT, MM, STEP, false, 1, false, true>(h, lp));
    return mu ? launch_lean_inst<NSLOT, MM, STEP, true, 0, false, true>(h, lp)
              : launch_lean_inst<NSLOT, MM, STEP, false, 0, false, true>(h, lp);
}
template <int NSLOT> static int launch_lean_bias_nslot(smolmc_handle *h, const LeanParams &lp) {
    const bool swap = h->cfg.step_type == SMOLMC_STEP_SWAP;
    if (h->lean_mm == 2)
        return swap ? launch_lean_bias_me<NSLOT, 2, SMOLMC_STEP_SWAP>(h, lp)
                    : launch_lean_bias_me<NSLOT, 2, SMOLMC_STEP_FLIP>(h, lp);
    return swap ? launch_lean_bias_me<NSLOT, 3, SMOLMC_STEP_SWAP>(h, lp)
                : launch_lean_bias_me<NSLOT, 3, SMOLMC_STEP_FLIP>(h, lp);
}

// biased replay variants (instantiated in lean_bias_replay_n*.hip only)
template <int NSLOT, int MM, int STEP>
static int launch_lean_bias_replay_me(smolmc_handle *h, const LeanParams &lp) {
    const bool mu = lp.mu_row != nullptr, ew = lp.ew_G != nullptr;
    if (ew)
        return mu ? (lp.ew_field ? launch_lean_inst<NSLOT, MM, STEP, true, 2, false, true, false, 0, 0, true>(h, lp) : launch_lean_inst<NSLOT, MM, STEP, true, 1, false, true, false, 0, 0, true>(h, lp))
                  : (lp.ew_field ? launch_lean_inst<NSLOT, MM, STEP, false, 2, false, true, false, 0, 0, true>(h, lp) : launch_lean_inst<NSLOT, MM, STEP, false, 1, false, true, false, 0, 0, true>(h, lp));
    return mu ? launch_lean_inst<NSLOT, MM, STEP, true, 0, false, true, false, 0, 0, true>(h, lp)
              : launch_lean_inst<NSLOT, MM, STEP, false, 0, false, true, false, 0, 0, true>(h, lp);
}
template <int NSLOT> static int launch_lean_bias_replay_nslot(smolmc_handle *h, const LeanParams &lp) {
    const bool swap = h->cfg.step_type == SMOLMC_STEP_SWAP;
    if (h->lean_mm == 2)
        return swap ? launch_lean_bias_replay_me<NSLOT, 2, SMOLMC_STEP_SWAP>(h, lp)
                    : launch_lean_bias_replay_me<NSLOT, 2, SMOLMC_STEP_FLIP>(h, lp);
    return swap ? launch_lean_bias_replay_me<NSLOT, 3, SMOLMC_STEP_SWAP>(h, lp)
                : launch_lean_bias_replay_me<NSLOT, 3, SMOLMC_STEP_FLIP>(h, lp);
}

// correlation features with several functions per orbit (instantiated in lean_corr_n*.hip only)
template <int NSLOT, int MM, int STEP>
static int launch_lean_corr_me(smolmc_handle *h, const LeanParams &lp) {
    const bool mu = lp.mu_row != nullptr, ew = lp.ew_G != nullptr;
    constexpr int KF = SMOLMC_LEAN_MAX_KF;
    if (ew)
        return mu ? (lp.ew_field ? launch_lean_inst<NSLOT, MM, STEP, true, 2, false, false, false, KF>(h, lp) : launch_lean_inst<NSLOT, MM, STEP, true, 1, false, false, false, KF>(h, lp))
                  : (lp.ew_field ? launch_lean_inst<NSLOT, MM, STEP, false, 2, false, false, false, KF>(h, lp) : launch_lean_inst<NSLOT, MM, STEP, false, 1, false, false, false, KF>(h, lp));
    return mu ? launch_lean_inst<NSLOT, MM, STEP, true, 0, false, false, false, KF>(h, lp)
              : launch_lean_inst<NSLOT, MM, STEP, false, 0, false, false, false, KF>(h, lp);
}
template <int NSLOT> static int launch_lean_corr_nslot(smolmc_handle *h, const LeanParams &lp) {
    const bool swap = h->cfg.step_type == SMOLMC_STEP_SWAP;
    if (h->lean_mm == 2)
        return swap ? launch_lean_corr_me<NSLOT, 2, SMOLMC_STEP_SWAP>(h, lp)
                    : launch_lean_corr_me<NSLOT, 2, SMOLMC_STEP_FLIP>(h, lp);
    return swap ? launch_lean_corr_me<NSLOT, 3, SMOLMC_STEP_SWAP>(h, lp)
                : launch_lean_corr_me<NSLOT, 3, SMOLMC_STEP_FLIP>(h, lp);
}

template <int NSLOT> static int launch_lean_nslot(smolmc_handle *h, const LeanParams &lp) {
    if (h->cfg.step_type == SMOLMC_STEP_TABLE_FLIP)
        return h->lean_mm == 2 ? launch_table_inst<NSLOT, 2>(h, lp) : launch_table_inst<NSLOT, 3>(h, lp);
    return h->lean_mm == 2 ? launch_lean_nm<NSLOT, 2>(h, lp) : launch_lean_nm<NSLOT, 3>(h, lp);
}

// replay variants (instantiated in lean_replay_n*.hip only): the handle's own layout (SOLO or four
// walkers per workgroup), Ewald mode and mu row; correlation features with several functions per
// orbit take the KF variant
template <int NSLOT, int MM, int STEP, int KF>
static int launch_lean_replay_me(smolmc_handle *h, const LeanParams &lp) {
    const bool mu = lp.mu_row != nullptr, ew = lp.ew_G != nullptr;
    if (ew) {
        if (lp.ew_field)
            return mu ? launch_lean_inst<NSLOT, MM, STEP, true, 2, false, false, false, KF, 0, true>(h, lp)
                      : launch_lean_inst<NSLOT, MM, STEP, false, 2, false, false, false, KF, 0, true>(h, lp);
        return mu ? launch_lean_inst<NSLOT, MM, STEP, true, 1, false, false, false, KF, 0, true>(h, lp)
                  : launch_lean_inst<NSLOT, MM, STEP, false, 1, false, false, false, KF, 0, true>(h, lp);
    }
    if constexpr (KF == 0) {
        if (h->lean_solo)
            return mu ? launch_lean_inst<NSLOT, MM, STEP, true, 0, false, false, true, 0, 0, true>(h, lp)
                      : launch_lean_inst<NSLOT, MM, STEP, false, 0, false, false, true, 0, 0, true>(h, lp);
    }
    return mu ? launch_lean_inst<NSLOT, MM, STEP, true, 0, false, false, false, KF, 0, true>(h, lp)
              : launch_lean_inst<NSLOT, MM, STEP, false, 0, false, false, false, KF, 0, true>(h, lp);
}
template <int NSLOT> static int launch_lean_replay_nslot(smolmc_handle *h, const LeanParams &lp) {
    const bool swap = h->cfg.step_type == SMOLMC_STEP_SWAP;
    constexpr int K = SMOLMC_LEAN_MAX_KF;
    if (h->lean_kf) {
        if (h->lean_mm == 2)
            return swap ? launch_lean_replay_me<NSLOT, 2, SMOLMC_STEP_SWAP, K>(h, lp) : launch_lean_replay_me<NSLOT, 2, SMOLMC_STEP_FLIP, K>(h, lp);
        return swap ? launch_lean_replay_me<NSLOT, 3, SMOLMC_STEP_SWAP, K>(h, lp) : launch_lean_replay_me<NSLOT, 3, SMOLMC_STEP_FLIP, K>(h, lp);
    }
    if (h->lean_mm == 2)
        return swap ? launch_lean_replay_me<NSLOT, 2, SMOLMC_STEP_SWAP, 0>(h, lp) : launch_lean_replay_me<NSLOT, 2, SMOLMC_STEP_FLIP, 0>(h, lp);
    return swap ? launch_lean_replay_me<NSLOT, 3, SMOLMC_STEP_SWAP, 0>(h, lp) : launch_lean_replay_me<NSLOT, 3, SMOLMC_STEP_FLIP, 0>(h, lp);
}
