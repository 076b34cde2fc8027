// smolmc_hip.hip -- MI355X (gfx950) ensemble Monte-Carlo engine: HIP kernels + C-ABI.
//
// Hot path: per-flip local cluster-interaction / correlation delta
// (smol/utils/cluster/evaluator.pyx:211-317), Ewald single-flip delta
// (smol/utils/cluster/ewald.pyx:9-59), Metropolis / Wang-Landau accept
// (smol/moca/kernel/metropolis.py:31-49, wanglandau.py:186-266), ushers
// (smol/moca/kernel/mcusher.py:154-200), batched over independent walkers
// (smol/moca/sampler/sampler.py:195-208, :436-440).
//
// Design (DESIGN.md has the long form):
//   * one 64-lane wavefront owns one Markov chain for the whole launch; 4 chains
//     per 256-thread workgroup share the read-only tables staged in LDS;
//   * the chain's occupancy lives in LDS as one byte per site for the whole launch
//     (loaded / stored coalesced once per launch); per-site cluster-member index
//     rows stream coalesced from L2/HBM ([site][member][slot], slot = lane);
//   * every lane evaluates <= NSLOT clusters of the flipped site, the enthalpy
//     delta is a DPP wave reduction, the accept decision is wave-uniform;
//   * feature (trace) deltas are accumulated per lane in LDS and reduced once per
//     launch; Philox4x32-10 counter RNG generated 16 steps at a time across lanes.
// MFMA is not used: the work is sparse integer gathers + table lookups.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/smolmc.h"
#include "philox.h"

// ----------------------------------------------------------------------------
// error plumbing
// ----------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(const std::string &m) {
    g_err = m;
    return 1;
}
#define HIPCHK(x)                                                                         \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess)                                                             \
            return fail(std::string(#x) + ": " + hipGetErrorString(e_));                  \
    } while (0)

static const double SMOLMC_KB = 8.617333262145e-5; // smol/constants.py:4

// ----------------------------------------------------------------------------
// device-side parameter block
// ----------------------------------------------------------------------------
// device-side sample recording (Sampler.sample + SampleContainer.save_sampled_trace,
// sampler/sampler.py:195-210, container.py:384-397): every `every` steps one row per walker
struct SampleBufs {
    long long every;   // 0 = off
    double *H;         // [nsamples][R]
    double *feat;      // [nsamples][R][F]
    uint8_t *acc;      // [nsamples][R]
    uint8_t *occ;      // [nsamples][R][Npad] or null
};

struct KParams {
    // model geometry
    int N, Npad, Fce, F, nclasses, Cpad, Mmax, nsub, step_type;
    int has_ewald, has_mu, ew_W, ew_M, mu_W, corr_mode;
    // optimised MC tables
    const void *idx;              // IdxT [N][Mmax][Cpad]
    const uint8_t *site_class;    // [N] (255 = no clusters)
    const uint4 *descA;           // [nclasses][Cpad]  {xoff, u16 strides[6]}
    const uint4 *descB;           // [nclasses][Cpad]  {foff, tlen, feat|K<<16, 0}
    const double *slot_fs;        // [nclasses][Cpad]  feature scale size/(ratio*J)
    const int *cls_niter;         // [nclasses]
    const double *xt;             // decision tensors (per class), xt_len doubles
    const double *ft;             // feature tensors, ft_len doubles
    int xt_len, ft_len;
    // ewald / mu
    const int *ew_inds;           // [N][ew_W]
    const double *ew_Mt;          // [M][M] transposed ewald matrix
    double ew_coef;
    // compact Ewald (when the matrix factorises as q_a q_b G[site_a][site_b]):
    int ew_compact, ew_nact;      // ew_nact = number of sites whose species can change
    int ew_act_base;              // first such site when they are contiguous, else -1
    const int *ew_act;            // [ew_nact] those sites
    const double *ew_frozen;      // [N] sum over single-species sites k of q_k G[s][k]
    const double *ew_G;           // [N][ew_nact] site kernel restricted to changeable sites
    const double *ew_qs;          // [N][ew_W] charge of (site, code), 0 for vacancies
    const double *ew_dg;          // [N][ew_W] diagonal entry M[a][a] of (site, code)
    const double *mu;             // [N][mu_W]
    // sublattices
    const int *sub_ptr;           // [nsub+1]
    const int *sub_sites;         // concatenated active sites
    const int *sub_base;          // [nsub] first site if contiguous else -1
    const int *sub_code_ptr;      // [nsub+1]
    const int *sub_codes;
    const double *sub_cum;        // [nsub] cumulative probabilities
    // walker state
    int R;
    uint8_t *occ;                 // [R][Npad]
    double *enthalpy;             // [R]
    double *features;             // [R][F]
    const double *beta;           // [R]
    const uint64_t *seeds;        // [R]
    uint64_t *nsteps, *nacc;      // [R]
    uint8_t *last_acc;            // [R]
    long long steps_to_run;
    // replay
    const int *rp_steps;          // [R][nsteps][4]
    const double *rp_u;           // [R][nsteps]
    uint8_t *rp_acc;              // [R][nsteps]
    double *rp_H;                 // [R][nsteps]
    // Wang-Landau
    int L;
    double wl_min, wl_max, wl_bin, wl_flat, wl_div;
    long long wl_check, wl_update;
    double *wl_entropy;           // [R][L]
    long long *wl_hist;           // [R][L]
    long long *wl_occur;          // [R][L]
    double *wl_meanf;             // [R][L][F]
    double *wl_m;                 // [R]
    long long *wl_counter;        // [R]
    // lds layout (bytes)
    int lds_tables, lds_per_wave;
    SampleBufs smp;
};

// ----------------------------------------------------------------------------
// wave helpers
// ----------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
    // inclusive DPP scan inside each 16-lane row, then row broadcasts; lane 63 holds
    // the total, returned wave-uniform.
#define SMOLMC_DPP_STEP(ctrl, rmask)                                                       \
    {                                                                                      \
        int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rmask, 0xf, false); \
        int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rmask, 0xf, false); \
        v += __hiloint2double(hi_, lo_);                                                   \
    }
    SMOLMC_DPP_STEP(0x111, 0xf) // row_shr:1
    SMOLMC_DPP_STEP(0x112, 0xf) // row_shr:2
    SMOLMC_DPP_STEP(0x114, 0xf) // row_shr:4
    SMOLMC_DPP_STEP(0x118, 0xf) // row_shr:8
    SMOLMC_DPP_STEP(0x142, 0xa) // row_bcast:15 -> rows 1,3
    SMOLMC_DPP_STEP(0x143, 0xc) // row_bcast:31 -> rows 2,3
#undef SMOLMC_DPP_STEP
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
__device__ __forceinline__ double uni_d(double v) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// exact floor(x / y), y > 0 : Python's float // (wanglandau.py:180)
__device__ __forceinline__ double floordiv_exact(double x, double y) {
    double q = floor(x / y);
    double r = fma(-q, y, x);
    if (r < 0) q -= 1.0;
    else if (r >= y) q += 1.0;
    return q;
}

struct Lds {
    const uint4 *descA;
    const uint4 *descB;
    const double *slot_fs;
    const double *xt;
    const double *ft;
    const uint8_t *site_class;
    const int *cls_niter;
    uint8_t *occ;   // this wave's occupancy bytes
    double *acc;    // this wave's feature accumulators [Fce][64]  (Metropolis)
    double *wl_S;   // WL: entropy [L]
    long long *wl_H; // WL: histogram [L]
    double *wl_cf;  // WL: current features [F]
};

// u16 stride m out of the packed descriptor words
__device__ __forceinline__ int stride_of(const uint4 &a, int m) {
    uint32_t w = (m < 2) ? a.y : (m < 4 ? a.z : a.w);
    return (m & 1) ? (int)(w >> 16) : (int)(w & 0xffffu);
}

// Evaluate one cluster slot of a flip at site s (old code -> new code).
//   GENERIC: every member of the cluster row is gathered (the flipped site included)
//            -> exactly the reference index arithmetic, handles aliased rows.
//   !GENERIC: the row excludes the flipped site; its stride is st[0].
//   PATCH: occupancy seen is the LDS state with site ps overridden to pc (second
//          flip of a swap sees the first, processor/expansion.py:217-229).
template <typename IdxT, int MM, bool GENERIC, bool PATCH>
__device__ __forceinline__ double eval_slot(const KParams &P, const Lds &L, int cls, int c, int s,
                                            int oldc, int newc, int ps, int pc, int &ind_i,
                                            int &ind_f) {
    const uint4 a = L.descA[cls * P.Cpad + c];
    const IdxT *ip = (const IdxT *)P.idx + ((size_t)s * P.Mmax) * P.Cpad + c;
    int x[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) x[m] = (int)ip[(size_t)m * P.Cpad];
    int bi = 0, bf = 0;
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        int v = L.occ[x[m]];
        if (PATCH) v = (x[m] == ps) ? pc : v;
        if (GENERIC) {
            int st = stride_of(a, m);
            int vf = (x[m] == s) ? newc : v;
            bi += st * v;
            bf += st * vf;
        } else {
            bi += stride_of(a, m + 1) * v;
        }
    }
    if (!GENERIC) {
        int ss = stride_of(a, 0);
        bf = bi + ss * newc;
        bi = bi + ss * oldc;
    }
    ind_i = bi;
    ind_f = bf;
    return L.xt[a.x + bf] - L.xt[a.x + bi];
}

// feature accumulation of one accepted slot (Metropolis: lane-private LDS cells)
template <bool WL>
__device__ __forceinline__ void accum_slot(const KParams &P, const Lds &L, int cls, int c, int lane,
                                           int ind_i, int ind_f) {
    const uint4 b = L.descB[cls * P.Cpad + c];
    const int K = (int)(b.z >> 16), feat = (int)(b.z & 0xffffu);
    const double fs = L.slot_fs[cls * P.Cpad + c];
    for (int k = 0; k < K; ++k) {
        const double *t = L.ft + b.x + (size_t)k * b.y;
        double d = t[ind_f] - t[ind_i];
        if (WL) {
            // WL needs the reduced current features every step: LDS atomics
            __hip_atomic_fetch_add(&L.wl_cf[feat + k], fs * d, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else {
            double *cell = L.acc + (size_t)(feat + k) * 64 + lane;
            *cell = fma(fs, d, *cell);
        }
    }
}

// Ewald delta of one flip (ewald.pyx:38-58), wave-parallel over sites; returns the
// lane-partial (caller reduces).  Reads ROWS of the transposed matrix, i.e. the same
// entries M[i, add] / M[j, sub] the reference reads as columns.
template <bool PATCH>
__device__ __forceinline__ double ewald_partial(const KParams &P, const Lds &L, int lane, int s,
                                                int oldc, int newc, int ps, int pc) {
    const int W = P.ew_W;
    const int add = P.ew_inds[(size_t)s * W + newc];
    const int sub = P.ew_inds[(size_t)s * W + oldc];
    const double *radd = P.ew_Mt + (size_t)(add < 0 ? 0 : add) * P.ew_M;
    const double *rsub = P.ew_Mt + (size_t)(sub < 0 ? 0 : sub) * P.ew_M;
    double out = 0;
    for (int k = lane; k < P.N; k += 64) {
        int v = L.occ[k];
        if (PATCH) v = (k == ps) ? pc : v;
        int vf = (k == s) ? newc : v;
        int i = P.ew_inds[(size_t)k * W + vf];
        int j = (k == s) ? P.ew_inds[(size_t)k * W + v] : i;
        double o = 0;
        if (i != -1 && add != -1) o += (i != add ? 2.0 : 1.0) * radd[i];
        if (j != -1 && sub != -1) o -= (j != sub ? 2.0 : 1.0) * rsub[j];
        out += o;
    }
    return out;
}

// Compact form of the same delta when M[a][b] = q_a q_b G[site_a][site_b] (a != b):
//   sum_k [2 M[i_k, add] - 2 M[i_k, sub]]  (k != s, i_k = j_k)  + M[add,add] - M[sub,sub]
//     = 2 (q_add - q_sub) * sum_{k != s} q(k, occ_k) G[s][k] + diag(add) - diag(sub)
// One fully-used row of G (N x 8 B) streams per flip instead of two strided matrix rows.
template <bool PATCH>
__device__ __forceinline__ double ewald_compact_partial(const KParams &P, const Lds &L, int lane, int s,
                                                        int ps, int pc) {
    // sites with a single allowed species never change: their part of the sum is the
    // precomputed ew_frozen[s]; only the changeable sites are streamed
    const double *g = P.ew_G + (size_t)s * P.ew_nact;
    const int W = P.ew_W, abase = P.ew_act_base;
    double out = 0;
#pragma unroll 4
    for (int j = lane; j < P.ew_nact; j += 64) {
        const int k = abase >= 0 ? abase + j : P.ew_act[j];
        int v = L.occ[k];
        if (PATCH) v = (k == ps) ? pc : v;
        const double q = P.ew_qs[(size_t)k * W + v];
        out = fma(k == s ? 0.0 : q, g[j], out);
    }
    return out;
}

// ----------------------------------------------------------------------------
// the Monte-Carlo kernel
// ----------------------------------------------------------------------------
template <typename IdxT, int NSLOT, int MM, bool GENERIC, bool WL>
__global__ void __launch_bounds__(256) mc_kernel(const KParams P, const int replay) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int r = uni(blockIdx.x * nwaves + wave);

    // ---- stage read-only tables in LDS (shared by the workgroup) -------------
    unsigned char *sp = smem;
    uint4 *s_descA = (uint4 *)sp;               sp += (size_t)P.nclasses * P.Cpad * 16;
    uint4 *s_descB = (uint4 *)sp;               sp += (size_t)P.nclasses * P.Cpad * 16;
    double *s_fs = (double *)sp;                sp += (size_t)P.nclasses * P.Cpad * 8;
    double *s_xt = (double *)sp;                sp += (size_t)P.xt_len * 8;
    double *s_ft = (double *)sp;                sp += (size_t)P.ft_len * 8;
    int *s_niter = (int *)sp;                   sp += (size_t)((P.nclasses + 3) & ~3) * 4;
    uint8_t *s_cls = (uint8_t *)sp;
    for (int i = threadIdx.x; i < P.nclasses * P.Cpad; i += blockDim.x) {
        s_descA[i] = P.descA[i];
        s_descB[i] = P.descB[i];
        s_fs[i] = P.slot_fs[i];
    }
    for (int i = threadIdx.x; i < P.xt_len; i += blockDim.x) s_xt[i] = P.xt[i];
    for (int i = threadIdx.x; i < P.ft_len; i += blockDim.x) s_ft[i] = P.ft[i];
    for (int i = threadIdx.x; i < P.nclasses; i += blockDim.x) s_niter[i] = P.cls_niter[i];
    if (P.nclasses > 1)
        for (int i = threadIdx.x; i < P.N; i += blockDim.x) s_cls[i] = P.site_class[i];

    Lds L;
    L.descA = s_descA; L.descB = s_descB; L.slot_fs = s_fs; L.xt = s_xt; L.ft = s_ft;
    L.site_class = s_cls; L.cls_niter = s_niter;
    unsigned char *wp = smem + P.lds_tables + (size_t)wave * P.lds_per_wave;
    L.occ = wp;
    wp += P.Npad;
    L.acc = nullptr; L.wl_S = nullptr; L.wl_H = nullptr; L.wl_cf = nullptr;
    if (WL) {
        L.wl_S = (double *)wp;        wp += (size_t)P.L * 8;
        L.wl_H = (long long *)wp;     wp += (size_t)P.L * 8;
        L.wl_cf = (double *)wp;
    } else {
        L.acc = (double *)wp;
    }

    // ---- this wave's chain: occupancy -> LDS (coalesced 16-byte loads) --------
    const bool live = r < P.R;
    if (live) {
        const uint4 *src = (const uint4 *)(P.occ + (size_t)r * P.Npad);
        uint4 *dst = (uint4 *)L.occ;
        for (int i = lane; i < P.Npad / 16; i += 64) dst[i] = src[i];
        if (WL) {
            for (int i = lane; i < P.L; i += 64) {
                L.wl_S[i] = P.wl_entropy[(size_t)r * P.L + i];
                L.wl_H[i] = P.wl_hist[(size_t)r * P.L + i];
            }
            for (int i = lane; i < P.F; i += 64) L.wl_cf[i] = P.features[(size_t)r * P.F + i];
        } else {
            for (int i = lane; i < P.Fce * 64; i += 64) L.acc[i] = 0.0;
        }
    }
    __syncthreads();
    if (!live) return;

    // ---- chain registers ------------------------------------------------------
    double H = P.enthalpy[r];
    const double beta = WL ? 0.0 : P.beta[r];
    unsigned long long step = P.nsteps[r];
    unsigned long long nacc = P.nacc[r];
    const uint32_t key0 = (uint32_t)P.seeds[r], key1 = (uint32_t)(P.seeds[r] >> 32);
    double acc_ew = 0.0, acc_mu = 0.0; // Ewald / chemical-work feature deltas (uniform)
    int last_acc = 1;
    double wl_m = 0.0;
    long long wl_counter = 0;
    if (WL) {
        wl_m = P.wl_m[r];
        wl_counter = P.wl_counter[r];
    }
    uint32_t W0 = 0, W1 = 0, W2 = 0, W3 = 0; // RNG batch: lane l = block (l&3) of step base+(l>>2)
    uint32_t w_site_carry = 0;
    unsigned long long batch_base = ~0ull - 64ull;
    long long smp_countdown = P.smp.every, smp_index = 0;

    for (long long it_step = 0; it_step < P.steps_to_run; ++it_step, ++step) {
        // ================= proposal =========================================
        int nfl = 0, s1 = 0, n1 = 0, o1 = 0, s2 = 0, n2 = 0, o2 = 0;
        double u = 0.0;
        if (replay) {
            const int *st = P.rp_steps + ((size_t)r * P.steps_to_run + it_step) * 4;
            int a0 = st[0], a1 = st[1], a2 = st[2], a3 = st[3];
            u = P.rp_u[(size_t)r * P.steps_to_run + it_step];
            if (u != u) u = 0.0; // NaN: the reference accepted without drawing
            a0 = uni(a0); a1 = uni(a1); a2 = uni(a2); a3 = uni(a3);
            u = uni_d(u);
            if (a0 >= 0) { nfl = 1; s1 = a0; n1 = a1; o1 = uni((int)L.occ[s1]); }
            if (a2 >= 0) { nfl = 2; s2 = a2; n2 = a3; o2 = uni((int)L.occ[s2]); if (s2 == s1) o2 = n1; }
        } else {
            const unsigned long long base = step & ~15ull;
            if (base != batch_base) {
                // the site word of a step comes from the PREVIOUS step's block 0 (word 1)
                if (batch_base == base - 16) {
                    w_site_carry = rdlane(W1, 60);
                } else {
                    const unsigned long long sp = base - 1ull;
                    w_site_carry = uni((int)philox4x32_10((uint32_t)sp, (uint32_t)(sp >> 32), 0u, 0u,
                                                          key0, key1).w[1]);
                }
                batch_base = base;
                unsigned long long st = base + (unsigned)(lane >> 2);
                philox_out o = philox4x32_10((uint32_t)st, (uint32_t)(st >> 32), (uint32_t)(lane & 3),
                                             0u, key0, key1);
                W0 = o.w[0]; W1 = o.w[1]; W2 = o.w[2]; W3 = o.w[3];
            }
            const int l4 = (int)(step & 15ull) * 4;
            const uint32_t w_sub = rdlane(W0, l4);
            const uint32_t w_site = l4 == 0 ? w_site_carry : rdlane(W1, l4 - 4);
            u = philox_u53(rdlane(W2, l4), rdlane(W3, l4));
            // sublattice: MCUsher.get_random_sublattice (mcusher.py:146-148)
            int sl = 0;
            if (P.nsub > 1) {
                const double x = (double)w_sub * (1.0 / 4294967296.0);
                sl = P.nsub - 1;
                for (int q = P.nsub - 2; q >= 0; --q)
                    if (x < P.sub_cum[q]) sl = q;
            }
            const int p0 = P.sub_ptr[sl];
            const uint32_t nact = (uint32_t)(P.sub_ptr[sl + 1] - p0);
            const int sbase = P.sub_base[sl];
            const uint32_t k1 = __umulhi(w_site, nact);
            s1 = sbase >= 0 ? sbase + (int)k1 : P.sub_sites[p0 + k1];
            s1 = uni(s1);
            o1 = uni((int)L.occ[s1]);
            if (P.step_type == SMOLMC_STEP_FLIP) {
                // Flip.propose_step (mcusher.py:154-170)
                const int c0 = P.sub_code_ptr[sl];
                const uint32_t nc = (uint32_t)(P.sub_code_ptr[sl + 1] - c0);
                const uint32_t kk = __umulhi(rdlane(W0, l4 + 1), nc - 1);
                int code = -1;
                uint32_t seen = 0;
                for (uint32_t c = 0; c < nc; ++c) {
                    int cc = P.sub_codes[c0 + c];
                    if (cc == o1) continue;
                    if (seen == kk && code < 0) code = cc;
                    seen++;
                }
                n1 = uni(code);
                nfl = 1;
            } else {
                // Swap.propose_step (mcusher.py:176-200) by rejection over the candidate
                // sequence ('random stream')
                int found = -1;
                {
                    // first 12 candidates c_t = W(step, 1 + t % 3, t / 3): word-major over the
                    // three candidate lanes of this step
                    const uint32_t ws[4] = {W0, W1, W2, W3};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (found < 0) {
                            const uint32_t kc = __umulhi(ws[j], nact);
                            const int cs = sbase >= 0 ? sbase + (int)kc : P.sub_sites[p0 + kc];
                            const bool hit = (int)L.occ[cs] != o1;
                            unsigned long long m = __ballot(hit) & (0xEull << l4);
                            if (m) found = (int)rdlane((uint32_t)cs, __ffsll((long long)m) - 1);
                        }
                    }
                }
                if (found < 0) {
                    // rare: continue the candidate sequence with blocks 4 + 64 q + lane
                    for (uint32_t q = 0;; ++q) {
                        philox_out o = philox4x32_10((uint32_t)step, (uint32_t)(step >> 32),
                                                     4u + 64u * q + (uint32_t)lane, 0u, key0, key1);
                        int selsite = -1;
#pragma unroll
                        for (int j = 3; j >= 0; --j) {
                            const uint32_t kc = __umulhi(o.w[j], nact);
                            const int cs = sbase >= 0 ? sbase + (int)kc : P.sub_sites[p0 + kc];
                            if ((int)L.occ[cs] != o1) selsite = cs;
                        }
                        unsigned long long m = __ballot(selsite >= 0);
                        if (m) {
                            found = (int)rdlane((uint32_t)selsite, __ffsll((long long)m) - 1);
                            break;
                        }
                        if ((q & 63u) == 0) { // swap_options.size == 0 -> empty step (:197-199)
                            int any = 0;
                            for (uint32_t a = lane; a < nact; a += 64) {
                                const int cs = sbase >= 0 ? sbase + (int)a : P.sub_sites[p0 + a];
                                any |= ((int)L.occ[cs] != o1);
                            }
                            if (__ballot(any) == 0ull) break;
                        }
                    }
                }
                if (found >= 0) {
                    s2 = uni(found);
                    o2 = uni((int)L.occ[s2]);
                    n1 = o2;
                    n2 = o1;
                    nfl = 2;
                }
            }
        }

        // ================= enthalpy delta ====================================
        int ii1[NSLOT], jf1[NSLOT], ii2[NSLOT], jf2[NSLOT];
        int cls1 = 0, cls2 = 0, nit1 = 0, nit2 = 0;
        double e = 0.0;
        if (nfl >= 1) {
            cls1 = P.nclasses > 1 ? uni((int)L.site_class[s1]) : 0;
            nit1 = cls1 == 255 ? 0 : uni(L.cls_niter[cls1]);
        }
        if (nfl == 2) {
            cls2 = P.nclasses > 1 ? uni((int)L.site_class[s2]) : 0;
            nit2 = cls2 == 255 ? 0 : uni(L.cls_niter[cls2]);
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) {
                const int c = lane + 64 * it;
                if (it < nit1)
                    e += eval_slot<IdxT, MM, GENERIC, false>(P, L, cls1, c, s1, o1, n1, 0, 0, ii1[it],
                                                             jf1[it]);
                if (it < nit2)
                    e += eval_slot<IdxT, MM, GENERIC, true>(P, L, cls2, c, s2, o2, n2, s1, n1, ii2[it],
                                                            jf2[it]);
            }
        } else if (nfl == 1) {
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) {
                const int c = lane + 64 * it;
                if (it < nit1)
                    e += eval_slot<IdxT, MM, GENERIC, false>(P, L, cls1, c, s1, o1, n1, 0, 0, ii1[it],
                                                             jf1[it]);
            }
        }
        double dEw = 0.0, dMu = 0.0;
        if (P.has_ewald && nfl >= 1) {
            if (P.ew_compact) {
                const int W = P.ew_W;
                const double s1sum = P.ew_frozen[s1] + wave_sum(ewald_compact_partial<false>(P, L, lane, s1, 0, 0));
                dEw = 2.0 * (P.ew_qs[(size_t)s1 * W + n1] - P.ew_qs[(size_t)s1 * W + o1]) * s1sum +
                      (P.ew_dg[(size_t)s1 * W + n1] - P.ew_dg[(size_t)s1 * W + o1]);
                if (nfl == 2) {
                    const double s2sum = P.ew_frozen[s2] + wave_sum(ewald_compact_partial<true>(P, L, lane, s2, s1, n1));
                    dEw += 2.0 * (P.ew_qs[(size_t)s2 * W + n2] - P.ew_qs[(size_t)s2 * W + o2]) * s2sum +
                           (P.ew_dg[(size_t)s2 * W + n2] - P.ew_dg[(size_t)s2 * W + o2]);
                }
                dEw = uni_d(dEw);
            } else {
                double pe = ewald_partial<false>(P, L, lane, s1, o1, n1, 0, 0);
                if (nfl == 2) pe += ewald_partial<true>(P, L, lane, s2, o2, n2, s1, n1);
                dEw = wave_sum(pe);
            }
        }
        if (P.has_mu && nfl >= 1) {
            // delta chemical work against the ORIGINAL occupancy (ensemble.py:368-374)
            dMu = P.mu[(size_t)s1 * P.mu_W + n1] - P.mu[(size_t)s1 * P.mu_W + o1];
            if (nfl == 2) {
                const int orig2 = uni((int)L.occ[s2]);
                dMu += P.mu[(size_t)s2 * P.mu_W + n2] - P.mu[(size_t)s2 * P.mu_W + orig2];
            }
            dMu = uni_d(dMu);
        }
        double dH = wave_sum(e);
        if (P.has_ewald) dH += P.ew_coef * dEw;
        if (P.has_mu) dH -= dMu;

        // ================= accept ==============================================
        bool accepted;
        if (!WL) {
            // MetropolisAcceptMixin._accept_step (metropolis.py:31-49)
            const double exponent = -beta * dH + 0.0;
            accepted = exponent >= 0.0 ? true : (exponent > log(u));
        } else {
            // WangLandau._accept_step (wanglandau.py:186-202)
            const double new_h = H + dH;
            if (new_h < P.wl_min || new_h >= P.wl_max) {
                accepted = false;
            } else {
                const int b = (int)floordiv_exact(H - P.wl_min, P.wl_bin);
                const int nb = (int)floordiv_exact(new_h - P.wl_min, P.wl_bin);
                const double exponent = L.wl_S[b] - L.wl_S[nb] + 0.0;
                accepted = exponent >= 0.0 ? true : (exponent > log(u));
            }
        }

        // ================= update ==============================================
        if (accepted) {
            // MCKernel._do_accept_step (kernel/base.py:327-343) + trace += delta
            // (sampler/sampler.py:204-207)
            if (nfl >= 1) {
#pragma unroll
                for (int it = 0; it < NSLOT; ++it)
                    if (it < nit1) accum_slot<WL>(P, L, cls1, lane + 64 * it, lane, ii1[it], jf1[it]);
            }
            if (nfl == 2) {
#pragma unroll
                for (int it = 0; it < NSLOT; ++it)
                    if (it < nit2) accum_slot<WL>(P, L, cls2, lane + 64 * it, lane, ii2[it], jf2[it]);
            }
            if (lane == 0) {
                if (nfl >= 1) L.occ[s1] = (uint8_t)n1;
                if (nfl == 2) L.occ[s2] = (uint8_t)n2;
                if (WL) {
                    if (P.has_ewald) L.wl_cf[P.Fce] += dEw;
                    if (P.has_mu) L.wl_cf[P.Fce + P.has_ewald] += dMu;
                }
            }
            acc_ew += dEw;
            acc_mu += dMu;
            H += dH;
            nacc++;
        }
        last_acc = accepted ? 1 : 0;

        if (WL) {
            // WangLandau._do_post_step (wanglandau.py:222-266)
            const double bq = floordiv_exact(H - P.wl_min, P.wl_bin);
            if (bq >= 0.0 && bq < (double)P.L) {
                const int b = (int)bq;
                wl_counter++;
                const size_t cell = (size_t)r * P.L + b;
                // lane 0 owns the occurrences counter (single-thread program order for
                // its own global read-after-write); broadcast to the wave
                long long total = 0;
                if (lane == 0) total = P.wl_occur[cell];
                total = ((long long)(unsigned)uni((int)(total >> 32)) << 32) |
                        (unsigned)uni((int)(total & 0xffffffffll));
                if (lane < P.F) {
                    double *mf = P.wl_meanf + cell * P.F + lane;
                    const double inv = 1.0 / (double)(total + 1);
                    *mf = inv * (L.wl_cf[lane] + (double)total * (*mf));
                }
                if (wl_counter % P.wl_update == 0) {
                    if (lane == 0) {
                        L.wl_S[b] += wl_m;
                        L.wl_H[b] += 1;
                        P.wl_occur[cell] = total + 1;
                    }
                }
            }
            if (wl_counter % P.wl_check == 0) {
                long cnt = 0;
                double sum = 0;
                for (int i = lane; i < P.L; i += 64)
                    if (L.wl_S[i] > 0) { cnt++; sum += (double)L.wl_H[i]; }
                const double tcnt = wave_sum((double)cnt), tsum = wave_sum(sum);
                if (tcnt >= 2.0) {
                    const double thr = P.wl_flat * (tsum / tcnt);
                    int bad = 0;
                    for (int i = lane; i < P.L; i += 64)
                        if (L.wl_S[i] > 0 && !((double)L.wl_H[i] > thr)) bad = 1;
                    if (__ballot(bad) == 0ull) {
                        for (int i = lane; i < P.L; i += 64) L.wl_H[i] = 0;
                        wl_m = wl_m / P.wl_div;
                    }
                }
            }
        }
        if (replay && lane == 0) {
            if (P.rp_acc) P.rp_acc[(size_t)r * P.steps_to_run + it_step] = (uint8_t)last_acc;
            if (P.rp_H) P.rp_H[(size_t)r * P.steps_to_run + it_step] = H;
        }
        if (P.smp.every && --smp_countdown == 0) { // record one thinned sample of this walker
            smp_countdown = P.smp.every;
            const size_t row = (size_t)smp_index * P.R + r;
            smp_index++;
            double *dstf = P.smp.feat + row * P.F;
            const double *base = P.features + (size_t)r * P.F;
            if (WL) {
                for (int i = lane; i < P.F; i += 64) dstf[i] = L.wl_cf[i];
            } else {
                for (int f = 0; f < P.Fce; ++f) {
                    const double sm = wave_sum(L.acc[(size_t)f * 64 + lane]);
                    if (lane == 0) dstf[f] = base[f] + sm;
                }
                if (lane == 0) {
                    if (P.has_ewald) dstf[P.Fce] = base[P.Fce] + acc_ew;
                    if (P.has_mu) dstf[P.Fce + P.has_ewald] = base[P.Fce + P.has_ewald] + acc_mu;
                }
            }
            if (lane == 0) {
                P.smp.H[row] = H;
                P.smp.acc[row] = (uint8_t)last_acc;
            }
            if (P.smp.occ) {
                uint4 *dst = (uint4 *)(P.smp.occ + row * P.Npad);
                const uint4 *src = (const uint4 *)L.occ;
                for (int i = lane; i < P.Npad / 16; i += 64) dst[i] = src[i];
            }
        }
    }

    // ---- write the chain back --------------------------------------------------
    {
        uint4 *dst = (uint4 *)(P.occ + (size_t)r * P.Npad);
        const uint4 *src = (const uint4 *)L.occ;
        for (int i = lane; i < P.Npad / 16; i += 64) dst[i] = src[i];
    }
    double *feat = P.features + (size_t)r * P.F;
    if (WL) {
        for (int i = lane; i < P.L; i += 64) {
            P.wl_entropy[(size_t)r * P.L + i] = L.wl_S[i];
            P.wl_hist[(size_t)r * P.L + i] = L.wl_H[i];
        }
        for (int i = lane; i < P.F; i += 64) feat[i] = L.wl_cf[i];
        if (lane == 0) {
            P.wl_m[r] = wl_m;
            P.wl_counter[r] = wl_counter;
        }
    } else {
        for (int f = 0; f < P.Fce; ++f) {
            const double s = wave_sum(L.acc[(size_t)f * 64 + lane]);
            if (lane == 0) feat[f] += s;
        }
        if (lane == 0) {
            if (P.has_ewald) feat[P.Fce] += acc_ew;
            if (P.has_mu) feat[P.Fce + P.has_ewald] += acc_mu;
        }
    }
    if (lane == 0) {
        P.enthalpy[r] = H;
        P.nsteps[r] = step;
        P.nacc[r] = nacc;
        P.last_acc[r] = (uint8_t)last_acc;
    }
}


// ----------------------------------------------------------------------------
// lean Metropolis kernel: one site class, one contiguous active sublattice with the
// default encoding, cluster-interaction features, no Ewald term, engine RNG.
// This is the shape of BASELINE configs 1/2/4; everything else takes mc_kernel.
//   * member index rows are lane-packed: idx[site][lane][NSLOT][MM] (u16), one vector
//     load per flip;
//   * per-(orbit, self position) DELTA tables dt[(old*S+new)][base] = T[.. new ..] - T[.. old ..]
//     live in LDS: one 8-byte LDS read per cluster instead of two reads + a subtract
//     (the subtraction is done once on the host in float64: identical value);
//   * slot constants and feature accumulators stay in registers for the whole launch.
// ----------------------------------------------------------------------------
struct WlParams { // Wang-Landau state of the walkers (kernel/wanglandau.py:107-122)
    int L;
    double vmin, vmax, bin, flat, div;
    long long check, update;
    double *entropy;     // [R][L]
    long long *hist;     // [R][L]
    long long *occur;    // [R][L]
    double *meanf;       // [R][L][F]
    double *m;           // [R]
    long long *counter;  // [R]
};

struct LeanSlot {
    uint32_t doff8;      // byte offset of the slot's delta table
    uint32_t stride8[3]; // 8 * stride of the other members
    uint32_t feat;       // feature index (orbit id)
    uint32_t live;       // 0 for padded slots
    double w;            // natural parameter * size / (ratio * J)
    double fs;           // size / (ratio * J)
};

struct LeanParams {
    const uint16_t *idx;   // [N][64][NSLOT][MM]
    const double *dt;      // delta tables, all padded to a common [S*S][NTP] shape
    const LeanSlot *slots; // [NSLOT][64]
    const double *mu_row;  // [ncodes] chemical potentials of the active sublattice (or null)
    uint8_t *occ;
    double *enthalpy, *features;
    const double *beta;
    const uint64_t *seeds;
    uint64_t *nsteps, *nacc;
    uint8_t *last_acc;
    int dt_len, R, N, Npad, F, Fce, sbase, nact, ncodes;
    uint32_t nt8, snt8;    // 8*NTP and 8*NTP*S: (old, new) -> byte offset old*snt8 + new*nt8
    // LDS address of site s = s ^ (((s >> swz_a) & swz_m) << swz_b): a bank swizzle chosen on
    // the host (bank-conflict model over the cluster tables); idx rows hold swizzled addresses
    int swz_a, swz_m, swz_b, Nlds;
    long long steps;
    SampleBufs smp;
    // compact Ewald term (see build_compact_ewald); feature index Fce, coefficient ew_coef
    int ew_W, ew_nact, ew_act_base;
    const int *ew_act;
    const double *ew_G, *ew_qs, *ew_dg, *ew_frozen;
    double ew_coef;
    WlParams wl;
    // TableFlip (mcusher.py:397-711) for the single active sublattice
    int tf_n;               // number of flip vectors
    const int *tf_table;    // [tf_n][ncodes]
    const double *tf_w;     // [2 tf_n]
    double tf_sw;           // swap_weight
};

__device__ __forceinline__ int lean_swz(int s, int a, int m, int b) { return s ^ (((s >> a) & m) << b); }

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

template <int NSLOT, int MM, int STEP, bool HAS_MU, bool HAS_EW, bool WL>
__global__ void __launch_bounds__(256) mc_lean_kernel(const LeanParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int r = uni(blockIdx.x * nwaves + wave);
    double *s_dt = (double *)smem;
    double *s_mu = s_dt + P.dt_len;               // 8 doubles
    const size_t per_wave = (size_t)P.Nlds + 64 * 8 + (WL ? (size_t)P.wl.L * 16 : 0);
    unsigned char *wbase = (unsigned char *)(s_mu + 8) + (size_t)wave * per_wave;
    uint8_t *occ = wbase;                         // indexed by SWIZZLED site address
    // Metropolis: scratch for the feature reduction; Wang-Landau: the CURRENT features
    // (wanglandau.py:216-218 needs them every step for the per-bin running mean)
    double *s_feat = (double *)(wbase + P.Nlds);
    double *wl_S = s_feat + 64;                   // WL: entropy [L]
    long long *wl_Hh = (long long *)(wl_S + (WL ? P.wl.L : 0)); // WL: histogram [L]
    const int swa = P.swz_a, swm = P.swz_m, swb = P.swz_b;
    for (int i = threadIdx.x; i < P.dt_len; i += blockDim.x) s_dt[i] = P.dt[i];
    if (HAS_MU && threadIdx.x < 8) s_mu[threadIdx.x] = threadIdx.x < P.ncodes ? P.mu_row[threadIdx.x] : 0.0;
    const bool live = r < P.R;
    if (live) {
        // the swizzle only touches address bits >= 2: move whole dwords
        const uint32_t *src = (const uint32_t *)(P.occ + (size_t)r * P.Npad);
        for (int i = lane; i < P.Npad / 4; i += 64)
            *(uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb)) = src[i];
        s_feat[lane] = (WL && lane < P.F) ? P.features[(size_t)r * P.F + lane] : 0.0;
        if (WL)
            for (int i = lane; i < P.wl.L; i += 64) {
                wl_S[i] = P.wl.entropy[(size_t)r * P.wl.L + i];
                wl_Hh[i] = P.wl.hist[(size_t)r * P.wl.L + i];
            }
    }
    __syncthreads();
    if (!live) return;

    // per-lane slot constants (registers for the whole launch)
    uint32_t doff8[NSLOT], st8[NSLOT][MM], sfeat[NSLOT];
    double wgt[NSLOT], acc[NSLOT], sfs[NSLOT];
#pragma unroll
    for (int it = 0; it < NSLOT; ++it) {
        const LeanSlot sl = P.slots[it * 64 + lane];
        doff8[it] = sl.doff8;
        sfeat[it] = sl.feat;      // only used by the Wang-Landau variant
        sfs[it] = sl.live ? sl.fs : 0.0;
#pragma unroll
        for (int m = 0; m < MM; ++m) st8[it][m] = sl.stride8[m];
        wgt[it] = sl.w;
        acc[it] = 0.0;
    }
    double H = P.enthalpy[r];
    const double nbeta = WL ? 0.0 : -P.beta[r];
    double wl_m = WL ? P.wl.m[r] : 0.0;
    long long wl_counter = WL ? P.wl.counter[r] : 0;
    unsigned long long step = P.nsteps[r];
    unsigned long long nacc = P.nacc[r];
    const uint32_t key0 = (uint32_t)P.seeds[r], key1 = (uint32_t)(P.seeds[r] >> 32);
    const uint32_t nact = (uint32_t)P.nact, nt8 = P.nt8, snt8 = P.snt8;
    const int sbase = P.sbase;
    double acc_mu = 0.0, acc_ew = 0.0;
    int last_acc = 1;
    // trace at launch start; features of a sample = base + sum over lanes of fs * acc
    double *featp = P.features + (size_t)r * P.F;
    const double base_feat = lane < P.F ? featp[lane] : 0.0;
    long long smp_countdown = P.smp.every, smp_index = 0;
    // random batch: lane l holds block (l & 3) of step batch_base + (l >> 2)
    uint32_t W0 = 0, W1 = 0;
    int cand[4] = {0, 0, 0, 0}, canda[4] = {0, 0, 0, 0}; // candidate sites / their LDS addresses
    double logu = 0.0; // log of the acceptance uniform of the lane's step (block-0 lanes)
    unsigned long long batch_base = ~0ull;
    constexpr int ROW = NSLOT * MM; // u16 entries per lane per site
    const uint16_t *idx_lane = P.idx + (size_t)lane * ROW;

    // software pipeline: the site of step k comes from W(k-1, 0, 1), so the index row of
    // the NEXT step is always known one step ahead and is fetched while this step runs.
    int s1;
    uint16_t row1[ROW];
    {
        const unsigned long long sp = step - 1ull;
        const uint32_t w = (uint32_t)uni((int)philox4x32_10((uint32_t)sp, (uint32_t)(sp >> 32), 0u, 0u,
                                                            key0, key1).w[1]);
        s1 = sbase + (int)__umulhi(w, nact);
        const uint16_t *p = idx_lane + (size_t)s1 * (64 * ROW);
#pragma unroll
        for (int q = 0; q < ROW; ++q) row1[q] = p[q];
    }

    for (long long it_step = 0; it_step < P.steps; ++it_step, ++step) {
        // -------- random words of this step (generated 16 steps at a time) --------
        const unsigned long long base = step & ~15ull;
        if (base != batch_base) {
            batch_base = base;
            const unsigned long long st = base + (unsigned)(lane >> 2);
            const philox_out o = philox4x32_10((uint32_t)st, (uint32_t)(st >> 32), (uint32_t)(lane & 3),
                                               0u, key0, key1);
            W0 = o.w[0]; W1 = o.w[1];
            // metropolis.py:46-48 compares the exponent with log(rng.random()): take the
            // float64 log of all 16 uniforms of the batch at once (lane-parallel)
            logu = log(philox_u53(o.w[2], o.w[3]));
            if (STEP == SMOLMC_STEP_SWAP) {
                cand[0] = sbase + (int)__umulhi(o.w[0], nact);
                cand[1] = sbase + (int)__umulhi(o.w[1], nact);
                cand[2] = sbase + (int)__umulhi(o.w[2], nact);
                cand[3] = sbase + (int)__umulhi(o.w[3], nact);
#pragma unroll
                for (int j = 0; j < 4; ++j) canda[j] = lean_swz(cand[j], swa, swm, swb);
            }
        }
        const int l4 = (int)(step & 15ull) * 4;
        // prefetch the index row of the next step's site (depends only on random words)
        const int s1n = sbase + (int)__umulhi(rdlane(W1, l4), nact);
        uint16_t rown[ROW];
        {
            const uint16_t *p = idx_lane + (size_t)s1n * (64 * ROW);
#pragma unroll
            for (int q = 0; q < ROW; ++q) rown[q] = p[q];
        }
        const int a1 = lean_swz(s1, swa, swm, swb);
        const int o1 = uni((int)occ[a1]);
        int nfl, s2 = s1, a2 = a1, n1, n2 = 0, o2 = 0;
        if (STEP == SMOLMC_STEP_FLIP) {
            // Flip.propose_step (mcusher.py:154-170), default encoding 0..nc-1
            const uint32_t kk = __umulhi(rdlane(W0, l4 + 1), (uint32_t)(P.ncodes - 1));
            n1 = (int)kk + ((int)kk >= o1 ? 1 : 0);
            nfl = 1;
        } else {
            // Swap.propose_step (mcusher.py:176-200) by rejection over the candidate sequence
            int found = -1, fo = 0, fa = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (found < 0) {
                    const int v = (int)occ[canda[j]];
                    const unsigned long long m = __ballot(v != o1) & (0xEull << l4);
                    if (m) {
                        const int b = __ffsll((long long)m) - 1;
                        found = (int)rdlane((uint32_t)cand[j], b);
                        fa = (int)rdlane((uint32_t)canda[j], b);
                        fo = (int)rdlane((uint32_t)v, b);
                    }
                }
            }
            if (found < 0) {
                for (uint32_t q = 0;; ++q) {
                    const philox_out o = philox4x32_10((uint32_t)step, (uint32_t)(step >> 32),
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
                        found = (int)rdlane((uint32_t)selsite, b);
                        fa = lean_swz(found, swa, swm, swb);
                        fo = (int)rdlane((uint32_t)selv, b);
                        break;
                    }
                    if ((q & 63u) == 0) { // swap_options.size == 0 -> empty step
                        int any = 0;
                        for (uint32_t a = lane; a < nact; a += 64)
                            any |= ((int)occ[lean_swz(sbase + (int)a, swa, swm, swb)] != o1);
                        if (__ballot(any) == 0ull) break;
                    }
                }
            }
            if (found >= 0) { s2 = found; a2 = fa; o2 = fo; n1 = o2; n2 = o1; nfl = 2; }
            else { nfl = 0; n1 = o1; s2 = s1; a2 = a1; o2 = o1; n2 = o1; } // empty step: no-op 'flips'
        }

        // data-dependent row of site 2: issued before flip 1 is evaluated (s2 == s1 for the
        // rare empty step, the loaded row is then unused)
        uint16_t row2[ROW];
        if (STEP == SMOLMC_STEP_SWAP) {
            const uint16_t *p = idx_lane + (size_t)s2 * (64 * ROW);
#pragma unroll
            for (int q = 0; q < ROW; ++q) row2[q] = p[q];
        }

        // -------- enthalpy delta ---------------------------------------------------
        double e = 0.0, d1[NSLOT], d2[NSLOT];
        {
            const uint32_t pair1 = (uint32_t)o1 * snt8 + (uint32_t)n1 * nt8; // uniform
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) {
                uint32_t a = doff8[it];
#pragma unroll
                for (int m = 0; m < MM; ++m) a += __umul24(st8[it][m], (uint32_t)occ[row1[it * MM + m]]);
                d1[it] = *(const double *)((const unsigned char *)s_dt + (a + pair1));
                e = fma(wgt[it], d1[it], e);
            }
        }
        double ew_part = 0.0, ew_uni = 0.0; // lane-partial / uniform parts of the Ewald delta
        if (HAS_EW) {
            const int W = P.ew_W;
            const double dq = P.ew_qs[(size_t)s1 * W + n1] - P.ew_qs[(size_t)s1 * W + o1];
            ew_part = 2.0 * dq * lean_ewald_partial(P, occ, lane, s1, swa, swm, swb);
            ew_uni = 2.0 * dq * P.ew_frozen[s1] +
                     (P.ew_dg[(size_t)s1 * W + n1] - P.ew_dg[(size_t)s1 * W + o1]);
        }
        if (STEP == SMOLMC_STEP_SWAP) {
            // the second flip sees the first (expansion.py:217-229): apply it tentatively in
            // LDS (undone below on rejection) instead of patching every gathered value
            if (lane == 0) occ[a1] = (uint8_t)n1;
            const uint32_t pair2 = (uint32_t)o2 * snt8 + (uint32_t)n2 * nt8;
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) {
                uint32_t a = doff8[it];
#pragma unroll
                for (int m = 0; m < MM; ++m) a += __umul24(st8[it][m], (uint32_t)occ[row2[it * MM + m]]);
                d2[it] = *(const double *)((const unsigned char *)s_dt + (a + pair2));
                e = fma(wgt[it], d2[it], e);
            }
            if (HAS_EW) {
                const int W = P.ew_W;
                const double dq = P.ew_qs[(size_t)s2 * W + n2] - P.ew_qs[(size_t)s2 * W + o2];
                ew_part += 2.0 * dq * lean_ewald_partial(P, occ, lane, s2, swa, swm, swb);
                ew_uni += 2.0 * dq * P.ew_frozen[s2] +
                          (P.ew_dg[(size_t)s2 * W + n2] - P.ew_dg[(size_t)s2 * W + o2]);
            }
        }
        double dH = LEAN_WAVE_SUM(e);
        double dEw = 0.0;
        if (HAS_EW) {
            dEw = LEAN_WAVE_SUM(ew_part) + ew_uni;
            dH += P.ew_coef * dEw;
        }
        double dMu = 0.0;
        if (HAS_MU && nfl >= 1) {
            dMu = s_mu[n1] - s_mu[o1];
            if (nfl == 2) dMu += s_mu[n2] - s_mu[o2];
            dH -= dMu;
        }

        // -------- accept / update (metropolis.py:31-49, kernel/base.py:327-343) --------
        const double lu = __hiloint2double((int)rdlane((uint32_t)__double2hiint(logu), l4),
                                           (int)rdlane((uint32_t)__double2loint(logu), l4));
        bool accepted;
        if (!WL) {
            const double exponent = nbeta * dH + 0.0;
            accepted = (exponent >= 0.0) || (exponent > lu);
        } else {
            // WangLandau._accept_step (wanglandau.py:186-202)
            const double new_h = H + dH;
            if (new_h < P.wl.vmin || new_h >= P.wl.vmax) {
                accepted = false;
            } else {
                const int b = (int)floordiv_exact(H - P.wl.vmin, P.wl.bin);
                const int nb = (int)floordiv_exact(new_h - P.wl.vmin, P.wl.bin);
                const double exponent = wl_S[b] - wl_S[nb] + 0.0;
                accepted = (exponent >= 0.0) || (exponent > lu);
            }
        }
        if (accepted) {
            if (!WL) {
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) acc[it] += d1[it];
                if (STEP == SMOLMC_STEP_SWAP) {
#pragma unroll
                    for (int it = 0; it < NSLOT; ++it) acc[it] += d2[it];
                }
            } else {
                // _do_accept_step (wanglandau.py:204-220): current features += delta features
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) {
                    const double dd = STEP == SMOLMC_STEP_SWAP ? d1[it] + d2[it] : d1[it];
                    __hip_atomic_fetch_add(&s_feat[sfeat[it]], sfs[it] * dd, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            if (lane == 0) {
                if (STEP == SMOLMC_STEP_FLIP) occ[a1] = (uint8_t)n1;
                if (STEP == SMOLMC_STEP_SWAP) occ[a2] = (uint8_t)n2; // (n2 == o1 == occ[a1] when empty)
            }
            acc_mu += dMu;
            acc_ew += dEw;
            H += dH;
            nacc++;
        } else if (STEP == SMOLMC_STEP_SWAP) {
            if (lane == 0) occ[a1] = (uint8_t)o1; // undo the tentative first flip
        }
        last_acc = accepted ? 1 : 0;
        s1 = s1n;
#pragma unroll
        for (int q = 0; q < ROW; ++q) row1[q] = rown[q];

        if (WL) {
            // WangLandau._do_post_step (wanglandau.py:222-266)
            const double bq = floordiv_exact(H - P.wl.vmin, P.wl.bin);
            if (bq >= 0.0 && bq < (double)P.wl.L) {
                const int b = (int)bq;
                wl_counter++;
                const size_t cell = (size_t)r * P.wl.L + b;
                long long total = 0;
                if (lane == 0) total = P.wl.occur[cell];
                total = ((long long)(unsigned)uni((int)(total >> 32)) << 32) |
                        (unsigned)uni((int)(total & 0xffffffffll));
                if (lane < P.F) {
                    double *mf = P.wl.meanf + cell * P.F + lane;
                    const double inv = 1.0 / (double)(total + 1);
                    *mf = inv * (s_feat[lane] + (double)total * (*mf));
                }
                if (wl_counter % P.wl.update == 0 && lane == 0) {
                    wl_S[b] += wl_m;
                    wl_Hh[b] += 1;
                    P.wl.occur[cell] = total + 1;
                }
            }
            if (wl_counter % P.wl.check == 0) {
                long cnt = 0;
                double sum = 0;
                for (int i = lane; i < P.wl.L; i += 64)
                    if (wl_S[i] > 0) { cnt++; sum += (double)wl_Hh[i]; }
                const double tcnt = wave_sum_all((double)cnt), tsum = wave_sum_all(sum);
                if (tcnt >= 2.0) {
                    const double thr = P.wl.flat * (tsum / tcnt);
                    int bad = 0;
                    for (int i = lane; i < P.wl.L; i += 64)
                        if (wl_S[i] > 0 && !((double)wl_Hh[i] > thr)) bad = 1;
                    if (__ballot(bad) == 0ull) {
                        for (int i = lane; i < P.wl.L; i += 64) wl_Hh[i] = 0;
                        wl_m = wl_m / P.wl.div;
                    }
                }
            }
        }

        if (P.smp.every && --smp_countdown == 0) { // record one thinned sample of this walker
            smp_countdown = P.smp.every;
            const size_t row = (size_t)smp_index * P.R + r;
            smp_index++;
            if (WL) {
                if (lane < P.F) P.smp.feat[row * P.F + lane] = s_feat[lane];
            } else {
                s_feat[lane] = 0.0;
#pragma unroll
                for (int it = 0; it < NSLOT; ++it)
                    __hip_atomic_fetch_add(&s_feat[sfeat[it]], sfs[it] * acc[it], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (lane < P.Fce) P.smp.feat[row * P.F + lane] = base_feat + s_feat[lane];
            }
            if (!WL && HAS_EW && lane == P.Fce) P.smp.feat[row * P.F + lane] = base_feat + acc_ew;
            if (!WL && HAS_MU && lane == P.Fce + (HAS_EW ? 1 : 0))
                P.smp.feat[row * P.F + lane] = base_feat + acc_mu;
            if (lane == 0) {
                P.smp.H[row] = H;
                P.smp.acc[row] = (uint8_t)last_acc;
            }
            if (P.smp.occ) {
                uint32_t *dst = (uint32_t *)(P.smp.occ + row * P.Npad);
                for (int i = lane; i < P.Npad / 4; i += 64)
                    dst[i] = *(const uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb));
            }
        }
    }

    // ---- write back ---------------------------------------------------------------
    {
        uint32_t *dst = (uint32_t *)(P.occ + (size_t)r * P.Npad);
        for (int i = lane; i < P.Npad / 4; i += 64)
            dst[i] = *(const uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb));
    }
    if (WL) {
        if (lane < P.F) featp[lane] = s_feat[lane];
        for (int i = lane; i < P.wl.L; i += 64) {
            P.wl.entropy[(size_t)r * P.wl.L + i] = wl_S[i];
            P.wl.hist[(size_t)r * P.wl.L + i] = wl_Hh[i];
        }
        if (lane == 0) {
            P.wl.m[r] = wl_m;
            P.wl.counter[r] = wl_counter;
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
        if (!WL && HAS_EW) featp[P.Fce] += acc_ew;
        if (!WL && HAS_MU) featp[P.Fce + (HAS_EW ? 1 : 0)] += acc_mu;
        P.enthalpy[r] = H;
        P.nsteps[r] = step;
        P.nacc[r] = nacc;
        P.last_acc[r] = (uint8_t)last_acc;
    }
}


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
template <int NSLOT, int MM>
__global__ void __launch_bounds__(256) mc_table_kernel(const LeanParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int r = uni(blockIdx.x * nwaves + wave);
    double *s_dt = (double *)smem;
    double *s_mu = s_dt + P.dt_len; // 8 doubles
    const size_t per_wave = (size_t)P.Nlds + 64 * 8 + 64;
    unsigned char *wbase = (unsigned char *)(s_mu + 8) + (size_t)wave * per_wave;
    uint8_t *occ = wbase;
    double *s_feat = (double *)(wbase + P.Nlds);
    int *s_cnt = (int *)(s_feat + 64); // species counts of the walker [<= 8]
    const int swa = P.swz_a, swm = P.swz_m, swb = P.swz_b;
    const bool has_mu = P.mu_row != nullptr, has_ew = P.ew_G != nullptr;
    for (int i = threadIdx.x; i < P.dt_len; i += blockDim.x) s_dt[i] = P.dt[i];
    if (threadIdx.x < 8) s_mu[threadIdx.x] = (has_mu && threadIdx.x < P.ncodes) ? P.mu_row[threadIdx.x] : 0.0;
    const bool live = r < P.R;
    if (live) {
        const uint32_t *src = (const uint32_t *)(P.occ + (size_t)r * P.Npad);
        for (int i = lane; i < P.Npad / 4; i += 64)
            *(uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb)) = src[i];
        s_feat[lane] = 0.0;
        if (lane < 16) s_cnt[lane] = 0;
    }
    __syncthreads();
    if (!live) return;
    const int nc = P.ncodes, sbase = P.sbase;
    const uint32_t nact = (uint32_t)P.nact, nt8 = P.nt8, snt8 = P.snt8;
    for (int a = lane; a < (int)nact; a += 64)
        atomicAdd(&s_cnt[(int)occ[lean_swz(sbase + a, swa, swm, swb)]], 1);

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
    unsigned long long nacc = P.nacc[r];
    const uint32_t key0 = (uint32_t)P.seeds[r], key1 = (uint32_t)(P.seeds[r] >> 32);
    double acc_mu = 0.0, acc_ew = 0.0;
    int last_acc = 1;
    double *featp = P.features + (size_t)r * P.F;
    const double base_feat = lane < P.F ? featp[lane] : 0.0;
    long long smp_countdown = P.smp.every, smp_index = 0;
    uint32_t W0 = 0, W1 = 0, W2 = 0, W3 = 0;
    double logu = 0.0;
    unsigned long long batch_base = ~0ull;
    uint32_t w_site_carry = 0;
    constexpr int ROW = NSLOT * MM;
    const uint16_t *idx_lane = P.idx + (size_t)lane * ROW;
    const int nf2 = 2 * P.tf_n;

    for (long long it_step = 0; it_step < P.steps; ++it_step, ++step) {
        const unsigned long long base = step & ~15ull;
        if (base != batch_base) {
            if (batch_base == base - 16) {
                w_site_carry = rdlane(W1, 60);
            } else {
                const unsigned long long sp = base - 1ull;
                w_site_carry = (uint32_t)uni((int)philox4x32_10((uint32_t)sp, (uint32_t)(sp >> 32), 0u, 0u,
                                                                key0, key1).w[1]);
            }
            batch_base = base;
            const unsigned long long st = base + (unsigned)(lane >> 2);
            const philox_out o = philox4x32_10((uint32_t)st, (uint32_t)(st >> 32), (uint32_t)(lane & 3),
                                               0u, key0, key1);
            W0 = o.w[0]; W1 = o.w[1]; W2 = o.w[2]; W3 = o.w[3];
            logu = log(philox_u53(o.w[2], o.w[3]));
        }
        const int l4 = (int)(step & 15ull) * 4;
        const uint32_t w_site = l4 == 0 ? w_site_carry : rdlane(W1, l4 - 4);

        // flips of this step live lane-indexed: lane f holds flip f
        int vsite = 0, vnew = 0, vold = 0;
        int nfl = 0, dir = -1;
        double log_priori = 0.0;
        bool do_swap = (double)rdlane(W0, l4) * (1.0 / 4294967296.0) < P.tf_sw;
        double sumw = 0.0;
        if (!do_swap) { // flip_weights_mask (math.py:832-867) at the current counts
            for (int idx = 0; idx < nf2; ++idx) {
                const int *row = P.tf_table + (idx >> 1) * nc;
                const int sg = (idx & 1) ? -1 : 1;
                bool ok = true;
                for (int c = 0; c < nc; ++c) {
                    const int v = s_cnt[c] + sg * row[c];
                    ok = ok && v >= 0 && v <= (int)nact;
                }
                sumw += ok ? P.tf_w[idx] : 0.0;
            }
            sumw = uni_d(sumw);
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
                    const philox_out o = philox4x32_10((uint32_t)step, (uint32_t)(step >> 32),
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
        } else {
            // choose_section_from_partition (math.py:870-893) with W(step, 1, 0)
            const double target = (double)rdlane(W0, l4 + 1) * (1.0 / 4294967296.0) * sumw;
            double cum = 0.0;
            int last = -1;
            for (int idx = 0; idx < nf2 && dir < 0; ++idx) {
                const int *row = P.tf_table + (idx >> 1) * nc;
                const int sg = (idx & 1) ? -1 : 1;
                bool ok = true;
                for (int c = 0; c < nc; ++c) {
                    const int v = s_cnt[c] + sg * row[c];
                    ok = ok && v >= 0 && v <= (int)nact;
                }
                if (!ok) continue;
                last = idx;
                cum += P.tf_w[idx];
                if (target < cum) dir = idx;
            }
            if (dir < 0) dir = last;
            dir = uni(dir);
            const int *urow = P.tf_table + (dir >> 1) * nc;
            const int usg = (dir & 1) ? -1 : 1;
            // compute_log_priori_factor (mcusher.py:656-711)
            {
                double sum_next = 0.0;
                for (int idx = 0; idx < nf2; ++idx) {
                    const int *row = P.tf_table + (idx >> 1) * nc;
                    const int sg = (idx & 1) ? -1 : 1;
                    bool ok = true;
                    for (int c = 0; c < nc; ++c) {
                        const int v = s_cnt[c] + usg * urow[c] + sg * row[c];
                        ok = ok && v >= 0 && v <= (int)nact;
                    }
                    sum_next += ok ? P.tf_w[idx] : 0.0;
                }
                const double p_now = (1.0 - P.tf_sw) * P.tf_w[dir] / sumw;
                const double p_next = (1.0 - P.tf_sw) * P.tf_w[dir ^ 1] / sum_next;
                double lf = log(p_next / p_now);
                for (int c = 0; c < nc; ++c) {
                    const int u = usg * urow[c], n0 = s_cnt[c];
                    for (int k = 1; k <= u; ++k) lf -= log((double)(n0 + k));
                    for (int k = 0; k < -u; ++k) lf += log((double)(n0 - k));
                }
                log_priori = uni_d(lf);
            }
            // pick the sites of the depleted species from the candidate stream
            // c_t = W(step, 4 + t / 4, t % 4): 256 candidates per wave round
            int vcol = 0, ncol = 0; // collected sites, lane-indexed
            long long tlast = -1;
            uint32_t round = 0;
            int cs[4] = {0, 0, 0, 0}, cv[4] = {0, 0, 0, 0};
            bool have_round = false;
            for (int c = 0; c < nc; ++c) {
                int need = -(usg * urow[c]);
                while (need > 0) {
                    if (!have_round) {
                        const philox_out o = philox4x32_10((uint32_t)step, (uint32_t)(step >> 32),
                                                           4u + 64u * round + (uint32_t)lane, 0u, key0, key1);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            cs[j] = sbase + (int)__umulhi(o.w[j], nact);
                            cv[j] = (int)occ[lean_swz(cs[j], swa, swm, swb)];
                        }
                        have_round = true;
                    }
                    // smallest stream position t > tlast in this lane that holds species c and
                    // was not collected yet
                    long long mint = -1;
#pragma unroll
                    for (int j = 3; j >= 0; --j) {
                        const long long t = (long long)round * 256 + 4 * lane + j;
                        bool ok = cv[j] == c && t > tlast;
                        for (int z = 0; z < ncol; ++z) ok = ok && cs[j] != (int)rdlane((uint32_t)vcol, z);
                        if (ok) mint = t;
                    }
                    const unsigned long long m = __ballot(mint >= 0);
                    if (!m) { round++; have_round = false; continue; }
                    const int b = __ffsll((long long)m) - 1;
                    const int tj = (int)(((unsigned)rdlane((uint32_t)(int)(mint & 0xffffffffll), b)) & 3u);
                    tlast = (long long)round * 256 + 4 * b + tj;
                    const int picked = (int)rdlane((uint32_t)(tj == 0 ? cs[0] : tj == 1 ? cs[1] : tj == 2 ? cs[2] : cs[3]), b);
                    if (lane == ncol) vcol = picked;
                    ncol++;
                    need--;
                }
            }
            // random assignment of the collected sites to the enriched species (:627-631)
            int qdraw = 0;
            for (int c = 0; c < nc; ++c) {
                const int u = usg * urow[c];
                for (int k = 0; k < u; ++k) {
                    const int wl = l4 + 2 + (qdraw >> 2);
                    const int wj = qdraw & 3;
                    const uint32_t word = rdlane(wj == 0 ? W0 : wj == 1 ? W1 : wj == 2 ? W2 : W3, wl);
                    qdraw++;
                    const int rr = (int)__umulhi(word, (uint32_t)ncol);
                    const int site = (int)rdlane((uint32_t)vcol, rr);
                    const int od = uni((int)occ[lean_swz(site, swa, swm, swb)]);
                    if (lane == nfl) { vsite = site; vnew = c; vold = od; }
                    nfl++;
                    const int nxt = __shfl_down(vcol, 1);
                    if (lane >= rr) vcol = nxt; // list.remove keeps the order of the rest
                    ncol--;
                }
            }
        }

        // -------- sequential evaluation of the flips of this step -----------------------
        double e = 0.0, pend[NSLOT], ew_part = 0.0, ew_uni = 0.0, dMu = 0.0;
#pragma unroll
        for (int it = 0; it < NSLOT; ++it) pend[it] = 0.0;
        for (int f = 0; f < nfl; ++f) {
            const int s = (int)rdlane((uint32_t)vsite, f), nw = (int)rdlane((uint32_t)vnew, f);
            const int od = (int)rdlane((uint32_t)vold, f);
            const uint16_t *p = idx_lane + (size_t)s * (64 * ROW);
            uint16_t row[ROW];
#pragma unroll
            for (int q = 0; q < ROW; ++q) row[q] = p[q];
            const uint32_t pair = (uint32_t)od * snt8 + (uint32_t)nw * nt8;
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) {
                uint32_t a = doff8[it];
#pragma unroll
                for (int m = 0; m < MM; ++m) a += __umul24(st8[it][m], (uint32_t)occ[row[it * MM + m]]);
                const double d = *(const double *)((const unsigned char *)s_dt + (a + pair));
                e = fma(wgt[it], d, e);
                pend[it] += d;
            }
            if (has_ew) {
                const int W = P.ew_W;
                const double dq = P.ew_qs[(size_t)s * W + nw] - P.ew_qs[(size_t)s * W + od];
                ew_part += 2.0 * dq * lean_ewald_partial(P, occ, lane, s, swa, swm, swb);
                ew_uni += 2.0 * dq * P.ew_frozen[s] + (P.ew_dg[(size_t)s * W + nw] - P.ew_dg[(size_t)s * W + od]);
            }
            if (has_mu) dMu += s_mu[nw] - s_mu[od];
            if (lane == 0) occ[lean_swz(s, swa, swm, swb)] = (uint8_t)nw; // tentative
        }
        double dH = wave_sum_all(e);
        double dEw = 0.0;
        if (has_ew) {
            dEw = wave_sum_all(ew_part) + ew_uni;
            dH += P.ew_coef * dEw;
        }
        if (has_mu) dH -= dMu;
        const double exponent = nbeta * dH + log_priori; // metropolis.py:41-42
        const double lu = __hiloint2double((int)rdlane((uint32_t)__double2hiint(logu), l4),
                                           (int)rdlane((uint32_t)__double2loint(logu), l4));
        const bool accepted = (exponent >= 0.0) || (exponent > lu);
        if (accepted) {
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) acc[it] += pend[it];
            if (dir >= 0 && lane < nc) {
                const int *urow = P.tf_table + (dir >> 1) * nc;
                s_cnt[lane] += ((dir & 1) ? -1 : 1) * urow[lane];
            }
            acc_mu += dMu;
            acc_ew += dEw;
            H += dH;
            nacc++;
        } else {
            for (int f = nfl - 1; f >= 0; --f) { // undo the tentative flips
                const int s = (int)rdlane((uint32_t)vsite, f), od = (int)rdlane((uint32_t)vold, f);
                if (lane == 0) occ[lean_swz(s, swa, swm, swb)] = (uint8_t)od;
            }
        }
        last_acc = accepted ? 1 : 0;

        if (P.smp.every && --smp_countdown == 0) {
            smp_countdown = P.smp.every;
            const size_t rowi = (size_t)smp_index * P.R + r;
            smp_index++;
            s_feat[lane] = 0.0;
#pragma unroll
            for (int it = 0; it < NSLOT; ++it)
                __hip_atomic_fetch_add(&s_feat[sfeat[it]], sfs[it] * acc[it], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (lane < P.Fce) P.smp.feat[rowi * P.F + lane] = base_feat + s_feat[lane];
            if (has_ew && lane == P.Fce) P.smp.feat[rowi * P.F + lane] = base_feat + acc_ew;
            if (has_mu && lane == P.Fce + (has_ew ? 1 : 0)) P.smp.feat[rowi * P.F + lane] = base_feat + acc_mu;
            if (lane == 0) {
                P.smp.H[rowi] = H;
                P.smp.acc[rowi] = (uint8_t)last_acc;
            }
            if (P.smp.occ) {
                uint32_t *dst = (uint32_t *)(P.smp.occ + rowi * P.Npad);
                for (int i = lane; i < P.Npad / 4; i += 64)
                    dst[i] = *(const uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb));
            }
        }
    }

    {
        uint32_t *dst = (uint32_t *)(P.occ + (size_t)r * P.Npad);
        for (int i = lane; i < P.Npad / 4; i += 64)
            dst[i] = *(const uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb));
    }
    s_feat[lane] = 0.0;
#pragma unroll
    for (int it = 0; it < NSLOT; ++it)
        __hip_atomic_fetch_add(&s_feat[sfeat[it]], sfs[it] * acc[it], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (lane < P.Fce) featp[lane] = base_feat + s_feat[lane];
    if (lane == 0) {
        if (has_ew) featp[P.Fce] += acc_ew;
        if (has_mu) featp[P.Fce + (has_ew ? 1 : 0)] += acc_mu;
        P.enthalpy[r] = H;
        P.nsteps[r] = step;
        P.nacc[r] = nacc;
        P.last_acc[r] = (uint8_t)last_acc;
    }
}

// ----------------------------------------------------------------------------
// reference-layout evaluation kernels (parity API + initial trace)
// ----------------------------------------------------------------------------
struct RefTables { // device copies of the smolmc_tables arrays
    int N, Npad, P, num_orbits, num_corr, n_orb, Fce, F, corr_mode;
    const int *orb_id, *orb_bit_id, *orb_nsites, *orb_nfunc, *orb_tensor_len, *orb_stride_off,
        *tensor_indices;
    const long long *orb_ctensor_off, *orb_itensor_off, *full_off, *site_ptr, *loc_off;
    const double *corr_tensors, *interaction_tensors, *loc_ratio;
    const int *full_idx, *loc_orbit, *loc_nrows, *loc_idx;
    double offset;
    int has_ewald, ew_W, ew_M, has_mu, mu_W;
    const int *ew_inds;
    const double *ew_M_rowmajor; // original matrix (for the full feature)
    const double *ew_Mt;
    const double *mu;
};

__device__ __forceinline__ double block_sum(double v, double *sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double s = 0;
    for (int i = 0; i < nw; ++i) s += sh[i];
    return s;
}

// Ensemble.compute_feature_vector for one occupancy per block
// (evaluator.pyx:121-209 x size; processor/ewald.py:128-145; ensemble.py:343-349)
__global__ void __launch_bounds__(256) eval_full_kernel(const RefTables T, const uint8_t *occ_all,
                                                        double *out_all) {
    __shared__ double sh[8];
    const uint8_t *occ = occ_all + (size_t)blockIdx.x * T.Npad;
    double *out = out_all + (size_t)blockIdx.x * T.F;
    if (threadIdx.x == 0) out[0] = (T.corr_mode ? 1.0 : T.offset) * (double)T.P;
    for (int n = 0; n < T.n_orb; ++n) {
        const int I = T.orb_nsites[n], K = T.corr_mode ? T.orb_nfunc[n] : 1;
        const int Nt = T.orb_tensor_len[n];
        const int *st = T.tensor_indices + T.orb_stride_off[n];
        const int *ind = T.full_idx + T.full_off[n];
        const long long J = (T.full_off[n + 1] - T.full_off[n]) / I;
        for (int k = 0; k < K; ++k) {
            const double *t = T.corr_mode ? T.corr_tensors + T.orb_ctensor_off[n] + (size_t)k * Nt
                                          : T.interaction_tensors + T.orb_itensor_off[n];
            double p = 0;
            for (long long j = threadIdx.x; j < J; j += blockDim.x) {
                int index = 0;
                for (int i = 0; i < I; ++i) index += st[i] * (int)occ[ind[j * I + i]];
                p += t[index];
            }
            p = block_sum(p, sh);
            if (threadIdx.x == 0) {
                const int o = T.corr_mode ? T.orb_bit_id[n] + k : T.orb_id[n];
                out[o] = p / (double)J * (double)T.P;
            }
        }
    }
    int f = T.Fce;
    if (T.has_ewald) {
        double s = 0;
        for (int a = 0; a < T.N; ++a) {
            const int ia = T.ew_inds[(size_t)a * T.ew_W + occ[a]];
            if (ia == -1) continue;
            const double *row = T.ew_M_rowmajor + (size_t)ia * T.ew_M;
            for (int b = threadIdx.x; b < T.N; b += blockDim.x) {
                const int ib = T.ew_inds[(size_t)b * T.ew_W + occ[b]];
                if (ib != -1) s += row[ib];
            }
        }
        s = block_sum(s, sh);
        if (threadIdx.x == 0) out[f] = s;
        f++;
    }
    if (T.has_mu) {
        double s = 0;
        for (int a = threadIdx.x; a < T.N; a += blockDim.x) s += T.mu[(size_t)a * T.mu_W + occ[a]];
        s = block_sum(s, sh);
        if (threadIdx.x == 0) out[f] = s;
    }
}

// Ensemble.compute_feature_vector_change for one step (<= 2 sequential flips) per
// wave, reference table layout and arithmetic chain p / ratio / J, x size
// (evaluator.pyx:244-262, :302-315; expansion.py:217-231).
__global__ void __launch_bounds__(64) eval_delta_kernel(const RefTables T, const uint8_t *occ,
                                                        const int *flips, double *out_all) {
    const int lane = threadIdx.x;
    const int *fl = flips + (size_t)blockIdx.x * 4;
    double *out = out_all + (size_t)blockIdx.x * T.F;
    for (int i = lane; i < T.F; i += 64) out[i] = 0.0;
    __syncthreads();
    const int nfl = fl[0] < 0 ? 0 : (fl[2] < 0 ? 1 : 2);
    double dew = 0, dmu = 0;
    for (int f = 0; f < nfl; ++f) {
        const int s = fl[2 * f], newc = fl[2 * f + 1];
        const int ps = f == 1 ? fl[0] : -1, pc = f == 1 ? fl[1] : 0;
        for (long long rr = T.site_ptr[s]; rr < T.site_ptr[s + 1]; ++rr) {
            const int n = T.loc_orbit[rr];
            const int I = T.orb_nsites[n], K = T.corr_mode ? T.orb_nfunc[n] : 1, Nt = T.orb_tensor_len[n];
            const int *st = T.tensor_indices + T.orb_stride_off[n];
            const int *ind = T.loc_idx + T.loc_off[rr];
            const int J = T.loc_nrows[rr];
            for (int k = 0; k < K; ++k) {
                const double *t = T.corr_mode ? T.corr_tensors + T.orb_ctensor_off[n] + (size_t)k * Nt
                                              : T.interaction_tensors + T.orb_itensor_off[n];
                double p = 0;
                for (int j = lane; j < J; j += 64) {
                    int ind_i = 0, ind_f = 0;
                    for (int i = 0; i < I; ++i) {
                        const int x = ind[j * I + i];
                        int v = occ[x];
                        if (x == ps) v = pc;
                        const int vf = (x == s) ? newc : v;
                        ind_i += st[i] * v;
                        ind_f += st[i] * vf;
                    }
                    p += t[ind_f] - t[ind_i];
                }
                p = wave_sum(p);
                if (lane == 0) {
                    const int o = T.corr_mode ? T.orb_bit_id[n] + k : T.orb_id[n];
                    out[o] += p / T.loc_ratio[rr] / (double)J;
                }
            }
        }
        if (T.has_ewald) {
            // ewald.pyx:38-58
            int oldc = occ[s];
            if (s == ps) oldc = pc;
            const int W = T.ew_W;
            const int add = T.ew_inds[(size_t)s * W + newc], sub = T.ew_inds[(size_t)s * W + oldc];
            double o = 0;
            for (int k = lane; k < T.N; k += 64) {
                int v = occ[k];
                if (k == ps) v = pc;
                const int vf = (k == s) ? newc : v;
                const int i = T.ew_inds[(size_t)k * W + vf], j = T.ew_inds[(size_t)k * W + v];
                if (i != -1 && add != -1)
                    o += (i != add ? 2.0 : 1.0) * T.ew_Mt[(size_t)add * T.ew_M + i];
                if (j != -1 && sub != -1)
                    o -= (j != sub ? 2.0 : 1.0) * T.ew_Mt[(size_t)sub * T.ew_M + j];
            }
            dew += wave_sum(o);
        }
        if (T.has_mu)
            dmu += T.mu[(size_t)s * T.mu_W + newc] - T.mu[(size_t)s * T.mu_W + occ[s]];
    }
    __syncthreads();
    for (int i = lane; i < T.Fce; i += 64) out[i] *= (double)T.P;
    if (lane == 0) {
        int f = T.Fce;
        if (T.has_ewald) out[f++] = dew;
        if (T.has_mu) out[f++] = dmu;
    }
}

__global__ void pack_occ_kernel(const int *occ32, uint8_t *occ8, int N, int Npad, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t rr = i / Npad;
    const int s = (int)(i % Npad);
    occ8[i] = s < N ? (uint8_t)occ32[rr * N + s] : 0;
}
__global__ void unpack_occ_kernel(const uint8_t *occ8, int *occ32, int N, int Npad, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t rr = i / N;
    const int s = (int)(i % N);
    occ32[i] = (int)occ8[rr * Npad + s];
}
__global__ void dot_features_kernel(const double *features, const double *natural, double *enthalpy,
                                    int R, int F) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    double s = 0;
    for (int i = 0; i < F; ++i) s += natural[i] * features[(size_t)r * F + i];
    enthalpy[r] = s;
}

// ----------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct smolmc_handle {
    smolmc_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    std::vector<void *> allocs;
    KParams kp;
    RefTables rt;
    int R = 0, N = 0, Npad = 0, F = 0, Fce = 0, L = 0;
    std::vector<double> natural;
    // dispatch
    int nslot = 0, mm = 0;
    bool generic = false, idx16 = false;
    size_t lds_bytes = 0;
    int waves_per_block = 4;
    // lean kernel (single class / single contiguous sublattice / interactions / no ewald)
    bool lean_tables = false, lean = false;
    int lean_nslot = 0, lean_mm = 0;
    size_t lean_lds = 0;
    LeanParams lp;
    // device-side samples of the last smolmc_run_sampled
    SampleBufs smp;
    long long smp_n = 0;
    bool smp_has_occ = false;
    // scratch
    uint8_t *d_eval_occ = nullptr;
    size_t eval_occ_cap = 0;
    double *d_natural = nullptr;
    double *d_beta = nullptr;
};

static void free_samples(smolmc_handle *h);

template <typename T>
static int dev_upload(smolmc_handle *h, const T *src, size_t n, const T **dst) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(n * sizeof(T), 16);
    HIPCHK(hipMalloc(&p, bytes));
    h->allocs.push_back(p);
    if (n) HIPCHK(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = (const T *)p;
    return 0;
}
template <typename T> static int dev_alloc(smolmc_handle *h, size_t n, T **dst, bool zero = true) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(n * sizeof(T), 16);
    HIPCHK(hipMalloc(&p, bytes));
    h->allocs.push_back(p);
    if (zero) HIPCHK(hipMemset(p, 0, bytes));
    *dst = (T *)p;
    return 0;
}
#define TRY(x)                                                                            \
    do {                                                                                  \
        if (int rc_ = (x)) return rc_;                                                    \
    } while (0)

static int num_ce_features(const smolmc_tables *t) {
    return t->feature_mode == SMOLMC_FEATURES_CORRELATIONS ? t->num_corr : t->num_orbits;
}

// Build the MC-optimised tables (classes, slot descriptors, member index rows).
static int build_mc_tables(smolmc_handle *h, const smolmc_tables *t) {
    const int N = t->num_sites;
    const bool corr = t->feature_mode == SMOLMC_FEATURES_CORRELATIONS;
    // does any local row contain a repeated site (aliased tiny supercells)?
    bool aliased = false;
    int maxI = 1;
    for (int s = 0; s < N && !aliased; ++s)
        for (int64_t r = t->site_ptr[s]; r < t->site_ptr[s + 1] && !aliased; ++r) {
            const int o = t->loc_orbit[r], I = t->orb_nsites[o];
            const int32_t *rows = t->loc_idx + t->loc_off[r];
            for (int j = 0; j < t->loc_nrows[r] && !aliased; ++j)
                for (int a = 0; a < I && !aliased; ++a)
                    for (int b = a + 1; b < I; ++b)
                        if (rows[j * I + a] == rows[j * I + b]) aliased = true;
        }
    for (int o = 0; o < t->n_orb; ++o) maxI = std::max(maxI, (int)t->orb_nsites[o]);
    if (maxI > SMOLMC_MAX_CLUSTER_SITES) return fail("cluster larger than SMOLMC_MAX_CLUSTER_SITES");
    for (int o = 0; o < t->n_orb; ++o)
        for (int i = 0; i < t->orb_nsites[o]; ++i)
            if (t->tensor_indices[t->orb_stride_off[o] + i] > 65535)
                return fail("tensor stride exceeds 16 bits");
    h->generic = aliased;
    const int need_mm = aliased ? maxI : std::max(1, maxI - 1);

    // per-site slot lists: (orbit, self position p, row pointer, record r)
    struct Slot {
        int orbit, p, nmem;
        int64_t rec;
        const int32_t *row;
    };
    std::vector<std::vector<Slot>> slots(N);
    std::vector<int> site_class(N, 255);
    std::map<std::vector<long long>, int> class_of;
    std::vector<int> class_rep; // representative site per class
    for (int s = 0; s < N; ++s) {
        if (t->site_ptr[s] == t->site_ptr[s + 1]) continue;
        std::vector<Slot> &sl = slots[s];
        std::vector<long long> sig;
        for (int64_t r = t->site_ptr[s]; r < t->site_ptr[s + 1]; ++r) {
            const int o = t->loc_orbit[r], I = t->orb_nsites[o], J = t->loc_nrows[r];
            const int32_t *rows = t->loc_idx + t->loc_off[r];
            std::vector<Slot> rec;
            for (int j = 0; j < J; ++j) {
                int p = 0;
                if (!aliased)
                    for (int a = 0; a < I; ++a)
                        if (rows[j * I + a] == s) p = a;
                rec.push_back(Slot{o, p, aliased ? I : I - 1, r, rows + (size_t)j * I});
            }
            std::stable_sort(rec.begin(), rec.end(), [](const Slot &a, const Slot &b) { return a.p < b.p; });
            sig.push_back(o);
            sig.push_back(J);
            long long rb;
            memcpy(&rb, &t->loc_ratio[r], 8);
            sig.push_back(rb);
            for (int a = 0; a < I; ++a) {
                long long cnt = 0;
                for (auto &q : rec) cnt += q.p == a;
                sig.push_back(cnt);
            }
            sl.insert(sl.end(), rec.begin(), rec.end());
        }
        // most members first so that iterations are homogeneous
        std::stable_sort(sl.begin(), sl.end(), [](const Slot &a, const Slot &b) { return a.nmem > b.nmem; });
        auto itc = class_of.find(sig);
        if (itc == class_of.end()) {
            if (class_rep.size() >= 255) return fail("more than 255 site classes");
            itc = class_of.emplace(sig, (int)class_rep.size()).first;
            class_rep.push_back(s);
        }
        site_class[s] = itc->second;
    }
    const int nclasses = std::max<int>(1, (int)class_rep.size());
    size_t Cmax = 1;
    for (int s : class_rep) Cmax = std::max(Cmax, slots[s].size());
    const int Cpad = (int)((Cmax + 63) / 64 * 64);
    const int niter_max = Cpad / 64;
    h->nslot = niter_max <= 2 ? 2 : (niter_max <= 4 ? 4 : 8);
    if (niter_max > 8) return fail("more than 512 clusters per site are not supported yet");
    if (aliased)
        h->mm = need_mm <= 3 ? 3 : 6;
    else
        h->mm = need_mm <= 2 ? 2 : (need_mm <= 3 ? 3 : 5);
    const int MM = h->mm;
    h->idx16 = (!aliased) && N <= 65535;

    // decision tensors per class, feature tensors shared
    std::vector<double> ft;
    std::vector<int> foff(t->n_orb);
    for (int o = 0; o < t->n_orb; ++o) {
        foff[o] = (int)ft.size();
        const int Nt = t->orb_tensor_len[o];
        if (corr) {
            const double *ct = t->corr_tensors + t->orb_ctensor_off[o];
            ft.insert(ft.end(), ct, ct + (size_t)t->orb_nfunc[o] * Nt);
        } else {
            const double *it = t->interaction_tensors + t->orb_itensor_off[o];
            ft.insert(ft.end(), it, it + Nt);
        }
    }
    std::vector<double> xt;
    std::vector<uint4> descA((size_t)nclasses * Cpad, make_uint4(0, 0, 0, 0));
    std::vector<uint4> descB((size_t)nclasses * Cpad, make_uint4(0, 0, 0, 0));
    std::vector<double> slot_fs((size_t)nclasses * Cpad, 0.0);
    std::vector<int> cls_niter(nclasses, 0);
    xt.push_back(0.0); // padded slots read xt[0] - xt[0]
    for (int c = 0; c < (int)class_rep.size(); ++c) {
        const int s = class_rep[c];
        const std::vector<Slot> &sl = slots[s];
        cls_niter[c] = (int)((sl.size() + 63) / 64);
        std::map<int64_t, int> xoff_of_rec;
        for (size_t q = 0; q < sl.size(); ++q) {
            const Slot &k = sl[q];
            const int o = k.orbit, I = t->orb_nsites[o], Nt = t->orb_tensor_len[o];
            const int32_t *st = t->tensor_indices + t->orb_stride_off[o];
            const double scale = (double)t->size / t->loc_ratio[k.rec] / (double)t->loc_nrows[k.rec];
            if (!xoff_of_rec.count(k.rec)) {
                xoff_of_rec[k.rec] = (int)xt.size();
                for (int i = 0; i < Nt; ++i) {
                    double v;
                    if (corr) {
                        // energy tensor: sum_k coef[bit_id+k] * ct[k][i], times scale
                        v = 0;
                        const double *ct = t->corr_tensors + t->orb_ctensor_off[o];
                        for (int kk = 0; kk < t->orb_nfunc[o]; ++kk)
                            v += t->ce_coefs[t->orb_bit_id[o] + kk] * ct[(size_t)kk * Nt + i];
                        v *= scale;
                    } else {
                        v = t->ce_coefs[t->orb_id[o]] * scale *
                            t->interaction_tensors[t->orb_itensor_off[o] + i];
                    }
                    xt.push_back(v);
                }
            }
            uint16_t sv[6] = {0, 0, 0, 0, 0, 0};
            if (aliased) {
                for (int a = 0; a < I; ++a) sv[a] = (uint16_t)st[a];
            } else {
                sv[0] = (uint16_t)st[k.p];
                int m = 1;
                for (int a = 0; a < I; ++a)
                    if (a != k.p) sv[m++] = (uint16_t)st[a];
            }
            uint4 A;
            A.x = (uint32_t)xoff_of_rec[k.rec];
            A.y = sv[0] | ((uint32_t)sv[1] << 16);
            A.z = sv[2] | ((uint32_t)sv[3] << 16);
            A.w = sv[4] | ((uint32_t)sv[5] << 16);
            descA[(size_t)c * Cpad + q] = A;
            uint4 B;
            B.x = (uint32_t)foff[o];
            B.y = (uint32_t)Nt;
            const uint32_t feat = corr ? (uint32_t)t->orb_bit_id[o] : (uint32_t)t->orb_id[o];
            const uint32_t K = corr ? (uint32_t)t->orb_nfunc[o] : 1u;
            B.z = feat | (K << 16);
            B.w = 0;
            descB[(size_t)c * Cpad + q] = B;
            slot_fs[(size_t)c * Cpad + q] = scale;
        }
    }
    if (xt.size() > 0xffffffffull) return fail("decision tensors too large");

    // member index rows [site][m][Cpad]; padded entries point at the site itself
    const size_t idx_n = (size_t)N * MM * Cpad;
    std::vector<int32_t> idx32(idx_n);
    for (int s = 0; s < N; ++s) {
        for (int m = 0; m < MM; ++m)
            for (int c = 0; c < Cpad; ++c) idx32[((size_t)s * MM + m) * Cpad + c] = s;
        const std::vector<Slot> &sl = slots[s];
        for (size_t q = 0; q < sl.size(); ++q) {
            const Slot &k = sl[q];
            const int I = t->orb_nsites[k.orbit];
            int m = 0;
            for (int a = 0; a < I; ++a) {
                if (!aliased && a == k.p) continue;
                idx32[((size_t)s * MM + m) * Cpad + q] = k.row[a];
                m++;
            }
        }
    }
    KParams &kp = h->kp;
    if (h->idx16) {
        std::vector<uint16_t> idx16(idx_n);
        for (size_t i = 0; i < idx_n; ++i) idx16[i] = (uint16_t)idx32[i];
        const uint16_t *d;
        TRY(dev_upload(h, idx16.data(), idx_n, &d));
        kp.idx = d;
    } else {
        const int32_t *d;
        TRY(dev_upload(h, idx32.data(), idx_n, &d));
        kp.idx = d;
    }
    std::vector<uint8_t> sc8(N);
    for (int s = 0; s < N; ++s) sc8[s] = (uint8_t)site_class[s];
    TRY(dev_upload(h, sc8.data(), (size_t)N, &kp.site_class));
    TRY(dev_upload(h, descA.data(), descA.size(), &kp.descA));
    TRY(dev_upload(h, descB.data(), descB.size(), &kp.descB));
    TRY(dev_upload(h, slot_fs.data(), slot_fs.size(), &kp.slot_fs));
    TRY(dev_upload(h, cls_niter.data(), cls_niter.size(), &kp.cls_niter));
    TRY(dev_upload(h, xt.data(), xt.size(), &kp.xt));
    TRY(dev_upload(h, ft.data(), ft.size(), &kp.ft));
    kp.xt_len = (int)xt.size();
    kp.ft_len = (int)ft.size();
    kp.nclasses = nclasses;
    kp.Cpad = Cpad;
    kp.Mmax = MM;

    // ---- lean tables (see mc_lean_kernel) ------------------------------------------
    memset(&h->lp, 0, sizeof(LeanParams));
    if (class_rep.size() == 1 && !aliased && !corr && N <= 65535 && niter_max <= 4 && need_mm <= 3 &&
        num_ce_features(t) <= 64) {
        const int NSL = niter_max <= 2 ? 2 : 4;
        const int MML = need_mm <= 2 ? 2 : 3;
        const int ROW = NSL * MML;
        std::vector<uint16_t> lidx((size_t)N * 64 * ROW);
        for (int s = 0; s < N; ++s) {
            for (int q = 0; q < 64 * ROW; ++q) lidx[(size_t)s * 64 * ROW + q] = (uint16_t)s;
            const std::vector<Slot> &sl = slots[s];
            for (size_t q = 0; q < sl.size(); ++q) {
                const Slot &k = sl[q];
                const int I = t->orb_nsites[k.orbit];
                const int it = (int)(q / 64), ln = (int)(q % 64);
                int m = 0;
                for (int a = 0; a < I; ++a) {
                    if (a == k.p) continue;
                    lidx[(((size_t)s * 64 + ln) * NSL + it) * MML + m] = (uint16_t)k.row[a];
                    m++;
                }
            }
        }
        // ---- LDS bank swizzle: pick the address permutation s ^ (((s >> a) & m) << b) that
        // minimises the modelled bank-conflict cycles of the occupancy gathers (each
        // ds_read_u8 is served in two 32-lane groups; a group costs the max number of
        // distinct dwords on one of the 32 banks).  See tools/lds_conflict_model.py.
        int Nlds = 16;
        while (Nlds < h->Npad) Nlds <<= 1;
        int best_a = 0, best_m = 0, best_b = 0;
        {
            auto model_cost = [&](int a, int m, int b) {
                double tot = 0;
                const int nsamp = std::min(N, 48);
                for (int k = 0; k < nsamp; ++k) {
                    const int s = (int)(((long long)k * 2654435761ll) % N);
                    if (slots[s].empty()) continue;
                    for (int q = 0; q < ROW; ++q)
                        for (int g = 0; g < 2; ++g) {
                            int cnt[32] = {0};
                            int seen[32];
                            int nseen = 0;
                            for (int ln = 32 * g; ln < 32 * g + 32; ++ln) {
                                const int x = lidx[((size_t)s * 64 + ln) * ROW + q];
                                const int dw = (x ^ (((x >> a) & m) << b)) >> 2;
                                bool dup = false;
                                for (int z = 0; z < nseen; ++z) dup |= seen[z] == dw;
                                if (!dup) { seen[nseen++] = dw; cnt[dw & 31]++; }
                            }
                            int mx = 0;
                            for (int z = 0; z < 32; ++z) mx = std::max(mx, cnt[z]);
                            tot += mx;
                        }
                }
                return tot;
            };
            double best = model_cost(0, 0, 0);
            const int amax = getenv("SMOLMC_NO_SWIZZLE") ? 0 : 12; // A/B switch for profiling
            for (int a = 3; a <= amax; ++a)
                for (int b = 2; b <= 5; ++b)
                    for (int m : {3, 7, 15, 31}) {
                        // bijection on [0, Nlds): source bits [a, a+k) above the destination
                        // bits [b, b+k) and inside the (power-of-two) array
                        const int k = m == 3 ? 2 : (m == 7 ? 3 : (m == 15 ? 4 : 5));
                        if (a < b + k || (1 << (a + k)) > Nlds) continue;
                        const double c = model_cost(a, m, b);
                        if (c < best * 0.97) { best = c; best_a = a; best_m = m; best_b = b; }
                    }
        }
        if (best_m == 0) Nlds = h->Npad; // identity: no power-of-two padding needed
        for (size_t i = 0; i < lidx.size(); ++i) {
            const int x = lidx[i];
            lidx[i] = (uint16_t)(x ^ (((x >> best_a) & best_m) << best_b));
        }
        h->lp.swz_a = best_a; h->lp.swz_m = best_m; h->lp.swz_b = best_b; h->lp.Nlds = Nlds;

        // delta tables, one per (orbit, self position), all padded to [S*S][NTP]
        int NTP = 1, SMAX = t->max_species;
        for (int o = 0; o < t->n_orb; ++o) NTP = std::max(NTP, (int)t->orb_tensor_len[o]);
        // one table = S*S*NTP doubles; the stride between tables is padded so that it is
        // not a multiple of the 64-dword LDS bank period (lanes of one wave read the same
        // (pair, base) entry of DIFFERENT tables: an unpadded power-of-two stride makes
        // them all collide on one bank pair)
        size_t tlen = (size_t)SMAX * SMAX * NTP;
        if ((tlen & 1) == 0) tlen += 1;
        std::vector<double> dt(tlen, 0.0); // table 0 = zeros, used by padded slots
        std::vector<LeanSlot> ls((size_t)NSL * 64);
        memset(ls.data(), 0, ls.size() * sizeof(LeanSlot));
        std::map<std::pair<int, int>, uint32_t> doff_of;
        const std::vector<Slot> &sl = slots[class_rep[0]];
        bool ok = true;
        for (size_t q = 0; q < sl.size() && ok; ++q) {
            const Slot &k = sl[q];
            const int o = k.orbit, I = t->orb_nsites[o], Nt = t->orb_tensor_len[o];
            const int32_t *st = t->tensor_indices + t->orb_stride_off[o];
            const double *T = t->interaction_tensors + t->orb_itensor_off[o];
            const int ss = st[k.p];
            const int Sself = k.p == 0 ? Nt / st[0] : st[k.p - 1] / st[k.p];
            const auto key = std::make_pair(o, k.p);
            if (!doff_of.count(key)) {
                doff_of[key] = (uint32_t)dt.size();
                dt.resize(dt.size() + tlen, 0.0);
                double *D = dt.data() + doff_of[key];
                for (int oldc = 0; oldc < Sself; ++oldc)
                    for (int newc = 0; newc < Sself; ++newc)
                        for (int b = 0; b < Nt; ++b) {
                            const int fi = b + ss * newc, ii = b + ss * oldc;
                            if (fi < Nt && ii < Nt) D[((size_t)oldc * SMAX + newc) * NTP + b] = T[fi] - T[ii];
                        }
            }
            const double scale = (double)t->size / t->loc_ratio[k.rec] / (double)t->loc_nrows[k.rec];
            LeanSlot &L = ls[(q / 64) * 64 + (q % 64)];
            L.doff8 = doff_of[key] * 8u;
            int m = 0;
            for (int a = 0; a < I; ++a)
                if (a != k.p) L.stride8[m++] = (uint32_t)st[a] * 8u;
            L.feat = (uint32_t)t->orb_id[o];
            L.live = 1;
            L.w = t->ce_coefs[t->orb_id[o]] * scale;
            L.fs = scale;
            if (dt.size() > 5500) ok = false; // keep the LDS tables within budget
        }
        h->lp.nt8 = (uint32_t)NTP * 8u;
        h->lp.snt8 = (uint32_t)NTP * 8u * (uint32_t)SMAX;
        if (ok) {
            TRY(dev_upload(h, lidx.data(), lidx.size(), &h->lp.idx));
            TRY(dev_upload(h, dt.data(), dt.size(), &h->lp.dt));
            TRY(dev_upload(h, ls.data(), ls.size(), &h->lp.slots));
            h->lp.dt_len = (int)dt.size();
            h->lean_tables = true;
            h->lean_nslot = NSL;
            h->lean_mm = MML;
        }
    }
    return 0;
}

static int build_ref_tables(smolmc_handle *h, const smolmc_tables *t) {
    RefTables &rt = h->rt;
    memset(&rt, 0, sizeof(rt));
    const int n = t->n_orb;
    rt.N = t->num_sites;
    rt.Npad = h->Npad;
    rt.P = t->size;
    rt.num_orbits = t->num_orbits;
    rt.num_corr = t->num_corr;
    rt.n_orb = n;
    rt.Fce = h->Fce;
    rt.F = h->F;
    rt.corr_mode = t->feature_mode == SMOLMC_FEATURES_CORRELATIONS;
    rt.offset = t->offset;
    int nstr = 0;
    long long nct = 0, nit = 0;
    for (int o = 0; o < n; ++o) {
        nstr += t->orb_nsites[o];
        nct += (long long)t->orb_nfunc[o] * t->orb_tensor_len[o];
        nit += t->orb_tensor_len[o];
    }
    const int64_t nloc = t->site_ptr[t->num_sites];
    int64_t nlocidx = 0;
    for (int64_t r = 0; r < nloc; ++r)
        nlocidx += (int64_t)t->loc_nrows[r] * t->orb_nsites[t->loc_orbit[r]];
    TRY(dev_upload(h, t->orb_id, n, &rt.orb_id));
    TRY(dev_upload(h, t->orb_bit_id, n, &rt.orb_bit_id));
    TRY(dev_upload(h, t->orb_nsites, n, &rt.orb_nsites));
    TRY(dev_upload(h, t->orb_nfunc, n, &rt.orb_nfunc));
    TRY(dev_upload(h, t->orb_tensor_len, n, &rt.orb_tensor_len));
    TRY(dev_upload(h, t->orb_stride_off, n, &rt.orb_stride_off));
    TRY(dev_upload(h, t->tensor_indices, nstr, &rt.tensor_indices));
    TRY(dev_upload(h, (const long long *)t->orb_ctensor_off, n, &rt.orb_ctensor_off));
    TRY(dev_upload(h, (const long long *)t->orb_itensor_off, n, &rt.orb_itensor_off));
    TRY(dev_upload(h, t->corr_tensors, nct, &rt.corr_tensors));
    TRY(dev_upload(h, t->interaction_tensors, nit, &rt.interaction_tensors));
    TRY(dev_upload(h, (const long long *)t->full_off, n + 1, &rt.full_off));
    TRY(dev_upload(h, t->full_idx, t->full_off[n], &rt.full_idx));
    TRY(dev_upload(h, (const long long *)t->site_ptr, t->num_sites + 1, &rt.site_ptr));
    TRY(dev_upload(h, t->loc_orbit, nloc, &rt.loc_orbit));
    TRY(dev_upload(h, t->loc_ratio, nloc, &rt.loc_ratio));
    TRY(dev_upload(h, t->loc_nrows, nloc, &rt.loc_nrows));
    TRY(dev_upload(h, (const long long *)t->loc_off, nloc, &rt.loc_off));
    TRY(dev_upload(h, t->loc_idx, nlocidx, &rt.loc_idx));
    rt.has_ewald = t->has_ewald;
    rt.has_mu = t->has_mu;
    if (t->has_ewald) {
        const size_t M = t->ewald_dim;
        rt.ew_W = t->ewald_width;
        rt.ew_M = (int)M;
        TRY(dev_upload(h, t->ewald_inds, (size_t)t->num_sites * t->ewald_width, &rt.ew_inds));
        TRY(dev_upload(h, t->ewald_matrix, M * M, &rt.ew_M_rowmajor));
        std::vector<double> mt(M * M);
        const size_t B = 64;
        for (size_t i0 = 0; i0 < M; i0 += B)
            for (size_t j0 = 0; j0 < M; j0 += B)
                for (size_t i = i0; i < std::min(M, i0 + B); ++i)
                    for (size_t j = j0; j < std::min(M, j0 + B); ++j)
                        mt[j * M + i] = t->ewald_matrix[i * M + j];
        TRY(dev_upload(h, mt.data(), M * M, &rt.ew_Mt));
    }
    if (t->has_mu) {
        rt.mu_W = t->mu_width;
        TRY(dev_upload(h, t->mu_table, (size_t)t->num_sites * t->mu_width, &rt.mu));
    }
    return 0;
}

// Check M[a][b] == q_a q_b G[site_a][site_b] (a, b on different sites) and build G.
static int build_compact_ewald(smolmc_handle *h, const smolmc_tables *t) {
    const int N = t->num_sites, W = t->ewald_width;
    const size_t M = (size_t)t->ewald_dim;
    const double *Mx = t->ewald_matrix, *q = t->ewald_charges;
    std::vector<double> G((size_t)N * N, 0.0), qs((size_t)N * W, 0.0), dg((size_t)N * W, 0.0);
    double mmax = 0;
    for (size_t i = 0; i < M * M; ++i) mmax = std::max(mmax, fabs(Mx[i]));
    const double tol = 1e-12 * std::max(mmax, 1e-300);
    for (int s = 0; s < N; ++s)
        for (int c = 0; c < W; ++c) {
            const int a = t->ewald_inds[(size_t)s * W + c];
            if (a < 0) continue;
            qs[(size_t)s * W + c] = q[a];
            dg[(size_t)s * W + c] = Mx[(size_t)a * M + a];
        }
    bool ok = true;
    for (int s = 0; s < N && ok; ++s)
        for (int u = 0; u < N && ok; ++u) {
            if (s == u) continue;
            double g = 0;
            bool have = false;
            for (int c = 0; c < W && !have; ++c)
                for (int d = 0; d < W && !have; ++d) {
                    const int a = t->ewald_inds[(size_t)s * W + c], b = t->ewald_inds[(size_t)u * W + d];
                    if (a < 0 || b < 0 || q[a] == 0.0 || q[b] == 0.0) continue;
                    g = Mx[(size_t)b * M + a] / (q[a] * q[b]); // the entry ewald.pyx reads: M[i_k, add]
                    have = true;
                }
            for (int c = 0; c < W && ok; ++c)
                for (int d = 0; d < W && ok; ++d) {
                    const int a = t->ewald_inds[(size_t)s * W + c], b = t->ewald_inds[(size_t)u * W + d];
                    if (a < 0 || b < 0) continue;
                    if (fabs(Mx[(size_t)b * M + a] - q[a] * q[b] * g) > tol) ok = false;
                }
            G[(size_t)s * N + u] = g; // row s: kernel between the flipped site s and site u
        }
    if (!ok) return 0; // not of product form: keep the dense rows
    // split sites into changeable ones and single-species ("frozen") ones
    std::vector<int> act;
    std::vector<char> frozen(N, 0);
    for (int s = 0; s < N; ++s) {
        int nvalid = 0, ncodes = 0;
        for (int c = 0; c < W; ++c) nvalid += t->ewald_inds[(size_t)s * W + c] >= 0;
        (void)ncodes;
        // a site is frozen when it is in no active sublattice (its code never changes) and
        // carries exactly one Ewald species (code 0)
        bool in_active = false;
        for (int64_t i = 0; i < t->sub_site_ptr[t->n_sublattices] && !in_active; ++i)
            in_active = t->sub_active_sites[i] == s;
        frozen[s] = (!in_active && nvalid == 1 && t->ewald_inds[(size_t)s * W] >= 0) ? 1 : 0;
        if (!frozen[s]) act.push_back(s);
    }
    const size_t na = act.size();
    std::vector<double> Gact((size_t)N * std::max<size_t>(na, 1), 0.0), fz(N, 0.0);
    for (int s = 0; s < N; ++s) {
        for (size_t j = 0; j < na; ++j) Gact[(size_t)s * na + j] = G[(size_t)s * N + act[j]];
        double c = 0;
        for (int u = 0; u < N; ++u)
            if (frozen[u] && u != s) c += qs[(size_t)u * W] * G[(size_t)s * N + u];
        fz[s] = c;
    }
    TRY(dev_upload(h, act.data(), act.size(), &h->kp.ew_act));
    TRY(dev_upload(h, fz.data(), fz.size(), &h->kp.ew_frozen));
    h->kp.ew_nact = (int)na;
    h->kp.ew_act_base = na ? act[0] : -1;
    for (size_t j = 0; j < na; ++j)
        if (act[j] != act[0] + (int)j) h->kp.ew_act_base = -1;
    G.swap(Gact);
    TRY(dev_upload(h, G.data(), G.size(), &h->kp.ew_G));
    TRY(dev_upload(h, qs.data(), qs.size(), &h->kp.ew_qs));
    TRY(dev_upload(h, dg.data(), dg.size(), &h->kp.ew_dg));
    h->kp.ew_compact = 1;
    return 0;
}

extern "C" int smolmc_abi_version(void) { return SMOLMC_ABI_VERSION; }
extern "C" const char *smolmc_last_error(void) { return g_err.c_str(); }

extern "C" int smolmc_create(const smolmc_tables *t, const smolmc_config *cfg, smolmc_handle **out) {
    if (!t || !cfg || !out) return fail("null argument");
    if (cfg->n_replicas <= 0) return fail("n_replicas must be positive");
    if (t->max_species > 255) return fail("more than 255 species codes per site");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail("no HIP device available: the smol_amd engine requires an AMD GPU (there is no "
                    "CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail("invalid device ordinal");
    smolmc_handle *h = new smolmc_handle();
    h->cfg = *cfg;
    h->device = cfg->device;
    auto bail = [&](int rc) {
        smolmc_destroy(h);
        return rc;
    };
    if (hipSetDevice(h->device) != hipSuccess) return bail(fail("hipSetDevice failed"));
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail("hipStreamCreate failed"));
    h->own_stream = true;
    hipEventCreate(&h->ev0);
    hipEventCreate(&h->ev1);
    h->R = cfg->n_replicas;
    h->N = t->num_sites;
    h->Npad = (t->num_sites + 15) / 16 * 16;
    h->Fce = num_ce_features(t);
    h->F = h->Fce + (t->has_ewald ? 1 : 0) + (t->has_mu ? 1 : 0);
    h->natural.assign(t->ce_coefs, t->ce_coefs + h->Fce);
    if (t->has_ewald) h->natural.push_back(t->ewald_coef);
    if (t->has_mu) h->natural.push_back(-1.0);
    const bool wl = cfg->kernel_type == SMOLMC_KERNEL_WANGLANDAU;
    if (wl) {
        if (cfg->wl_min_enthalpy > cfg->wl_max_enthalpy)
            return bail(fail("min_enthalpy can not be larger than max_enthalpy.")); // wanglandau.py:82
        if ((cfg->wl_max_enthalpy - cfg->wl_min_enthalpy) / cfg->wl_bin_size <= 1)
            return bail(fail("The values provided for min and max enthalpy and bin sizer result in a "
                             "single bin!"));
        if (cfg->wl_mod_factor <= 0) return bail(fail("mod_factor must be greater than 0."));
        h->L = (int)ceil((cfg->wl_max_enthalpy - cfg->wl_min_enthalpy) / cfg->wl_bin_size);
    }
    memset(&h->kp, 0, sizeof(KParams));
    memset(&h->smp, 0, sizeof(SampleBufs));
    KParams &kp = h->kp;
    if (int rc = build_mc_tables(h, t)) return bail(rc);
    if (int rc = build_ref_tables(h, t)) return bail(rc);
    kp.N = h->N;
    kp.Npad = h->Npad;
    kp.Fce = h->Fce;
    kp.F = h->F;
    kp.step_type = cfg->step_type;
    kp.corr_mode = h->rt.corr_mode;
    kp.has_ewald = t->has_ewald;
    kp.has_mu = t->has_mu;
    kp.ew_W = h->rt.ew_W;
    kp.ew_M = h->rt.ew_M;
    kp.mu_W = h->rt.mu_W;
    kp.ew_inds = h->rt.ew_inds;
    kp.ew_Mt = h->rt.ew_Mt;
    kp.ew_coef = t->ewald_coef;
    kp.mu = h->rt.mu;
    if (t->has_ewald && t->ewald_charges && getenv("SMOLMC_DENSE_EWALD") == nullptr)
        if (int rc = build_compact_ewald(h, t)) return bail(rc);
    // sublattices
    {
        const int ns = t->n_sublattices;
        if (ns <= 0) return bail(fail("no active sublattice"));
        std::vector<int> ptr(ns + 1), cptr(ns + 1), base(ns);
        std::vector<double> cum(ns);
        double c = 0;
        for (int s = 0; s <= ns; ++s) {
            ptr[s] = (int)t->sub_site_ptr[s];
            cptr[s] = (int)t->sub_code_ptr[s];
        }
        for (int s = 0; s < ns; ++s) {
            c += t->sub_probs[s];
            cum[s] = c;
            const int a = ptr[s], b = ptr[s + 1];
            if (b <= a) return bail(fail("empty active sublattice"));
            bool contig = true;
            for (int i = a + 1; i < b; ++i)
                if (t->sub_active_sites[i] != t->sub_active_sites[a] + (i - a)) contig = false;
            base[s] = contig ? t->sub_active_sites[a] : -1;
            for (int i = a; i < b; ++i)
                if (t->sub_active_sites[i] < 0 || t->sub_active_sites[i] >= t->num_sites)
                    return bail(fail("sublattice site index out of range"));
        }
        kp.nsub = ns;
        if (dev_upload(h, ptr.data(), ptr.size(), &kp.sub_ptr) ||
            dev_upload(h, t->sub_active_sites, (size_t)ptr[ns], &kp.sub_sites) ||
            dev_upload(h, base.data(), base.size(), &kp.sub_base) ||
            dev_upload(h, cptr.data(), cptr.size(), &kp.sub_code_ptr) ||
            dev_upload(h, t->sub_codes, (size_t)cptr[ns], &kp.sub_codes) ||
            dev_upload(h, cum.data(), cum.size(), &kp.sub_cum))
            return bail(1);
    }
    // walker state
    const size_t R = h->R;
    kp.R = h->R;
    double *dbeta = nullptr;
    uint64_t *dseeds = nullptr;
    if (dev_alloc(h, R * h->Npad, &kp.occ) || dev_alloc(h, R, &kp.enthalpy) ||
        dev_alloc(h, R * h->F, &kp.features) || dev_alloc(h, R, &dbeta) || dev_alloc(h, R, &dseeds) ||
        dev_alloc(h, R, &kp.nsteps) || dev_alloc(h, R, &kp.nacc) || dev_alloc(h, R, &kp.last_acc))
        return bail(1);
    kp.beta = dbeta;
    h->d_beta = dbeta;
    kp.seeds = dseeds;
    if (dev_upload(h, h->natural.data(), h->natural.size(), (const double **)&h->d_natural))
        return bail(1);
    if (wl) {
        kp.L = h->L;
        kp.wl_min = cfg->wl_min_enthalpy;
        kp.wl_max = cfg->wl_max_enthalpy;
        kp.wl_bin = cfg->wl_bin_size;
        kp.wl_flat = cfg->wl_flatness;
        kp.wl_div = cfg->wl_mod_divisor;
        kp.wl_check = cfg->wl_check_period;
        kp.wl_update = cfg->wl_update_period;
        if (kp.wl_check <= 0 || kp.wl_update <= 0) return bail(fail("WL periods must be positive"));
        if (h->F > 64) return bail(fail("Wang-Landau supports at most 64 features"));
        if (dev_alloc(h, R * h->L, &kp.wl_entropy) || dev_alloc(h, R * h->L, &kp.wl_hist) ||
            dev_alloc(h, R * h->L, &kp.wl_occur) || dev_alloc(h, R * h->L * h->F, &kp.wl_meanf) ||
            dev_alloc(h, R, &kp.wl_m) || dev_alloc(h, R, &kp.wl_counter))
            return bail(1);
        std::vector<double> m0(R, cfg->wl_mod_factor);
        if (hipMemcpy(kp.wl_m, m0.data(), R * 8, hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail("hipMemcpy failed"));
    }
    // LDS layout
    size_t tb = (size_t)kp.nclasses * kp.Cpad * (16 + 16 + 8) + (size_t)(kp.xt_len + kp.ft_len) * 8 +
                (size_t)((kp.nclasses + 3) & ~3) * 4 + (kp.nclasses > 1 ? (size_t)h->N : 0);
    tb = (tb + 15) / 16 * 16;
    size_t pw = (size_t)h->Npad;
    if (wl)
        pw += (size_t)h->L * 16 + (size_t)h->F * 8;
    else
        pw += (size_t)h->Fce * 64 * 8;
    pw = (pw + 15) / 16 * 16;
    kp.lds_tables = (int)tb;
    kp.lds_per_wave = (int)pw;
    h->waves_per_block = 4;
    while (h->waves_per_block > 1 && tb + pw * h->waves_per_block > 160 * 1024) h->waves_per_block /= 2;
    h->lds_bytes = tb + pw * h->waves_per_block;
    if (h->lds_bytes > 160 * 1024)
        return bail(fail("model does not fit the 160 KiB LDS budget (tables + one chain)"));
    // lean-kernel eligibility (everything else runs mc_kernel)
    {
        bool lean = h->lean_tables && h->F <= 64 && (!wl || (!t->has_ewald && !t->has_mu)) &&
                    (!t->has_ewald || kp.ew_compact) && t->n_sublattices == 1 &&
                    getenv("SMOLMC_FORCE_GENERAL") == nullptr;
        int sbase = -1, nact = 0, nc = 0;
        std::vector<double> mu_row;
        if (lean) {
            nact = (int)(t->sub_site_ptr[1] - t->sub_site_ptr[0]);
            nc = (int)(t->sub_code_ptr[1] - t->sub_code_ptr[0]);
            sbase = t->sub_active_sites[0];
            for (int i = 0; i < nact; ++i)
                if (t->sub_active_sites[i] != sbase + i) lean = false;
            for (int c = 0; c < nc; ++c)
                if (t->sub_codes[c] != c) lean = false;
            if (nc < 2 || nc > 8) lean = false;
        }
        if (lean && t->has_mu) {
            if (t->mu_width < nc) lean = false;
            for (int c = 0; lean && c < nc; ++c) mu_row.push_back(t->mu_table[(size_t)sbase * t->mu_width + c]);
            for (int i = 0; lean && i < nact; ++i)
                for (int c = 0; c < nc; ++c)
                    if (t->mu_table[(size_t)(sbase + i) * t->mu_width + c] != mu_row[c]) lean = false;
        }
        if (lean) {
            LeanParams &lp = h->lp;
            if (t->has_mu && dev_upload(h, mu_row.data(), mu_row.size(), &lp.mu_row)) return bail(1);
            lp.occ = kp.occ;
            lp.enthalpy = kp.enthalpy;
            lp.features = kp.features;
            lp.beta = kp.beta;
            lp.seeds = kp.seeds;
            lp.nsteps = kp.nsteps;
            lp.nacc = kp.nacc;
            lp.last_acc = kp.last_acc;
            lp.R = h->R;
            lp.N = h->N;
            lp.Npad = h->Npad;
            lp.F = h->F;
            lp.Fce = h->Fce;
            lp.sbase = sbase;
            lp.nact = nact;
            lp.ncodes = nc;
            if (t->has_ewald) {
                lp.ew_W = kp.ew_W;
                lp.ew_nact = kp.ew_nact;
                lp.ew_act_base = kp.ew_act_base;
                lp.ew_act = kp.ew_act;
                lp.ew_G = kp.ew_G;
                lp.ew_qs = kp.ew_qs;
                lp.ew_dg = kp.ew_dg;
                lp.ew_frozen = kp.ew_frozen;
                lp.ew_coef = kp.ew_coef;
            }
            if (cfg->step_type == SMOLMC_STEP_TABLE_FLIP) {
                if (dev_upload(h, t->flip_table, (size_t)t->n_flip_vectors * nc, &lp.tf_table) ||
                    dev_upload(h, t->flip_weights, (size_t)2 * t->n_flip_vectors, &lp.tf_w))
                    return bail(1);
                lp.tf_n = t->n_flip_vectors;
                lp.tf_sw = t->swap_weight;
            }
            if (wl) {
                lp.wl.L = h->L;
                lp.wl.vmin = kp.wl_min; lp.wl.vmax = kp.wl_max; lp.wl.bin = kp.wl_bin;
                lp.wl.flat = kp.wl_flat; lp.wl.div = kp.wl_div;
                lp.wl.check = kp.wl_check; lp.wl.update = kp.wl_update;
                lp.wl.entropy = kp.wl_entropy; lp.wl.hist = kp.wl_hist; lp.wl.occur = kp.wl_occur;
                lp.wl.meanf = kp.wl_meanf; lp.wl.m = kp.wl_m; lp.wl.counter = kp.wl_counter;
            }
            h->lean_lds = ((size_t)lp.dt_len + 8) * 8 +
                          (size_t)4 * (lp.Nlds + 64 * 8 + 64 + (wl ? (size_t)h->L * 16 : 0));
            if (h->lean_lds > 150 * 1024) lean = false;
        }
        h->lean = lean;
        if (cfg->step_type == SMOLMC_STEP_TABLE_FLIP) {
            if (t->n_flip_vectors <= 0 || !t->flip_table || !t->flip_weights)
                return bail(fail("TableFlip needs a flip table (CompositionSpace.flip_table, "
                                 "smol/moca/composition/space.py:404-429)"));
            if (!lean || wl)
                return bail(fail("TableFlip is implemented for single-class, single-sublattice "
                                 "Metropolis models (the lean path) only"));
            if (!(t->swap_weight >= 0.0 && t->swap_weight < 1.0))
                return bail(fail("swap_weight must be in [0, 1)"));
        }
    }
    *out = h;
    return 0;
}

extern "C" int smolmc_destroy(smolmc_handle *h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (void *p : h->allocs) hipFree(p);
    if (h->d_eval_occ) hipFree(h->d_eval_occ);
    free_samples(h);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

extern "C" int smolmc_num_features(const smolmc_handle *h) { return h ? h->F : -1; }
extern "C" int smolmc_wl_num_levels(const smolmc_handle *h) { return h ? h->L : -1; }
extern "C" int smolmc_natural_parameters(const smolmc_handle *h, double *out) {
    if (!h || !out) return fail("null argument");
    memcpy(out, h->natural.data(), h->natural.size() * 8);
    return 0;
}

extern "C" int smolmc_set_stream(smolmc_handle *h, void *stream) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->own_stream) hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)stream;
    h->own_stream = false;
    return 0;
}

static int launch_eval_full(smolmc_handle *h, const uint8_t *d_occ8, int nocc, double *d_out) {
    hipLaunchKernelGGL(eval_full_kernel, dim3(nocc), dim3(256), 0, h->stream, h->rt, d_occ8, d_out);
    HIPCHK(hipGetLastError());
    return 0;
}

static int upload_occ(smolmc_handle *h, const int32_t *occ, size_t nocc, uint8_t *d_occ8) {
    const size_t n32 = nocc * h->N;
    for (size_t i = 0; i < n32; ++i)
        if (occ[i] < 0 || occ[i] > 255) return fail("occupancy code out of range [0, 255]");
    int *d32 = nullptr;
    HIPCHK(hipMalloc((void **)&d32, std::max<size_t>(n32 * 4, 16)));
    hipError_t e = hipMemcpyAsync(d32, occ, n32 * 4, hipMemcpyHostToDevice, h->stream);
    const size_t total = nocc * h->Npad;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pack_occ_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream,
                           d32, d_occ8, h->N, h->Npad, total);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d32);
    if (e != hipSuccess) return fail(std::string("occupancy upload: ") + hipGetErrorString(e));
    return 0;
}

static void set_betas(smolmc_handle *h, const double *temperature, std::vector<double> &beta) {
    beta.resize(h->R);
    for (int r = 0; r < h->R; ++r) {
        const double T = temperature ? temperature[r] : 0.0;
        beta[r] = 1.0 / (SMOLMC_KB * T); // ThermalKernelMixin (kernel/base.py:398)
    }
}

extern "C" int smolmc_set_state(smolmc_handle *h, const int32_t *occ, const uint64_t *seeds,
                                const double *temperature, int reset_aux) {
    if (!h || !occ) return fail("null argument");
    HIPCHK(hipSetDevice(h->device));
    KParams &kp = h->kp;
    const size_t R = h->R;
    TRY(upload_occ(h, occ, R, kp.occ));
    std::vector<double> beta;
    set_betas(h, temperature, beta);
    HIPCHK(hipMemcpy(h->d_beta, beta.data(), R * 8, hipMemcpyHostToDevice));
    if (reset_aux) {
        std::vector<uint64_t> sd(R);
        for (size_t r = 0; r < R; ++r) sd[r] = seeds ? seeds[r] : (uint64_t)r;
        HIPCHK(hipMemcpy((void *)kp.seeds, sd.data(), R * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemsetAsync(kp.nsteps, 0, R * 8, h->stream));
        HIPCHK(hipMemsetAsync(kp.nacc, 0, R * 8, h->stream));
    }
    HIPCHK(hipMemsetAsync(kp.last_acc, 1, R, h->stream));
    TRY(launch_eval_full(h, kp.occ, (int)R, kp.features));
    hipLaunchKernelGGL(dot_features_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, h->stream,
                       kp.features, h->d_natural, kp.enthalpy, (int)R, h->F);
    HIPCHK(hipGetLastError());
    if (h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU && reset_aux) {
        const size_t RL = R * h->L;
        HIPCHK(hipMemsetAsync(kp.wl_entropy, 0, RL * 8, h->stream));
        HIPCHK(hipMemsetAsync(kp.wl_hist, 0, RL * 8, h->stream));
        HIPCHK(hipMemsetAsync(kp.wl_occur, 0, RL * 8, h->stream));
        HIPCHK(hipMemsetAsync(kp.wl_meanf, 0, RL * h->F * 8, h->stream));
        HIPCHK(hipMemsetAsync(kp.wl_counter, 0, R * 8, h->stream));
        std::vector<double> m0(R, h->cfg.wl_mod_factor);
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(kp.wl_m, m0.data(), R * 8, hipMemcpyHostToDevice));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int smolmc_set_temperature(smolmc_handle *h, const double *temperature) {
    if (!h || !temperature) return fail("null argument");
    HIPCHK(hipSetDevice(h->device));
    std::vector<double> beta;
    set_betas(h, temperature, beta);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->d_beta, beta.data(), (size_t)h->R * 8, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int smolmc_sync(smolmc_handle *h) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int smolmc_get_state(smolmc_handle *h, int32_t *occ, double *features, double *enthalpy,
                                uint64_t *n_accepted, uint64_t *n_steps, uint8_t *last_accepted) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t R = h->R;
    KParams &kp = h->kp;
    if (occ) {
        int *d32 = nullptr;
        const size_t total = R * h->N;
        HIPCHK(hipMalloc((void **)&d32, total * 4));
        hipLaunchKernelGGL(unpack_occ_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           h->stream, kp.occ, d32, h->N, h->Npad, total);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(occ, d32, total * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        hipFree(d32);
        if (e != hipSuccess) return fail(std::string("occupancy download: ") + hipGetErrorString(e));
    }
    if (features) HIPCHK(hipMemcpy(features, kp.features, R * h->F * 8, hipMemcpyDeviceToHost));
    if (enthalpy) HIPCHK(hipMemcpy(enthalpy, kp.enthalpy, R * 8, hipMemcpyDeviceToHost));
    if (n_accepted) HIPCHK(hipMemcpy(n_accepted, kp.nacc, R * 8, hipMemcpyDeviceToHost));
    if (n_steps) HIPCHK(hipMemcpy(n_steps, kp.nsteps, R * 8, hipMemcpyDeviceToHost));
    if (last_accepted) HIPCHK(hipMemcpy(last_accepted, kp.last_acc, R, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int smolmc_get_wl(smolmc_handle *h, double *entropy, int64_t *histogram,
                             int64_t *occurrences, double *mean_features, double *mod_factor) {
    if (!h) return fail("null handle");
    if (h->cfg.kernel_type != SMOLMC_KERNEL_WANGLANDAU) return fail("handle is not a Wang-Landau kernel");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t RL = (size_t)h->R * h->L;
    KParams &kp = h->kp;
    if (entropy) HIPCHK(hipMemcpy(entropy, kp.wl_entropy, RL * 8, hipMemcpyDeviceToHost));
    if (histogram) HIPCHK(hipMemcpy(histogram, kp.wl_hist, RL * 8, hipMemcpyDeviceToHost));
    if (occurrences) HIPCHK(hipMemcpy(occurrences, kp.wl_occur, RL * 8, hipMemcpyDeviceToHost));
    if (mean_features)
        HIPCHK(hipMemcpy(mean_features, kp.wl_meanf, RL * h->F * 8, hipMemcpyDeviceToHost));
    if (mod_factor) HIPCHK(hipMemcpy(mod_factor, kp.wl_m, (size_t)h->R * 8, hipMemcpyDeviceToHost));
    return 0;
}

// ---- kernel dispatch ----------------------------------------------------------
template <typename IdxT, int NSLOT, int MM, bool GENERIC, bool WL>
static int launch_mc_inst(smolmc_handle *h, const KParams &kp, int replay) {
    auto kern = mc_kernel<IdxT, NSLOT, MM, GENERIC, WL>;
    if (h->lds_bytes > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h->lds_bytes));
    const int wpb = h->waves_per_block;
    const unsigned grid = (unsigned)((h->R + wpb - 1) / wpb);
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wpb), h->lds_bytes, h->stream, kp, replay);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return 0;
}

template <typename IdxT, int NSLOT, bool GENERIC, bool WL>
static int launch_mc_mm(smolmc_handle *h, const KParams &kp, int replay) {
    if (GENERIC) {
        if (h->mm == 3) return launch_mc_inst<IdxT, NSLOT, 3, GENERIC, WL>(h, kp, replay);
        return launch_mc_inst<IdxT, NSLOT, 6, GENERIC, WL>(h, kp, replay);
    }
    if (h->mm == 2) return launch_mc_inst<IdxT, NSLOT, 2, GENERIC, WL>(h, kp, replay);
    if (h->mm == 3) return launch_mc_inst<IdxT, NSLOT, 3, GENERIC, WL>(h, kp, replay);
    return launch_mc_inst<IdxT, NSLOT, 5, GENERIC, WL>(h, kp, replay);
}

template <typename IdxT, bool GENERIC, bool WL>
static int launch_mc_slot(smolmc_handle *h, const KParams &kp, int replay) {
    if (h->nslot == 2) return launch_mc_mm<IdxT, 2, GENERIC, WL>(h, kp, replay);
#ifndef SMOLMC_FAST_BUILD
    if (h->nslot == 4) return launch_mc_mm<IdxT, 4, GENERIC, WL>(h, kp, replay);
    return launch_mc_mm<IdxT, 8, GENERIC, WL>(h, kp, replay);
#else
    return fail("this build only carries NSLOT=2 kernels");
#endif
}

static int launch_mc(smolmc_handle *h, const KParams &kp, int replay) {
    const bool wl = h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU;
    if (h->generic)
        return wl ? launch_mc_slot<int32_t, true, true>(h, kp, replay)
                  : launch_mc_slot<int32_t, true, false>(h, kp, replay);
    if (h->idx16)
        return wl ? launch_mc_slot<uint16_t, false, true>(h, kp, replay)
                  : launch_mc_slot<uint16_t, false, false>(h, kp, replay);
    return wl ? launch_mc_slot<int32_t, false, true>(h, kp, replay)
              : launch_mc_slot<int32_t, false, false>(h, kp, replay);
}

template <int NSLOT, int MM, int STEP, bool MU, bool EW, bool WL>
static int launch_lean_inst(smolmc_handle *h, const LeanParams &lp) {
    const unsigned grid = (unsigned)((h->R + 3) / 4);
    auto kern = mc_lean_kernel<NSLOT, MM, STEP, MU, EW, WL>;
    if (h->lean_lds > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h->lean_lds));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), h->lean_lds, h->stream, lp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return 0;
}
template <int NSLOT, int MM, int STEP>
static int launch_lean_me(smolmc_handle *h, const LeanParams &lp) {
    const bool mu = lp.mu_row != nullptr, ew = lp.ew_G != nullptr;
    if (h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU)
        return launch_lean_inst<NSLOT, MM, STEP, false, false, true>(h, lp);
    if (ew)
        return mu ? launch_lean_inst<NSLOT, MM, STEP, true, true, false>(h, lp)
                  : launch_lean_inst<NSLOT, MM, STEP, false, true, false>(h, lp);
    return mu ? launch_lean_inst<NSLOT, MM, STEP, true, false, false>(h, lp)
              : launch_lean_inst<NSLOT, MM, STEP, false, false, false>(h, lp);
}
template <int NSLOT, int MM>
static int launch_lean_nm(smolmc_handle *h, const LeanParams &lp) {
    if (h->cfg.step_type == SMOLMC_STEP_SWAP) return launch_lean_me<NSLOT, MM, SMOLMC_STEP_SWAP>(h, lp);
    return launch_lean_me<NSLOT, MM, SMOLMC_STEP_FLIP>(h, lp);
}
template <int NSLOT, int MM>
static int launch_table_inst(smolmc_handle *h, const LeanParams &lp) {
    const unsigned grid = (unsigned)((h->R + 3) / 4);
    auto kern = mc_table_kernel<NSLOT, MM>;
    if (h->lean_lds > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h->lean_lds));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), h->lean_lds, h->stream, lp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return 0;
}

static int launch_lean(smolmc_handle *h, LeanParams lp, int64_t nsteps) {
    lp.steps = nsteps;
    if (h->cfg.step_type == SMOLMC_STEP_TABLE_FLIP) {
        if (h->lean_nslot == 2)
            return h->lean_mm == 2 ? launch_table_inst<2, 2>(h, lp) : launch_table_inst<2, 3>(h, lp);
        return h->lean_mm == 2 ? launch_table_inst<4, 2>(h, lp) : launch_table_inst<4, 3>(h, lp);
    }
    if (h->lean_nslot == 2)
        return h->lean_mm == 2 ? launch_lean_nm<2, 2>(h, lp) : launch_lean_nm<2, 3>(h, lp);
    return h->lean_mm == 2 ? launch_lean_nm<4, 2>(h, lp) : launch_lean_nm<4, 3>(h, lp);
}

static void free_samples(smolmc_handle *h) {
    if (h->smp.H) hipFree(h->smp.H);
    if (h->smp.feat) hipFree(h->smp.feat);
    if (h->smp.acc) hipFree(h->smp.acc);
    if (h->smp.occ) hipFree(h->smp.occ);
    memset(&h->smp, 0, sizeof(SampleBufs));
    h->smp_n = 0;
    h->smp_has_occ = false;
}

static int run_steps(smolmc_handle *h, int64_t nsteps, const SampleBufs &smp) {
    if (h->lean) {
        LeanParams lp = h->lp;
        lp.smp = smp;
        return launch_lean(h, lp, nsteps);
    }
    KParams kp = h->kp;
    kp.steps_to_run = nsteps;
    kp.smp = smp;
    return launch_mc(h, kp, 0);
}

extern "C" int smolmc_run(smolmc_handle *h, int64_t nsteps) {
    if (!h) return fail("null handle");
    if (nsteps < 0) return fail("nsteps must be non-negative");
    if (nsteps == 0) return 0;
    HIPCHK(hipSetDevice(h->device));
    SampleBufs none;
    memset(&none, 0, sizeof(none));
    return run_steps(h, nsteps, none);
}

extern "C" int smolmc_run_sampled(smolmc_handle *h, int64_t nsamples, int64_t thin_by, int flags) {
    if (!h) return fail("null handle");
    if (nsamples <= 0 || thin_by <= 0) return fail("nsamples and thin_by must be positive");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    free_samples(h);
    const size_t rows = (size_t)nsamples * h->R;
    HIPCHK(hipMalloc((void **)&h->smp.H, rows * 8));
    HIPCHK(hipMalloc((void **)&h->smp.feat, rows * h->F * 8));
    HIPCHK(hipMalloc((void **)&h->smp.acc, rows));
    if (flags & 1) {
        HIPCHK(hipMalloc((void **)&h->smp.occ, rows * h->Npad));
        h->smp_has_occ = true;
    }
    h->smp.every = thin_by;
    h->smp_n = nsamples;
    return run_steps(h, nsamples * thin_by, h->smp);
}

extern "C" int smolmc_get_samples(smolmc_handle *h, double *enthalpy, double *features,
                                  uint8_t *accepted, int32_t *occupancy) {
    if (!h) return fail("null handle");
    if (h->smp_n == 0) return fail("no samples recorded: call smolmc_run_sampled first");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t rows = (size_t)h->smp_n * h->R;
    if (enthalpy) HIPCHK(hipMemcpy(enthalpy, h->smp.H, rows * 8, hipMemcpyDeviceToHost));
    if (features) HIPCHK(hipMemcpy(features, h->smp.feat, rows * h->F * 8, hipMemcpyDeviceToHost));
    if (accepted) HIPCHK(hipMemcpy(accepted, h->smp.acc, rows, hipMemcpyDeviceToHost));
    if (occupancy) {
        if (!h->smp_has_occ) return fail("occupancies were not recorded (flags bit 0)");
        int *d32 = nullptr;
        const size_t total = rows * h->N;
        HIPCHK(hipMalloc((void **)&d32, total * 4));
        hipLaunchKernelGGL(unpack_occ_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           h->stream, h->smp.occ, d32, h->N, h->Npad, total);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(occupancy, d32, total * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        hipFree(d32);
        if (e != hipSuccess) return fail(std::string("sample download: ") + hipGetErrorString(e));
    }
    return 0;
}

extern "C" int smolmc_replay(smolmc_handle *h, int64_t nsteps, const int32_t *steps,
                             const double *uniforms, uint8_t *accepted_out, double *enthalpy_out) {
    if (!h || !steps || !uniforms) return fail("null argument");
    if (nsteps <= 0) return 0;
    HIPCHK(hipSetDevice(h->device));
    const size_t n = (size_t)h->R * nsteps;
    for (size_t i = 0; i < n; ++i) {
        const int32_t *st = steps + i * 4;
        for (int f = 0; f < 2; ++f)
            if (st[2 * f] >= h->N || (st[2 * f] >= 0 && (st[2 * f + 1] < 0 || st[2 * f + 1] > 255)))
                return fail("replay step out of range");
    }
    int *d_steps = nullptr;
    double *d_u = nullptr, *d_H = nullptr;
    uint8_t *d_acc = nullptr;
    hipError_t e = hipMalloc((void **)&d_steps, n * 16);
    if (e == hipSuccess) e = hipMalloc((void **)&d_u, n * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_H, n * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_acc, n);
    if (e == hipSuccess) e = hipMemcpy(d_steps, steps, n * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_u, uniforms, n * 8, hipMemcpyHostToDevice);
    int rc = 0;
    if (e == hipSuccess) {
        KParams kp = h->kp;
        kp.steps_to_run = nsteps;
        kp.rp_steps = d_steps;
        kp.rp_u = d_u;
        kp.rp_acc = d_acc;
        kp.rp_H = d_H;
        rc = launch_mc(h, kp, 1);
        if (!rc) e = hipStreamSynchronize(h->stream);
        if (!rc && e == hipSuccess && accepted_out) e = hipMemcpy(accepted_out, d_acc, n, hipMemcpyDeviceToHost);
        if (!rc && e == hipSuccess && enthalpy_out) e = hipMemcpy(enthalpy_out, d_H, n * 8, hipMemcpyDeviceToHost);
    }
    hipFree(d_steps);
    hipFree(d_u);
    hipFree(d_H);
    hipFree(d_acc);
    if (rc) return rc;
    if (e != hipSuccess) return fail(std::string("replay: ") + hipGetErrorString(e));
    return 0;
}

extern "C" int smolmc_last_kernel_ms(smolmc_handle *h, float *ms) {
    if (!h || !ms) return fail("null argument");
    if (!h->timed) return fail("no kernel has been launched yet");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return 0;
}

static int ensure_eval_occ(smolmc_handle *h, size_t nocc) {
    const size_t need = nocc * h->Npad;
    if (need > h->eval_occ_cap) {
        if (h->d_eval_occ) hipFree(h->d_eval_occ);
        h->d_eval_occ = nullptr;
        h->eval_occ_cap = 0;
        HIPCHK(hipMalloc((void **)&h->d_eval_occ, need));
        h->eval_occ_cap = need;
    }
    return 0;
}

extern "C" int smolmc_eval_full(smolmc_handle *h, const int32_t *occ, int nocc, double *features) {
    if (!h || !occ || !features) return fail("null argument");
    if (nocc <= 0) return 0;
    HIPCHK(hipSetDevice(h->device));
    TRY(ensure_eval_occ(h, nocc));
    TRY(upload_occ(h, occ, nocc, h->d_eval_occ));
    double *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_out, (size_t)nocc * h->F * 8));
    int rc = launch_eval_full(h, h->d_eval_occ, nocc, d_out);
    hipError_t e = hipStreamSynchronize(h->stream);
    if (!rc && e == hipSuccess) e = hipMemcpy(features, d_out, (size_t)nocc * h->F * 8, hipMemcpyDeviceToHost);
    hipFree(d_out);
    if (rc) return rc;
    if (e != hipSuccess) return fail(std::string("eval_full: ") + hipGetErrorString(e));
    return 0;
}

extern "C" int smolmc_eval_delta(smolmc_handle *h, const int32_t *occ, const int32_t *flips, int nstep,
                                 double *dfeatures) {
    if (!h || !occ || !flips || !dfeatures) return fail("null argument");
    if (nstep <= 0) return 0;
    HIPCHK(hipSetDevice(h->device));
    for (int i = 0; i < nstep; ++i)
        for (int f = 0; f < 2; ++f) {
            const int s = flips[i * 4 + 2 * f], c = flips[i * 4 + 2 * f + 1];
            if (s >= h->N || (s >= 0 && (c < 0 || c > 255))) return fail("flip out of range");
        }
    TRY(ensure_eval_occ(h, 1));
    TRY(upload_occ(h, occ, 1, h->d_eval_occ));
    int *d_fl = nullptr;
    double *d_out = nullptr;
    hipError_t e = hipMalloc((void **)&d_fl, (size_t)nstep * 16);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, (size_t)nstep * h->F * 8);
    if (e == hipSuccess) e = hipMemcpy(d_fl, flips, (size_t)nstep * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(eval_delta_kernel, dim3(nstep), dim3(64), 0, h->stream, h->rt, h->d_eval_occ,
                           d_fl, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipMemcpy(dfeatures, d_out, (size_t)nstep * h->F * 8, hipMemcpyDeviceToHost);
    hipFree(d_fl);
    hipFree(d_out);
    if (e != hipSuccess) return fail(std::string("eval_delta: ") + hipGetErrorString(e));
    return 0;
}

extern "C" int smolmc_export_enthalpy_dev(smolmc_handle *h, double *dst_dev) {
    if (!h || !dst_dev) return fail("null argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(dst_dev, h->kp.enthalpy, (size_t)h->R * 8, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

__global__ void beta_from_T_kernel(const double *T, double *beta, int R, double kB) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) beta[r] = 1.0 / (kB * T[r]);
}

extern "C" int smolmc_import_temperature_dev(smolmc_handle *h, const double *src_dev) {
    if (!h || !src_dev) return fail("null argument");
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(beta_from_T_kernel, dim3((h->R + 63) / 64), dim3(64), 0, h->stream, src_dev,
                       h->d_beta, h->R, SMOLMC_KB);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
