// mc_general.h -- the general Monte-Carlo kernel (mc_kernel) and its launch templates.
#pragma once
#include "smolmc_common.h"

// Kernel arguments that only some paths of the step loop need are re-read from the kernel-argument
// segment where they are used (see rare_params in smolmc_common.h: held in SGPRs across the loop
// they push the hot path into SGPR spills -- this kernel had 260-300 of them).  Valid in kernels
// whose first argument is the KParams block.
typedef const KParams __attribute__((address_space(4))) *KParamsKernarg;
__device__ __forceinline__ KParamsKernarg gen_params() {
    KParamsKernarg p = (KParamsKernarg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
#define PK (*gen_params())

// what every step needs of the index table, hoisted once
struct GenHot {
    const void *idx;
    int Cpad;
    size_t row_stride; // Mmax * Cpad
};

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
    long long *wl_O; // WL: occurrences [L]
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
__device__ __forceinline__ double eval_slot(const GenHot &P, const Lds &L, int cls, int c, int s,
                                            int oldc, int newc, int ps, int pc, int &ind_i,
                                            int &ind_f) {
    const uint4 a = L.descA[cls * P.Cpad + c];
    const IdxT *ip = (const IdxT *)P.idx + (size_t)s * P.row_stride + c;
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
__device__ __forceinline__ void accum_slot(const GenHot &P, const bool acc_by_slot, const Lds &L, int cls, int c, int lane,
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
            double *cell = acc_by_slot ? L.acc + (size_t)cls * P.Cpad + c : L.acc + (size_t)(feat + k) * 64 + lane;
            *cell = fma(fs, d, *cell);
        }
    }
}

// Ewald delta of one flip (ewald.pyx:38-58), wave-parallel over sites; returns the
// lane-partial (caller reduces).  Reads ROWS of the transposed matrix, i.e. the same
// entries M[i, add] / M[j, sub] the reference reads as columns.
// The sum is a streaming gather -- per site one LDS byte (species), one index-table entry (L2) and one
// entry of each of the two rows (HBM: the matrix of BASELINE config 3 is 382 MB) -- i.e. a chain of
// three dependent loads per site.  Site by site that chain is all a wave does (measured, round 4:
// 2650 cycles per site and lane, 1.9 TB/s of HBM fetch at two waves per SIMD); here U = 8 sites per
// lane are in flight together: their species, then their eight index entries, then their sixteen
// row entries are issued back to back, one exposed latency per stage and batch.
// Every site k != s keeps its index (i == j) and that index is neither `add` nor `sub` (those belong
// to site s), so its term is 2 (M[i, add] - M[i, sub]); site s itself contributes the diagonal
// entries M[add, add] - M[sub, sub] (ewald.pyx:46-57 with i == add, j == sub).
template <bool PATCH, typename PT>
__device__ __forceinline__ double ewald_partial(const PT &P, const Lds &L, int lane, int s,
                                                int oldc, int newc, int ps, int pc) {
    constexpr int U = 8;
    const int W = P.ew_W, N = P.N;
    const int *inds = P.ew_inds;
    const int add = inds[(size_t)s * W + newc];
    const int sub = inds[(size_t)s * W + oldc];
    const double *radd = P.ew_Mt + (size_t)(add < 0 ? 0 : add) * P.ew_M;
    const double *rsub = P.ew_Mt + (size_t)(sub < 0 ? 0 : sub) * P.ew_M;
    const double fa = add < 0 ? 0.0 : 2.0, fs = sub < 0 ? 0.0 : 2.0;
    double out = 0;
    for (int k0 = lane; k0 < N; k0 += 64 * U) {
        int kk[U], ii[U];
        double va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            kk[u] = min(k0 + 64 * u, N - 1); // (clamped: loads only; masked below)
            int v = L.occ[kk[u]];
            if (PATCH) v = (kk[u] == ps) ? pc : v;
            ii[u] = (int)((uint32_t)kk[u] * (uint32_t)W + (uint32_t)v);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) ii[u] = inds[ii[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = ii[u] < 0 ? 0 : ii[u];
            va[u] = radd[i];
            vb[u] = rsub[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = k0 + 64 * u < N && kk[u] != s && ii[u] >= 0;
            out += live ? fa * va[u] - fs * vb[u] : 0.0;
        }
    }
    if (lane == 0) out += (add < 0 ? 0.0 : radd[add]) - (sub < 0 ? 0.0 : rsub[sub]);
    return out;
}

// Compact form of the same delta when M[a][b] = q_a q_b G[site_a][site_b] (a != b):
//   sum_k [2 M[i_k, add] - 2 M[i_k, sub]]  (k != s, i_k = j_k)  + M[add,add] - M[sub,sub]
//     = 2 (q_add - q_sub) * sum_{k != s} q(k, occ_k) G[s][k] + diag(add) - diag(sub)
// One fully-used row of G (N x 8 B) streams per flip instead of two strided matrix rows.
template <bool PATCH, typename PT>
__device__ __forceinline__ double ewald_compact_partial(const PT &P, const Lds &L, int lane, int s,
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

// potential-field update in HBM after an accepted flip of site s by charge dq (general
// kernel): phi[j] += dq * G[s][j] for every other changeable site j; lane-strided, so each
// address is always touched by the same lane (program order keeps later updates coherent; the
// self-term patch after the sweep is one store of the same value from every lane).
template <typename PT>
__device__ __forceinline__ void field_apply_global(const PT &P, double *phi, int lane, int s, double dq) {
    const double *g = P.ew_G + (size_t)s * P.ew_nact;
    const int js = s - P.ew_act_base;
    const double keep = phi[js]; // phi excludes the self term: put back after the sweep
    field_sweep<false, 6>(phi, g, g, lane, P.ew_nact, dq, 0.0);
    phi[js] = keep;
}

// both flips of a swap in one pass over phi
template <typename PT>
__device__ __forceinline__ void field_apply_global2(const PT &P, double *phi, int lane, int s1, double dq1,
                                                    int s2, double dq2) {
    const double *g1 = P.ew_G + (size_t)s1 * P.ew_nact, *g2 = P.ew_G + (size_t)s2 * P.ew_nact;
    const int j1 = s1 - P.ew_act_base, j2 = s2 - P.ew_act_base;
    const double keep1 = phi[j1], keep2 = phi[j2];
    const double c12 = g2[j1], c21 = g1[j2]; // cross terms: entry j1 sees flip 2 only, j2 flip 1 only
    field_sweep<true, 4>(phi, g1, g2, lane, P.ew_nact, dq1, dq2);
    if (j1 != j2) {
        phi[j1] = fma(dq2, c12, keep1);
        phi[j2] = fma(dq1, c21, keep2);
    } else {
        phi[j1] = keep1;
    }
}

// Metropolis feature deltas accumulated since launch start, reduced over the wave:
// calls emit(f, value) on lane 0 for every cluster-expansion feature f.
template <typename PT, typename F>
__device__ __forceinline__ void reduce_feature_acc(const PT &P, const Lds &L, int lane, F emit) {
    if (P.acc_by_slot) {
        double *out = L.acc + (size_t)P.nclasses * P.Cpad; // [Fce] scratch
        for (int f = lane; f < P.Fce; f += 64) out[f] = 0.0;
        for (int cls = 0; cls < P.nclasses; ++cls)
            for (int c = lane; c < P.Cpad; c += 64) {
                const uint4 b = L.descB[cls * P.Cpad + c];
                const double v = L.acc[(size_t)cls * P.Cpad + c];
                if (v != 0.0)
                    __hip_atomic_fetch_add(&out[b.z & 0xffffu], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        for (int f = 0; f < P.Fce; ++f) {
            const double sm = out[f];
            if (lane == 0) emit(f, sm);
        }
    } else {
        for (int f = 0; f < P.Fce; ++f) {
            const double sm = wave_sum(L.acc[(size_t)f * 64 + lane]);
            if (lane == 0) emit(f, sm);
        }
    }
}

template <typename IdxT, int NSLOT, int MM, bool GENERIC, bool WL>
__global__ void __launch_bounds__(256, (NSLOT <= 4 ? 4 : 2)) mc_kernel(const KParams P, const int replay) {
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
    L.acc = nullptr; L.wl_S = nullptr; L.wl_H = nullptr; L.wl_O = nullptr; L.wl_cf = nullptr;
    if (WL) {
        L.wl_S = (double *)wp;        wp += (size_t)P.L * 8;
        L.wl_H = (long long *)wp;     wp += (size_t)P.L * 8;
        L.wl_O = (long long *)wp;     wp += (size_t)P.L * 8;
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
                L.wl_O[i] = P.wl_occur[(size_t)r * P.L + i];
            }
            for (int i = lane; i < P.F; i += 64) L.wl_cf[i] = P.features[(size_t)r * P.F + i];
        } else {
            const int ncell = P.acc_by_slot ? P.nclasses * P.Cpad + P.Fce : P.Fce * 64;
            for (int i = lane; i < ncell; i += 64) L.acc[i] = 0.0;
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
    double bias = P.bias_type ? P.bias[r] : 0.0;     // trace.bias (kernel/base.py:362-363)
    double charge = P.bias_type == SMOLMC_BIAS_SQUARE_CHARGE ? P.charge[(size_t)r * SMOLMC_MAX_BIAS_ROWS] : 0.0;
    int last_acc = 1;
    double wl_m = 0.0;
    long long wl_counter = 0;
    if (WL) {
        wl_m = P.wl_m[r];
        wl_counter = P.wl_counter[r];
    }
    uint32_t W0 = 0, W1 = 0, W2 = 0, W3 = 0; // RNG batch: lane l = block (l&3) of step base+(l>>2)
    uint32_t w_site_carry = 0;
    double logu_b = 0.0; // native mode: log(u) of the lane's step, computed per batch
    unsigned long long batch_base = ~0ull - 64ull;
    const long long smp_every = P.smp.every;
    long long smp_countdown = smp_every, smp_index = 0;
    // the few kernel arguments every step needs (everything else: PK, see gen_params)
    GenHot hot;
    hot.idx = P.idx;
    hot.Cpad = P.Cpad;
    hot.row_stride = (size_t)P.Mmax * P.Cpad;
    const long long nsteps_run = P.steps_to_run;
    const int nsub = P.nsub, step_type = P.step_type, nclasses = P.nclasses;
    // (feature switches packed into one register)
    const uint32_t gflags = (P.has_ewald ? 1u : 0u) | (P.has_mu ? 2u : 0u) | (P.acc_by_slot ? 4u : 0u) | ((uint32_t)P.bias_type << 3);
#define has_ewald ((gflags & 1u) != 0u)
#define has_mu ((gflags & 2u) != 0u)
#define acc_by_slot ((gflags & 4u) != 0u)
#define bias_type ((int)(gflags >> 3))
    const int p0_0 = P.sub_ptr[0], sbase_0 = P.sub_base[0];
    const uint32_t nact_0 = (uint32_t)(P.sub_ptr[1] - p0_0);

    for (long long it_step = 0; it_step < nsteps_run; ++it_step, ++step) {
        // ================= proposal =========================================
        int nfl = 0, s1 = 0, n1 = 0, o1 = 0, s2 = 0, n2 = 0, o2 = 0;
        double u = 0.0, lu = 0.0;
        bool have_lu = false; // native mode supplies log(u) from the batch; replay takes it per step
        if (replay) {
            const int *st = PK.rp_steps + ((size_t)r * nsteps_run + it_step) * 4;
            int a0 = st[0], a1 = st[1], a2 = st[2], a3 = st[3];
            u = PK.rp_u[(size_t)r * nsteps_run + it_step];
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
                // log of the acceptance uniform of all 16 steps at once (block-0 lanes)
                logu_b = log(philox_u53(o.w[2], o.w[3]));
            }
            const int l4 = (int)(step & 15ull) * 4;
            const uint32_t w_sub = rdlane(W0, l4);
            const uint32_t w_site = l4 == 0 ? w_site_carry : rdlane(W1, l4 - 4);
            u = philox_u53(rdlane(W2, l4), rdlane(W3, l4));
            lu = __hiloint2double((int)rdlane((uint32_t)__double2hiint(logu_b), l4),
                                  (int)rdlane((uint32_t)__double2loint(logu_b), l4));
            have_lu = true;
            // sublattice: MCUsher.get_random_sublattice (mcusher.py:146-148)
            int sl = 0;
            if (nsub > 1) {
                const double x = (double)w_sub * (1.0 / 4294967296.0);
                sl = nsub - 1;
                for (int q = nsub - 2; q >= 0; --q)
                    if (x < PK.sub_cum[q]) sl = q;
            }
            int p0 = p0_0, sbase = sbase_0;
            uint32_t nact = nact_0;
            if (nsub > 1) {
                p0 = PK.sub_ptr[sl];
                nact = (uint32_t)(PK.sub_ptr[sl + 1] - p0);
                sbase = PK.sub_base[sl];
            }
            const uint32_t k1 = __umulhi(w_site, nact);
            s1 = sbase >= 0 ? sbase + (int)k1 : PK.sub_sites[p0 + k1];
            s1 = uni(s1);
            o1 = uni((int)L.occ[s1]);
            if (step_type == SMOLMC_STEP_FLIP) {
                // Flip.propose_step (mcusher.py:154-170)
                const int c0 = PK.sub_code_ptr[sl];
                const uint32_t nc = (uint32_t)(PK.sub_code_ptr[sl + 1] - c0);
                const uint32_t kk = __umulhi(rdlane(W0, l4 + 1), nc - 1);
                int code = -1;
                uint32_t seen = 0;
                for (uint32_t c = 0; c < nc; ++c) {
                    int cc = PK.sub_codes[c0 + c];
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
                            const int cs = sbase >= 0 ? sbase + (int)kc : PK.sub_sites[p0 + kc];
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
                            const int cs = sbase >= 0 ? sbase + (int)kc : PK.sub_sites[p0 + kc];
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
                                const int cs = sbase >= 0 ? sbase + (int)a : PK.sub_sites[p0 + a];
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
            cls1 = nclasses > 1 ? uni((int)L.site_class[s1]) : 0;
            nit1 = cls1 == 255 ? 0 : uni(L.cls_niter[cls1]);
        }
        // The two-group kernels evaluate both groups WITHOUT a condition per group: groups a class does
        // not use have all-zero descriptors and index rows that point at the site itself (they add
        // exactly 0.0), a site without clusters (class 255) borrows class 0 with weight zero.  With a
        // branch per group the compiler drains its wait counters at every group and the loads of a
        // step -- index entries, occupancy bytes, tensor entries: three dependent round trips per
        // group -- run one group after the other.
        constexpr bool ALL = NSLOT <= 2;
        if (nfl == 2) {
            cls2 = nclasses > 1 ? uni((int)L.site_class[s2]) : 0;
            nit2 = cls2 == 255 ? 0 : uni(L.cls_niter[cls2]);
            const int c1e = cls1 == 255 ? 0 : cls1, c2e = cls2 == 255 ? 0 : cls2;
            const double k1 = cls1 == 255 ? 0.0 : 1.0, k2 = cls2 == 255 ? 0.0 : 1.0;
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) {
                const int c = lane + 64 * it;
                if (ALL || it < nit1)
                    e += k1 * eval_slot<IdxT, MM, GENERIC, false>(hot, L, c1e, c, s1, o1, n1, 0, 0, ii1[it],
                                                                  jf1[it]);
                if (ALL || it < nit2)
                    e += k2 * eval_slot<IdxT, MM, GENERIC, true>(hot, L, c2e, c, s2, o2, n2, s1, n1, ii2[it],
                                                                 jf2[it]);
            }
        } else if (nfl == 1) {
            const int c1e = cls1 == 255 ? 0 : cls1;
            const double k1 = cls1 == 255 ? 0.0 : 1.0;
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) {
                const int c = lane + 64 * it;
                if (ALL || it < nit1)
                    e += k1 * eval_slot<IdxT, MM, GENERIC, false>(hot, L, c1e, c, s1, o1, n1, 0, 0, ii1[it],
                                                                  jf1[it]);
            }
        }
        double dEw = 0.0, dMu = 0.0;
        double fdq1 = 0.0, fdq2 = 0.0; // potential-field mode: charge changes of the flips
        if (has_ewald && nfl >= 1 && PK.ew_field) {
            // O(1) proposal from the walker's potential field (HBM copy, read past the L1 so
            // that the row updates of earlier accepted steps are seen)
            double *phi = PK.ew_phi + (size_t)r * PK.ew_nact;
            const int W = PK.ew_W, ab = PK.ew_act_base;
            fdq1 = PK.ew_qs[(size_t)s1 * W + n1] - PK.ew_qs[(size_t)s1 * W + o1];
            const double p1 = __hip_atomic_load(&phi[s1 - ab], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dEw = 2.0 * fdq1 * p1 + (PK.ew_dg[(size_t)s1 * W + n1] - PK.ew_dg[(size_t)s1 * W + o1]);
            if (nfl == 2) {
                fdq2 = PK.ew_qs[(size_t)s2 * W + n2] - PK.ew_qs[(size_t)s2 * W + o2];
                const double p2 = __hip_atomic_load(&phi[s2 - ab], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double cross = s2 != s1 ? PK.ew_G[(size_t)s2 * PK.ew_nact + (s1 - ab)] : 0.0;
                dEw += 2.0 * fdq2 * (p2 + fdq1 * cross) +
                       (PK.ew_dg[(size_t)s2 * W + n2] - PK.ew_dg[(size_t)s2 * W + o2]);
            }
            dEw = uni_d(dEw);
        } else if (has_ewald && nfl >= 1) {
            if (PK.ew_compact) {
                const int W = PK.ew_W;
                const double s1sum = PK.ew_frozen[s1] + wave_sum(ewald_compact_partial<false>(PK, L, lane, s1, 0, 0));
                dEw = 2.0 * (PK.ew_qs[(size_t)s1 * W + n1] - PK.ew_qs[(size_t)s1 * W + o1]) * s1sum +
                      (PK.ew_dg[(size_t)s1 * W + n1] - PK.ew_dg[(size_t)s1 * W + o1]);
                if (nfl == 2) {
                    const double s2sum = PK.ew_frozen[s2] + wave_sum(ewald_compact_partial<true>(PK, L, lane, s2, s1, n1));
                    dEw += 2.0 * (PK.ew_qs[(size_t)s2 * W + n2] - PK.ew_qs[(size_t)s2 * W + o2]) * s2sum +
                           (PK.ew_dg[(size_t)s2 * W + n2] - PK.ew_dg[(size_t)s2 * W + o2]);
                }
                dEw = uni_d(dEw);
            } else {
                double pe = ewald_partial<false>(PK, L, lane, s1, o1, n1, 0, 0);
                if (nfl == 2) pe += ewald_partial<true>(PK, L, lane, s2, o2, n2, s1, n1);
                dEw = wave_sum(pe);
            }
        }
        if (has_mu && nfl >= 1) {
            // delta chemical work against the ORIGINAL occupancy (ensemble.py:368-374)
            dMu = PK.mu[(size_t)s1 * PK.mu_W + n1] - PK.mu[(size_t)s1 * PK.mu_W + o1];
            if (nfl == 2) {
                const int orig2 = uni((int)L.occ[s2]);
                dMu += PK.mu[(size_t)s2 * PK.mu_W + n2] - PK.mu[(size_t)s2 * PK.mu_W + orig2];
            }
            dMu = uni_d(dMu);
        }
        double dH = wave_sum(e);
        if (has_ewald) dH += PK.ew_coef * dEw;
        if (has_mu) dH -= dMu;
        // MCBias.compute_bias_change against the ORIGINAL occupancy (kernel/base.py:307-311):
        // FugacityBias log-ratio per flipped site (bias.py:188-206); SquareChargeBias the
        // difference of -penalty * charge^2 (bias.py:75-93, :264-277) on the running charge
        double dB = 0.0, dQ = 0.0;
        int bias_orig2 = 0; // species of site 2 BEFORE the step (the second flip sees the first)
        if (!WL && bias_type && nfl >= 1) {
            const double *b1 = PK.bias_tab + (size_t)s1 * PK.bias_W;
            const int orig2 = nfl == 2 ? uni((int)L.occ[s2]) : 0;
            bias_orig2 = orig2;
            const double *b2 = PK.bias_tab + (size_t)s2 * PK.bias_W;
            if (bias_type == SMOLMC_BIAS_FUGACITY) {
                dB = log(b1[n1] / b1[o1]);
                if (nfl == 2) dB += log(b2[n2] / b2[orig2]);
            } else if (bias_type == SMOLMC_BIAS_SQUARE_CHARGE) {
                dQ = b1[n1] - b1[o1];
                if (nfl == 2) dQ += b2[n2] - b2[orig2];
                const double cn = charge + dQ;
                dB = -PK.bias_pen * (cn * cn) - (-PK.bias_pen * (charge * charge));
            } else {
                // SquareHyperplaneBias (bias.py:290-366; compute_bias_change is the inherited
                // recompute-and-subtract, :75-93, restated on the running A_r . n - b_r kept per
                // walker in HBM: a rarely used term, not worth registers in this kernel)
                double sq_new = 0.0, sq_old = 0.0;
                for (int k = 0; k < PK.bias_rows; ++k) {
                    const size_t ro = (size_t)k * PK.bias_row_stride;
                    double dq = b1[ro + n1] - b1[ro + o1];
                    if (nfl == 2) dq += b2[ro + n2] - b2[ro + orig2];
                    const double c = ((volatile const double *)PK.charge)[(size_t)r * SMOLMC_MAX_BIAS_ROWS + k]; // (vector load: written by this wave)
                    sq_old += c * c;
                    sq_new += (c + dq) * (c + dq);
                }
                dB = -PK.bias_pen * sq_new - (-PK.bias_pen * sq_old);
            }
            dB = uni_d(dB);
            dQ = uni_d(dQ);
        }

        // ================= accept ==============================================
        bool accepted;
        if (!WL) {
            // MetropolisAcceptMixin._accept_step (metropolis.py:31-49)
            const double exponent = -beta * dH + 0.0 + dB;
            accepted = __ballot(exponent >= 0.0 ? true : (exponent > (have_lu ? lu : log(u)))) != 0ull;
        } else {
            // WangLandau._accept_step (wanglandau.py:186-202)
            const double new_h = H + dH;
            if (new_h < PK.wl_min || new_h >= PK.wl_max) {
                accepted = false;
            } else {
                const int b = (int)floordiv_exact(H - PK.wl_min, PK.wl_bin);
                const int nb = (int)floordiv_exact(new_h - PK.wl_min, PK.wl_bin);
                const double exponent = L.wl_S[b] - L.wl_S[nb] + 0.0;
                accepted = __ballot(exponent >= 0.0 ? true : (exponent > (have_lu ? lu : log(u)))) != 0ull;
            }
        }

        // ================= update ==============================================
        if (accepted) {
            // MCKernel._do_accept_step (kernel/base.py:327-343) + trace += delta
            // (sampler/sampler.py:204-207)
            if (nfl >= 1) {
#pragma unroll
                for (int it = 0; it < NSLOT; ++it)
                    if (it < nit1) accum_slot<WL>(hot, acc_by_slot != 0, L, cls1, lane + 64 * it, lane, ii1[it], jf1[it]);
            }
            if (nfl == 2) {
#pragma unroll
                for (int it = 0; it < NSLOT; ++it)
                    if (it < nit2) accum_slot<WL>(hot, acc_by_slot != 0, L, cls2, lane + 64 * it, lane, ii2[it], jf2[it]);
            }
            if (nfl >= 1) L.occ[s1] = (uint8_t)n1; // every lane stores the same byte
            if (nfl == 2) L.occ[s2] = (uint8_t)n2;
            if (lane == 0) {
                if (WL) {
                    if (has_ewald) L.wl_cf[PK.Fce] += dEw;
                    if (has_mu) L.wl_cf[PK.Fce + has_ewald] += dMu;
                }
            }
            if (has_ewald && PK.ew_field) {
                double *phi = PK.ew_phi + (size_t)r * PK.ew_nact;
                // two flips: one pass over phi (the update is bound by HBM / Infinity-Cache traffic);
                // a replayed step may flip the same site twice -> two passes keep the self-exclusion
                if (nfl == 2 && s2 != s1) {
                    if (fdq1 != 0.0 || fdq2 != 0.0) field_apply_global2(PK, phi, lane, s1, fdq1, s2, fdq2);
                } else {
                    if (fdq1 != 0.0) field_apply_global(PK, phi, lane, s1, fdq1);
                    if (nfl == 2 && fdq2 != 0.0) field_apply_global(PK, phi, lane, s2, fdq2);
                }
            }
            acc_ew += dEw;
            acc_mu += dMu;
            H += dH;
            bias += dB;
            charge += dQ;
            if (bias_type == SMOLMC_BIAS_SQUARE_HYPERPLANE) {
                const double *b1 = PK.bias_tab + (size_t)s1 * PK.bias_W, *b2 = PK.bias_tab + (size_t)s2 * PK.bias_W;
                for (int k = 0; k < PK.bias_rows; ++k) {
                    const size_t ro = (size_t)k * PK.bias_row_stride;
                    double dq = b1[ro + n1] - b1[ro + o1];
                    if (nfl == 2) dq += b2[ro + n2] - b2[ro + bias_orig2];
                    if (lane == 0) {
                        volatile double *cq = (volatile double *)PK.charge + (size_t)r * SMOLMC_MAX_BIAS_ROWS + k;
                        *cq = *cq + dq;
                    }
                }
            }
            nacc++;
        }
        last_acc = accepted ? 1 : 0;

        if (WL) {
            // WangLandau._do_post_step (wanglandau.py:222-266)
            const double bq = floordiv_exact(H - PK.wl_min, PK.wl_bin);
            if (bq >= 0.0 && bq < (double)PK.L) {
                const int b = (int)bq;
                wl_counter++;
                const size_t cell = (size_t)r * PK.L + b;
                // occurrences live in LDS beside entropy and histogram (a global counter cost a dependent
                // load on every step); the per-bin feature statistics are running SUMS when update_period
                // is 1 (wl_sum_mode: mean = sum / occurrences, converted when read) -- one atomic without
                // return value per feature -- else the reference's running mean (wanglandau.py:235-239)
                const long long total = L.wl_O[b];
                if (lane < PK.F) {
                    double *mf = PK.wl_meanf + cell * PK.F + lane;
                    if (PK.wl_sum_mode) {
                        unsafeAtomicAdd(mf, L.wl_cf[lane]);
                    } else {
                        const double inv = 1.0 / (double)(total + 1);
                        *mf = inv * (L.wl_cf[lane] + (double)total * (*mf));
                    }
                }
                if (wl_counter % PK.wl_update == 0) {
                    if (lane == 0) {
                        L.wl_S[b] += wl_m;
                        L.wl_H[b] += 1;
                        L.wl_O[b] = total + 1;
                    }
                }
            }
            if (PK.wl_check != 0 && wl_counter % PK.wl_check == 0) { // (check period 0: no device-side check)
                long cnt = 0;
                double sum = 0;
                for (int i = lane; i < PK.L; i += 64)
                    if (L.wl_S[i] > 0) { cnt++; sum += (double)L.wl_H[i]; }
                const double tcnt = wave_sum((double)cnt), tsum = wave_sum(sum);
                if (tcnt >= 2.0) {
                    const double thr = PK.wl_flat * (tsum / tcnt);
                    int bad = 0;
                    for (int i = lane; i < PK.L; i += 64)
                        if (L.wl_S[i] > 0 && !((double)L.wl_H[i] > thr)) bad = 1;
                    if (__ballot(bad) == 0ull) {
                        for (int i = lane; i < PK.L; i += 64) L.wl_H[i] = 0;
                        wl_m = wl_m / PK.wl_div;
                    }
                }
            }
        }
        if (replay && lane == 0) {
            if (PK.rp_acc) PK.rp_acc[(size_t)r * nsteps_run + it_step] = (uint8_t)last_acc;
            if (PK.rp_H) PK.rp_H[(size_t)r * nsteps_run + it_step] = H;
        }
        if (smp_every && --smp_countdown == 0) { // record one thinned sample of this walker
            smp_countdown = smp_every;
            const size_t row = (size_t)smp_index * PK.R + r;
            smp_index++;
            double *dstf = PK.smp.feat + row * PK.F;
            const double *base = PK.features + (size_t)r * PK.F;
            if (WL) {
                for (int i = lane; i < PK.F; i += 64) dstf[i] = L.wl_cf[i];
            } else {
                reduce_feature_acc(PK, L, lane, [&](int f, double sm) { dstf[f] = base[f] + sm; });
                if (lane == 0) {
                    if (has_ewald) dstf[PK.Fce] = base[PK.Fce] + acc_ew;
                    if (has_mu) dstf[PK.Fce + has_ewald] = base[PK.Fce + has_ewald] + acc_mu;
                }
            }
            if (lane == 0) {
                PK.smp.H[row] = H;
                PK.smp.acc[row] = (uint8_t)last_acc;
            }
            if (PK.smp.occ) {
                uint4 *dst = (uint4 *)(PK.smp.occ + row * PK.Npad);
                const uint4 *src = (const uint4 *)L.occ;
                for (int i = lane; i < PK.Npad / 16; i += 64) dst[i] = src[i];
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
            P.wl_occur[(size_t)r * P.L + i] = L.wl_O[i];
        }
        for (int i = lane; i < P.F; i += 64) feat[i] = L.wl_cf[i];
        if (lane == 0) {
            P.wl_m[r] = wl_m;
            P.wl_counter[r] = wl_counter;
        }
    } else {
        reduce_feature_acc(P, L, lane, [&](int f, double sm) { feat[f] += sm; });
        if (lane == 0) {
            if (has_ewald) feat[P.Fce] += acc_ew;
            if (has_mu) feat[P.Fce + has_ewald] += acc_mu;
        }
    }
    if (lane == 0) {
        P.enthalpy[r] = H;
        P.nsteps[r] = step;
        P.nacc[r] = nacc;
        P.last_acc[r] = (uint8_t)last_acc;
        if (bias_type) P.bias[r] = bias;
        if (bias_type == SMOLMC_BIAS_SQUARE_CHARGE) P.charge[(size_t)r * SMOLMC_MAX_BIAS_ROWS] = charge;
    }
}


#undef has_ewald
#undef has_mu
#undef acc_by_slot
#undef bias_type

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


// all (index type, GENERIC, WL) combinations of one NSLOT
template <int NSLOT> static int launch_general_nslot(smolmc_handle *h, const KParams &kp, int replay) {
    const bool wl = h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU;
    if (h->generic)
        return wl ? launch_mc_mm<int32_t, NSLOT, true, true>(h, kp, replay)
                  : launch_mc_mm<int32_t, NSLOT, true, false>(h, kp, replay);
    if (h->idx16)
        return wl ? launch_mc_mm<uint16_t, NSLOT, false, true>(h, kp, replay)
                  : launch_mc_mm<uint16_t, NSLOT, false, false>(h, kp, replay);
    return wl ? launch_mc_mm<int32_t, NSLOT, false, true>(h, kp, replay)
              : launch_mc_mm<int32_t, NSLOT, false, false>(h, kp, replay);
}
