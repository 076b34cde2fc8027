// mc_univ.h -- the universal Monte-Carlo kernel (mc_univ_kernel): the backstop that takes every
// model, usher and kernel combination the reference composes (kernel/base.py:192-239,
// kernel/__init__.py:35-56) and that the specialised kernels (mc_lean*.h, mc_wl.h, mc_general.h)
// decline:
//   * steps of up to SMOLMC_MAX_STEP_FLIPS sequential flips (TableFlip, mcusher.py:397-711, native
//     stream and replay) with Metropolis OR Wang-Landau, with any MCBias, correlation or interaction
//     features, any number of correlation functions per orbit, aliased supercells, any sublattice
//     layout / encoding, site-dependent chemical potentials, dense / compact / field Ewald;
//   * any number of clusters per site and of site classes: it walks the reference's own per-site
//     tables (LocalEvalData, processor/expansion.py:24-36,120-156) instead of slot descriptors;
//   * occupancies that do not fit a wave's share of LDS: one byte per site in HBM (L2-resident),
//     read past the L1 so that the wave sees its own tentative flips.
// One 64-lane wavefront owns one walker, as everywhere in this engine; lanes split the cluster rows
// of a local record.  A step is evaluated in two passes: (1) the enthalpy change, flips applied
// tentatively one after the other (sequential-flip semantics, expansion.py:217-229) -- one wave
// reduction per step; (2) on acceptance only, the feature deltas per (record, function), reduced
// per feature like the reference's p / ratio / J (evaluator.pyx:244-262, :302-315).  Rejected steps
// undo the tentative writes.  The native random stream and the usher logic restate
// oracle/smolmc_oracle.c (propose_step, propose_swap_in, propose_table_flip) word for word:
// identical Philox words -> identical trajectories.
#pragma once
#include "mc_general.h"

// The parameter block is read from the kernel-argument segment where it is used (gen_params / rare_params): held in
// SGPRs across the step loop it cost this kernel 350-400 SGPR spills (v_readlane reloads on every step).  The empty
// asm makes the pointer opaque: the scalar loads of a region stay in that region.
typedef const UParams __attribute__((address_space(4))) *UParamsKernarg;
__device__ __forceinline__ UParamsKernarg univ_params() {
    UParamsKernarg p = (UParamsKernarg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// occupancy access: LDS bytes, or HBM bytes past the L1 (agent-scope relaxed atomics compile to
// global_load/store_ubyte with sc1: the wave reads what it has just written)
template <bool OL> __device__ __forceinline__ int uocc_ld(const uint8_t *occ, int s) {
    if (OL) return (int)occ[s];
    return (int)__hip_atomic_load(occ + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool OL> __device__ __forceinline__ void uocc_st(uint8_t *occ, int s, int v, int lane) {
    if (OL) {
        occ[s] = (uint8_t)v; // (every lane, same byte)
    } else {
        if (lane == 0) __hip_atomic_store(occ + s, (uint8_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    }
}

// ... without the fence (the caller fences once after the last store)
template <bool OL> __device__ __forceinline__ void uocc_st_nf(uint8_t *occ, int s, int v, int lane) {
    if (OL) occ[s] = (uint8_t)v;
    else if (lane == 0) __hip_atomic_store(occ + s, (uint8_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ philox_out univ_block(unsigned long long step, uint32_t block, uint32_t k0, uint32_t k1) {
    return philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), block, 0u, k0, k1);
}

// lane-partial enthalpy change of a WHOLE step (up to SMOLMC_MAX_STEP_FLIPS sequential flips) against
// the occupancy before the step: sum over the cluster rows of every flipped site and their
// correlation functions of natural[feature] * (t[ind_f] - t[ind_i]) * size / ratio / J.
// The rows of ALL flips of the step are dealt to the lanes in one sweep, two groups of 64 in flight.
// The sequential-flip semantics (expansion.py:217-229: flip f sees the flips before it) come from a
// lane-local patch instead of tentative writes: a member site that an EARLIER flip of the step
// changed reads that flip's species.  A row costs a chain of three dependent round trips
// (packed row -> {member species, record} -> tensor entries); round 4 walked five (row -> record
// descriptor -> member sites -> species -> tensor entries), flip after flip, with a write + fence
// between the flips, and that chain was all a wave did (config 2: 17 us per step and walker).
// The flips arrive lane-indexed: lane j < nfl holds site / new species / first row / rows before
// flip j (vS, vC, vQ0, vCum); `total` rows.
// dF (or null): LDS cells of the step's feature deltas; every row adds scale * (t_k[ind_f] - t_k[ind_i])
// to the cell of its feature (committed to the walker's features when the step is accepted).
// MAXF: the most flips a step of this call has (2: flips and swaps without the loops over eight flips)
template <bool OL, bool K1, bool DL, int MAXF>
__device__ __forceinline__ double univ_step_partial(const unsigned char *sh, const uint8_t *occ, const int lane, const int nfl,
                                                    const int vS, const int vC, const uint32_t vQ0, const uint32_t vCum,
                                                    const uint32_t total, double *dF, const int cshift) {
    const UParamsKernarg Q = univ_params();
    // DL: the record dictionary, the tensors and the natural parameters sit in LDS (sh: the workgroup's copy)
    const double *tens = DL ? (const double *)(sh + SMOLMC_UNIV_DICT_RECS * sizeof(URecE))
                            : (Q->T.corr_mode ? Q->T.corr_tensors : Q->T.interaction_tensors);
    const int Imax = Q->max_I;
    const uint4 *rows = Q->rows;
#ifdef SMOLMC_UNIV_ROWS16 // experiment build (make EXTRA=-DSMOLMC_UNIV_ROWS16; see URow16): one 16-byte record per row instead of two
    const bool rows16 = uni(Q->rows16) != 0;
#else
    constexpr bool rows16 = false; // (even a never-taken uniform branch here cost the default path 3-4 %)
#endif
    const URecE *recs_e = DL ? (const URecE *)sh : Q->recs_e;
    const double *natural = DL ? (const double *)(sh + SMOLMC_UNIV_DICT_RECS * sizeof(URecE) + SMOLMC_UNIV_DICT_TENS * 8) : Q->natural;
    // (LDS atomics serialise per address for the whole CU: the cells of a feature exist 1 << cshift times, the
    // lane picks its copy; the copies are added up when the step is accepted)
    const int ccopy = lane & ((1 << cshift) - 1);
    const int S0 = (int)rdlane((uint32_t)vS, 0), C0 = (int)rdlane((uint32_t)vC, 0);
    const uint32_t Q00 = rdlane(vQ0, 0);
    double e = 0.0;
    constexpr int G = 2;
    for (uint32_t pb = 0; pb < total; pb += 64 * G) {
        bool live[G];
        int fi[G], sm[G], cm[G];
        uint4 ra[G], rb[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t p = pb + 64u * g + (uint32_t)lane;
            live[g] = p < total;
            const uint32_t pp = live[g] ? p : total - 1u; // (clamped: loads only)
            fi[g] = 0; sm[g] = S0; cm[g] = C0;
            uint32_t qo = Q00;
#pragma unroll
            for (int j = 1; j < MAXF; ++j)
                if (j < nfl) {
                    const uint32_t cj = rdlane(vCum, j);
                    const bool in = pp >= cj;
                    fi[g] = in ? j : fi[g];
                    sm[g] = in ? (int)rdlane((uint32_t)vS, j) : sm[g];
                    cm[g] = in ? (int)rdlane((uint32_t)vC, j) : cm[g];
                    qo = in ? rdlane(vQ0, j) - cj : qo;
                }
            if (rows16) {
                const uint4 w = rows[(size_t)(qo + pp)];
                ra[g] = make_uint4(w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16);
                rb[g] = make_uint4(w.z & 0xffffu, w.z >> 16, w.w, 0u);
            } else {
                const uint4 *rp = rows + 2u * (size_t)(qo + pp);
                ra[g] = rp[0];
                rb[g] = rp[1];
            }
        }
        URecE R[G];
#pragma unroll
        for (int g = 0; g < G; ++g) R[g] = recs_e[rb[g].z];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int x[SMOLMC_MAX_CLUSTER_SITES] = {(int)ra[g].x, (int)ra[g].y, (int)ra[g].z, (int)ra[g].w, (int)rb[g].x, (int)rb[g].y};
            int v[SMOLMC_MAX_CLUSTER_SITES];
#pragma unroll
            for (int i = 0; i < SMOLMC_MAX_CLUSTER_SITES; ++i)
                if (i < Imax) v[i] = uocc_ld<OL>(occ, x[i]);
#pragma unroll
            for (int j = 0; j + 1 < MAXF; ++j)
                if (j + 1 < nfl) { // flips before this row's own
                    const int sj = (int)rdlane((uint32_t)vS, j), cj = (int)rdlane((uint32_t)vC, j);
                    const bool before = j < fi[g];
#pragma unroll
                    for (int i = 0; i < SMOLMC_MAX_CLUSTER_SITES; ++i)
                        if (i < Imax) v[i] = (before && x[i] == sj) ? cj : v[i];
                }
            uint32_t ind_i = 0, ind_f = 0;
#pragma unroll
            for (int i = 0; i < SMOLMC_MAX_CLUSTER_SITES; ++i)
                if (i < Imax) { // (members beyond the cluster repeat the first one with stride 0)
                    const int vf = (x[i] == sm[g]) ? cm[g] : v[i];
                    ind_i += (uint32_t)R[g].st[i] * (uint32_t)v[i];
                    ind_f += (uint32_t)R[g].st[i] * (uint32_t)vf;
                }
            const double *t0 = tens + R[g].t_off;
            const double *nat = natural + R[g].feat;
            double p;
            if (K1) {
                const double d = t0[ind_f] - t0[ind_i];
                p = R[g].nat0 * d;
                if (dF != nullptr && live[g] && d != 0.0)
                    __hip_atomic_fetch_add(&dF[(R[g].feat << cshift) + ccopy], R[g].scale * d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            } else {
                p = 0.0;
                for (int k = 0; k < R[g].K; ++k) {
                    const uint32_t o = (uint32_t)k * (uint32_t)R[g].Nt;
                    const double d = t0[o + ind_f] - t0[o + ind_i];
                    p = fma(nat[k], d, p);
                    if (dF != nullptr && live[g] && d != 0.0)
                        __hip_atomic_fetch_add(&dF[((R[g].feat + k) << cshift) + ccopy], R[g].scale * d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            e = fma(live[g] ? R[g].scale : 0.0, p, e);
        }
    }
    return e;
}

// feature deltas of ONE accepted flip, added to feat[] (wave-reduced per (record, function): the
// reference's p / ratio / J, x size)
template <bool OL, typename UT>
__device__ __forceinline__ void univ_flip_features(const UT &U, const uint8_t *occ, int lane, int s, int newc,
                                                   double *feat) {
    const auto &T = U.T;
    const long long r0 = T.site_ptr[s], r1 = T.site_ptr[s + 1];
    const double *tens = T.corr_mode ? T.corr_tensors : T.interaction_tensors;
    for (long long rr = r0; rr < r1; ++rr) {
        const URec R = U.recs[rr];
        const int *ind = T.loc_idx + R.idx_off;
        for (int k = 0; k < R.K; ++k) {
            const double *t = tens + R.t_off + (size_t)k * R.Nt;
            double p = 0.0;
            for (int j = lane; j < R.J; j += 64) {
                int ind_i = 0, ind_f = 0;
#pragma unroll
                for (int i = 0; i < SMOLMC_MAX_CLUSTER_SITES; ++i)
                    if (i < R.I) {
                        const int x = ind[j * R.I + i];
                        const int v = uocc_ld<OL>(occ, x);
                        const int vf = (x == s) ? newc : v;
                        ind_i += R.st[i] * v;
                        ind_f += R.st[i] * vf;
                    }
                p += t[ind_f] - t[ind_i];
            }
            p = wave_sum(p);
            if (lane == 0) unsafeAtomicAdd(&feat[R.feat + k], p / R.ratio / (double)R.J * (double)T.P); // (no return value: nobody waits)
        }
    }
}

// Ewald delta of one flip from the dense matrix (ewald.pyx:38-58), lane partial
template <bool OL, typename PT>
__device__ __forceinline__ double univ_ewald_dense(const PT &P, const uint8_t *occ, int lane, int s, int oldc, int newc) {
    const int W = P.ew_W;
    const int add = P.ew_inds[(size_t)s * W + newc];
    const int sub = P.ew_inds[(size_t)s * W + oldc];
    const double *radd = P.ew_Mt + (size_t)(add < 0 ? 0 : add) * P.ew_M;
    const double *rsub = P.ew_Mt + (size_t)(sub < 0 ? 0 : sub) * P.ew_M;
    double out = 0;
    for (int k = lane; k < P.N; k += 64) {
        const int v = uocc_ld<OL>(occ, k);
        const int vf = (k == s) ? newc : v;
        const int i = P.ew_inds[(size_t)k * W + vf];
        const int j = (k == s) ? P.ew_inds[(size_t)k * W + v] : i;
        double o = 0;
        if (i != -1 && add != -1) o += (i != add ? 2.0 : 1.0) * radd[i];
        if (j != -1 && sub != -1) o -= (j != sub ? 2.0 : 1.0) * rsub[j];
        out += o;
    }
    return out;
}
// ... from the site kernel G (compact form, mc_general.h): sum over changeable sites k != s of q_k G[s][k]
template <bool OL, typename PT>
__device__ __forceinline__ double univ_ewald_compact(const PT &P, const uint8_t *occ, int lane, int s) {
    const double *g = P.ew_G + (size_t)s * P.ew_nact;
    const int W = P.ew_W, abase = P.ew_act_base;
    double out = 0;
    for (int j = lane; j < P.ew_nact; j += 64) {
        const int k = abase >= 0 ? abase + j : P.ew_act[j];
        const int v = uocc_ld<OL>(occ, k);
        out = fma(k == s ? 0.0 : P.ew_qs[(size_t)k * W + v], g[j], out);
    }
    return out;
}

// WPS: waves per SIMD the register allocation aims at: 4 (128 registers, a few spilled) when the launch has more than
// two walkers per SIMD -- 4096 walkers = 4 waves per SIMD: at 3 resident ones config 2 loses a third --, else 2
// TABLE: the handle's usher is TableFlip (the flip-table machinery is compiled out otherwise);
// K1: every local record has ONE function (interaction mode, or correlation mode with one function per orbit)
template <bool OL, bool K1, bool TABLE, int WPS, bool DL>
__global__ void __launch_bounds__(256, WPS) mc_univ_kernel(const UParams U_kernarg, const int replay) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = uni(blockIdx.x * nwaves + wave);
    const UParamsKernarg Q0 = univ_params(); // (set-up region)
    if (DL) {
        // the dictionaries of the enthalpy pass, one copy per workgroup: distinct records, tensors, natural parameters
        static_assert(sizeof(URecE) == 56, "URecE layout");
        const int nw = Q0->n_recs_e * (int)(sizeof(URecE) / 8), nt = Q0->tens_len, nf = Q0->K.F;
        const double *gt = Q0->T.corr_mode ? Q0->T.corr_tensors : Q0->T.interaction_tensors;
        double *s_rec = (double *)smem, *s_ten = s_rec + SMOLMC_UNIV_DICT_RECS * (sizeof(URecE) / 8), *s_nat = s_ten + SMOLMC_UNIV_DICT_TENS;
        for (int i = threadIdx.x; i < nw; i += blockDim.x) s_rec[i] = ((const double *)Q0->recs_e)[i];
        for (int i = threadIdx.x; i < nt; i += blockDim.x) s_ten[i] = gt[i];
        for (int i = threadIdx.x; i < nf; i += blockDim.x) s_nat[i] = Q0->natural[i];
        __syncthreads(); // (the only block-wide barrier of this kernel)
    }
    if (r >= Q0->K.R) return;
    unsigned char *wp = smem + Q0->lds_shared + (size_t)wave * Q0->lds_per_wave;
    // per-wave scratch: the flips of a table / replayed step, species counts, direction weights
    int *fl_site = (int *)wp;            // [8]
    int *fl_new = fl_site + 8;           // [8]
    int *s_cnt = fl_new + 8 + 16;        // [64] species counts over the active sites ("counts" format)
    int *s_col = s_cnt + 64;             // [8] sites collected for one sublattice
    double *s_mw = (double *)(s_col + 8); // [64] masked direction weights at the current counts
    double *s_mw2 = s_mw + 64;            // [64] ... at the counts after a direction (a-priori factor)
    double *s_lp = s_mw2 + 64;            // [64] a-priori factor of the directions at the current counts
    double *s_cum = s_lp + 64;            // [64] running sums of the masked weights (direction order)
    int last_feas = -1;                   // last direction with a weight
    bool head_valid = false;
    unsigned long long lp_valid = 0ull;
    double sumw_now = 0.0;
    // [dfeat_cells << dfeat_shift] feature deltas of the step in flight (copies of a feature adjacent), then
    // [acc_cells] the feature changes of the accepted steps of this launch: they join the walker's features at
    // the end of the launch (and where a sample / the Wang-Landau statistics read them) -- global atomics and a
    // wait for them on every accepted step otherwise
    const int dcells = Q0->dfeat_cells << Q0->dfeat_shift, cshift = Q0->dfeat_shift;
    // (the TableFlip scratch exists for TableFlip handles only: 4096 walkers need four workgroups per CU, 40 KB each)
    double *s_dF = Q0->dfeat_cells ? (double *)(wp + (TABLE ? SMOLMC_UNIV_SCRATCH_TABLE : SMOLMC_UNIV_SCRATCH)) : nullptr;
    double *s_acc = Q0->dfeat_cells ? s_dF + dcells : nullptr;
    uint8_t *occ = OL ? (uint8_t *)(wp + (TABLE ? SMOLMC_UNIV_SCRATCH_TABLE : SMOLMC_UNIV_SCRATCH)) + (size_t)(dcells + Q0->acc_cells) * 8 : Q0->K.occ + (size_t)r * Q0->K.Npad;
    for (int i = lane; i < dcells + Q0->acc_cells; i += 64) s_dF[i] = 0.0;
    if (OL) {
        const uint4 *src = (const uint4 *)(Q0->K.occ + (size_t)r * Q0->K.Npad);
        for (int i = lane; i < Q0->K.Npad / 16; i += 64) ((uint4 *)occ)[i] = src[i];
    }
    const bool WL = Q0->wl != 0;
    const int F = Q0->K.F, Fce = Q0->K.Fce, nsub = Q0->K.nsub;
    const bool has_ewald = Q0->K.has_ewald != 0, has_mu = Q0->K.has_mu != 0;
    // dense / compact Ewald sweeps read the whole occupancy flip after flip: those models keep the
    // tentative writes of round 4; everything else leaves the occupancy alone until the step is accepted
    const bool seq_occ = has_ewald && !Q0->K.ew_field;
    const int bias_type = Q0->K.bias_type;
    double *feat = Q0->K.features + (size_t)r * F;
    double H = Q0->K.enthalpy[r];
    const double beta = WL ? 0.0 : Q0->K.beta[r];
    unsigned long long step = Q0->K.nsteps[r], nacc = Q0->K.nacc[r];
    const uint32_t key0 = (uint32_t)Q0->K.seeds[r], key1 = (uint32_t)(Q0->K.seeds[r] >> 32);
    double bias = bias_type ? Q0->K.bias[r] : 0.0;
    double *qrow = Q0->K.charge + (size_t)r * SMOLMC_MAX_BIAS_ROWS; // running A_r . n - b_r (one row: net charge)
    int last_acc = 1;
    double wl_m = 0.0;
    long long wl_counter = 0;
    double *wl_S = nullptr, *wl_mf = nullptr;
    long long *wl_Hh = nullptr, *wl_oc = nullptr;
    if (WL) {
        wl_m = Q0->K.wl_m[r];
        wl_counter = Q0->K.wl_counter[r];
        wl_S = Q0->K.wl_entropy + (size_t)r * Q0->K.L;
        wl_Hh = Q0->K.wl_hist + (size_t)r * Q0->K.L;
        wl_oc = Q0->K.wl_occur + (size_t)r * Q0->K.L;
        wl_mf = Q0->K.wl_meanf + (size_t)r * Q0->K.L * F;
    }
    // species counts (table_counts, oracle): dim = sub_code_ptr[sl] + position of the code
    if (TABLE) {
        s_cnt[lane] = 0;
        for (int sl = 0; sl < nsub; ++sl) {
            const int c0 = Q0->K.sub_code_ptr[sl], nc = Q0->K.sub_code_ptr[sl + 1] - c0;
            for (int a = Q0->K.sub_ptr[sl] + lane; a < Q0->K.sub_ptr[sl + 1]; a += 64) {
                const int v = uocc_ld<OL>(occ, Q0->K.sub_sites[a]);
                for (int c = 0; c < nc; ++c)
                    if (Q0->K.sub_codes[c0 + c] == v) atomicAdd(&s_cnt[c0 + c], 1);
            }
        }
    }
    // random words: W(step, block, word).  Flip / Swap ushers: a batch of 16 steps x blocks 0..3, lane
    // 4 (step & 15) + block (as mc_kernel); TableFlip: one step's blocks 0..63, lane = block.  Blocks
    // beyond a batch are evaluated on demand (univ_block).
    uint32_t W0 = 0, W1 = 0, W2 = 0, W3 = 0;
    double logu_b = 0.0; // Flip / Swap batches: log of the acceptance uniform of the lane's step (block-0 lanes)
    unsigned long long batch_base = ~0ull;
    uint32_t wprev1 = 0; // W(step - 1, 0, 1): the site word of the next step
    if (!replay) wprev1 = univ_block(step - 1ull, 0u, key0, key1).w[1];
    const long long nsteps_run = Q0->K.steps_to_run;
    const long long smp_every = Q0->K.smp.every;
    long long smp_countdown = smp_every, smp_index = 0;

    // (sublattice 0 -- the only one of many models -- without the look-ups)
    const int base0 = Q0->K.sub_base[0];
    const uint32_t nact0 = (uint32_t)(Q0->K.sub_ptr[1] - Q0->K.sub_ptr[0]);
    // word j of block blk of the current step (TableFlip batches)
    auto tword = [&](const uint32_t blk, const uint32_t j) -> uint32_t {
        if (blk < 64u) {
            const uint32_t a = rdlane(W0, (int)blk), b = rdlane(W1, (int)blk), c = rdlane(W2, (int)blk), d = rdlane(W3, (int)blk);
            return j == 0 ? a : j == 1 ? b : j == 2 ? c : d;
        }
        const philox_out w = univ_block(step, blk, key0, key1);
        return uni((int)(j == 0 ? w.w[0] : j == 1 ? w.w[1] : j == 2 ? w.w[2] : w.w[3]));
    };

    // feasibility-masked weights of the directions at counts n (+ u of direction `plus`, or -1):
    // table_masked_weights (oracle) / flip_weights_mask (math.py:832-867); returns their sum, added
    // in direction order
    auto masked_weights = [&](const int plus, double *s_mw) -> double {
        const UParamsKernarg Q = univ_params();
        const int d = Q->tf_d, n2 = 2 * Q->tf_n;
        if (lane < n2) {
            const int *row = Q->tf_table + (size_t)(lane >> 1) * d;
            const int sgn = (lane & 1) ? -1 : 1;
            const int *prow = plus >= 0 ? Q->tf_table + (size_t)(plus >> 1) * d : row;
            const int psgn = plus >= 0 ? ((plus & 1) ? -1 : 1) : 0;
            bool ok = true;
            for (int i = 0; i < d; ++i) {
                const int sl = Q->tf_dim_sub[i];
                const int max_n = Q->K.sub_ptr[sl + 1] - Q->K.sub_ptr[sl]; // mcusher.py:497-501
                const int v = s_cnt[i] + psgn * prow[i] + sgn * row[i];
                if (v < 0 || v > max_n) ok = false;
            }
            s_mw[lane] = ok ? Q->tf_w[lane] : 0.0;
        }
        double sum = 0.0;
        for (int idx = 0; idx < n2; ++idx) sum += s_mw[idx];
        return sum;
    };
    // compute_log_priori_factor (mcusher.py:656-711) of direction idx at the current counts; s_mw
    // must hold the masked weights at the current counts and sum_now their sum (table_log_priori, oracle).
    // The counts only change on accepted table steps: the masked weights, their sum and the factor of every
    // direction are kept until then (head_valid, lp_valid, s_lp).
    auto log_priori_of = [&](const int idx, const double sum_now) -> double {
        if ((lp_valid >> idx) & 1ull) return s_lp[idx];
        const UParamsKernarg Q = univ_params();
        const int d = Q->tf_d;
        const double w_now = s_mw[idx];
        const double sum_next = masked_weights(idx, s_mw2);
        const double w_back = s_mw2[idx ^ 1];
        const double p_now = (1.0 - Q->tf_sw) * w_now / sum_now;
        const double p_next = (1.0 - Q->tf_sw) * w_back / sum_next;
        double lf = log(p_next / p_now);
        const int *row = Q->tf_table + (size_t)(idx >> 1) * d;
        const int sgn = (idx & 1) ? -1 : 1;
        for (int i = 0; i < d; ++i) {
            const int u = sgn * row[i], n0 = s_cnt[i];
            for (int k = 1; k <= u; ++k) lf -= Q->tf_ln[n0 + k];
            for (int k = 0; k < -u; ++k) lf += Q->tf_ln[n0 - k];
        }
        lf = uni_d(lf);
        s_lp[idx] = lf; // (every lane, same value)
        lp_valid |= 1ull << idx;
        return lf;
    };
    auto current_weights = [&]() -> double { // masked weights at the current counts (s_mw) and their sum
        if (!head_valid) {
            sumw_now = masked_weights(-1, s_mw);
            head_valid = true;
            lp_valid = 0ull;
            // the running sums choose_section_from_partition forms (math.py:870-893), added in its order: the
            // per-step choice is then one compare + ballot
            const int n2 = 2 * univ_params()->tf_n;
            double cum = 0.0;
            last_feas = -1;
            for (int i = 0; i < n2; ++i) {
                const double m = s_mw[i];
                if (m > 0.0) { cum += m; last_feas = i; }
                s_cum[i] = cum; // (every lane, same value)
            }
        }
        return sumw_now;
    };
    auto site_of = [&](const int sl, const uint32_t k) -> int { // k-th active site of sublattice sl
        if (sl == 0 && base0 >= 0) return base0 + (int)k;
        const UParamsKernarg Q = univ_params();
        const int b = Q->K.sub_base[sl];
        return b >= 0 ? b + (int)k : Q->K.sub_sites[Q->K.sub_ptr[sl] + k];
    };
    auto nact_of = [&](const int sl) -> uint32_t {
        if (sl == 0) return nact0;
        const UParamsKernarg Q = univ_params();
        return (uint32_t)(Q->K.sub_ptr[sl + 1] - Q->K.sub_ptr[sl]);
    };
    // Swap.propose_step (mcusher.py:176-200) on the candidate stream (propose_swap_in, oracle): candidate t
    // < 12 is word t / 3 of block 1 + t % 3, then the words of blocks 4, 5, ... in order.  The first twelve
    // are tried at once, one per lane.  wsrc: the lane of block 0 of this step in the batch.
    // Returns the number of flips (0: no site of another species, mcusher.py:197-199).
    auto propose_swap = [&](const int sl, const uint32_t w_site, const int wsrc, int &s1, int &c1, int &s2, int &c2) -> int {
        const uint32_t nact = nact_of(sl);
        const int site1 = uni(site_of(sl, __umulhi(w_site, nact)));
        const int sp1 = uni(uocc_ld<OL>(occ, site1));
        int site2 = -1;
        {
            const int t = lane < 12 ? lane : 0;
            const int src = wsrc + 1 + t % 3;
            const uint32_t a = (uint32_t)__shfl((int)W0, src), b = (uint32_t)__shfl((int)W1, src), c = (uint32_t)__shfl((int)W2, src),
                           d = (uint32_t)__shfl((int)W3, src);
            const uint32_t wd = t < 3 ? a : t < 6 ? b : t < 9 ? c : d;
            const int cand = site_of(sl, __umulhi(wd, nact));
            const bool ok = lane < 12 && uocc_ld<OL>(occ, cand) != sp1;
            const unsigned long long m = __ballot(ok);
            if (m != 0ull) site2 = (int)rdlane((uint32_t)cand, (int)__builtin_ctzll(m));
        }
        for (uint32_t blk = 4; site2 < 0; ++blk) {
            const philox_out w = univ_block(step, blk, key0, key1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = site_of(sl, __umulhi(w.w[j], nact));
                if (site2 < 0 && uocc_ld<OL>(occ, c) != sp1) site2 = c;
            }
            if (site2 < 0 && (blk == 4u + 63u || ((blk - 4u) & 4095u) == 4095u)) { // swap_options.size == 0 (:197-199)
                int any = 0;
                for (uint32_t a = lane; a < nact; a += 64) any |= uocc_ld<OL>(occ, site_of(sl, a)) != sp1;
                if (__ballot(any) == 0ull) return 0;
            }
        }
        site2 = uni(site2);
        s1 = site1; c1 = uni(uocc_ld<OL>(occ, site2));
        s2 = site2; c2 = sp1;
        return 2;
    };
    auto pick_sublattice = [&](const uint32_t w) -> int { // MCUsher.get_random_sublattice (mcusher.py:146-148)
        const UParamsKernarg Q = univ_params();
        if (nsub == 1) return 0;
        const double x = (double)w * (1.0 / 4294967296.0);
        for (int s = 0; s < nsub; ++s)
            if (x < Q->K.sub_cum[s]) return s;
        return nsub - 1;
    };

    // _get_flip_id (mcusher.py:641-654) over delta_counts_from_step (occu_utils.py:131-168) for the
    // nfl recorded flips (in fl_site / fl_new, not applied): the direction index, or -1 for a canonical swap.
    // bad: 1 = not in the flip table (ValueError, :673-674), 2 = inactive site / impossible code.
    auto step_direction = [&](const int nfl, int &bad) -> int {
        const UParamsKernarg Q = univ_params();
        const int d = Q->tf_d;
        int dn = 0; // lane i < d: change of count i
        for (int f = 0; f < nfl; ++f) {
            const int s = fl_site[f], code = fl_new[f];
            int cur = uocc_ld<OL>(occ, s);
            for (int g = 0; g < f; ++g)
                if (fl_site[g] == s) cur = fl_new[g];
            int dim_ori = -1, dim_nex = -1;
            for (int sl = 0; sl < nsub; ++sl) {
                const int b = Q->K.sub_base[sl], a0 = Q->K.sub_ptr[sl], a1 = Q->K.sub_ptr[sl + 1];
                int in_sl;
                if (b >= 0) in_sl = s >= b && s < b + (a1 - a0);
                else {
                    int hit = 0;
                    for (int a = a0 + lane; a < a1; a += 64) hit |= Q->K.sub_sites[a] == s;
                    in_sl = __ballot(hit) != 0ull;
                }
                if (!in_sl) continue;
                for (int c = Q->K.sub_code_ptr[sl]; c < Q->K.sub_code_ptr[sl + 1]; ++c) {
                    if (Q->K.sub_codes[c] == cur) dim_ori = c;
                    if (Q->K.sub_codes[c] == code) dim_nex = c;
                }
            }
            if (dim_ori < 0 || dim_nex < 0) { bad = 2; return -1; }
            dn += (lane == dim_nex) - (lane == dim_ori);
        }
        if (__ballot(lane < d && dn != 0) == 0ull) return -1;
        for (int vv = 0; vv < Q->tf_n; ++vv) {
            const int e = lane < d ? Q->tf_table[(size_t)vv * d + lane] : 0;
            if (__ballot(lane < d && e != dn) == 0ull) return 2 * vv;
            if (__ballot(lane < d && -e != dn) == 0ull) return 2 * vv + 1;
        }
        bad = 1;
        return -1;
    };

    for (long long it_step = 0; it_step < nsteps_run; ++it_step, ++step) {
        UParamsKernarg Q = univ_params();
        // ================= proposal =====================================================
        // the flips of the step, lane-indexed: lane j < nfl holds site / new species of flip j
        int nfl = 0, vS = 0, vC = 0;
        double lu = 0.0, log_priori = 0.0;
        int dir = -1; // accepted table direction: the counts follow it
        if (replay) {
            const int *rec = Q->K.rp_steps + ((size_t)r * nsteps_run + it_step) * SMOLMC_STEP_ROW;
            int v = lane < SMOLMC_STEP_ROW ? rec[lane] : -1;
            while (nfl < SMOLMC_MAX_STEP_FLIPS && (int)rdlane((uint32_t)v, 2 * nfl) >= 0) nfl++;
            {   // (shuffles in uniform control flow: ds_bpermute reads switched-off lanes as garbage)
                const int a = __shfl(v, 2 * (lane & 7)), b = __shfl(v, 2 * (lane & 7) + 1);
                if (lane < nfl) { vS = a; vC = b; }
                if (TABLE && lane < 8) { fl_site[lane] = a; fl_new[lane] = b; }
            }
            double u = uni_d(Q->K.rp_u[(size_t)r * nsteps_run + it_step]);
            if (u != u) u = 0.0; // NaN: the reference accepted without drawing
            lu = log(u);
            double lp = Q->rp_lp ? uni_d(Q->rp_lp[(size_t)r * nsteps_run + it_step]) : __builtin_nan("");
            if (TABLE && nfl) {
                // the table direction of the step (the counts follow it when the step is accepted) and,
                // when no factor is given, TableFlip.compute_log_priori_factor at the current counts
                int bad = 0;
                const int idx = step_direction(nfl, bad);
                if (bad) {
                    if (lane == 0) atomicOr(Q->rp_err, bad);
                    nfl = 0; // (the call fails; this walker idles through the rest of it)
                    lp = 0.0;
                } else {
                    dir = idx;
                    if (lp != lp) lp = idx >= 0 ? log_priori_of(idx, current_weights()) : 0.0;
                }
            } else if (lp != lp) {
                lp = 0.0; // Flip / Swap ushers: MCUsher.compute_log_priori_factor (mcusher.py:118-134)
            }
            log_priori = lp;
        } else if (!TABLE) {
            const unsigned long long base = step & ~15ull;
            if (base != batch_base) {
                batch_base = base;
                const unsigned long long st = base + (unsigned)(lane >> 2);
                const philox_out o = philox4x32_10((uint32_t)st, (uint32_t)(st >> 32), (uint32_t)(lane & 3), 0u, key0, key1);
                W0 = o.w[0]; W1 = o.w[1]; W2 = o.w[2]; W3 = o.w[3];
                logu_b = log(philox_u53(o.w[2], o.w[3])); // (all 16 steps at once)
            }
            const int l4 = (int)(step & 15ull) * 4;
            lu = __hiloint2double((int)rdlane((uint32_t)__double2hiint(logu_b), l4), (int)rdlane((uint32_t)__double2loint(logu_b), l4));
            const uint32_t w_site = wprev1;
            wprev1 = rdlane(W1, l4);
            // propose_step (oracle): Flip.propose_step (mcusher.py:154-170) / Swap
            const int sl = pick_sublattice(rdlane(W0, l4));
            int s1 = 0, c1 = 0, s2 = 0, c2 = 0;
            if (Q->K.step_type == SMOLMC_STEP_FLIP) {
                const uint32_t nact = nact_of(sl);
                s1 = uni(site_of(sl, __umulhi(w_site, nact)));
                const int c0 = Q->K.sub_code_ptr[sl];
                const uint32_t nc = (uint32_t)(Q->K.sub_code_ptr[sl + 1] - c0);
                const uint32_t kk = __umulhi(rdlane(W0, l4 + 1), nc - 1u);
                const int cur = uni(uocc_ld<OL>(occ, s1));
                int code = -1;
                uint32_t seen = 0;
                for (uint32_t c = 0; c < nc && code < 0; ++c) {
                    const int cc = Q->K.sub_codes[c0 + c];
                    if (cc == cur) continue;
                    if (seen == kk) code = cc;
                    seen++;
                }
                c1 = uni(code);
                nfl = 1;
            } else {
                nfl = propose_swap(sl, w_site, l4, s1, c1, s2, c2);
            }
            vS = lane == 0 ? s1 : lane == 1 ? s2 : 0;
            vC = lane == 0 ? c1 : lane == 1 ? c2 : 0;
            if (lane >= nfl) { vS = 0; vC = 0; }
        } else {
            // TableFlip.propose_step (mcusher.py:553-639): propose_table_flip (oracle)
            {
                const philox_out o = philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), (uint32_t)lane, 0u, key0, key1);
                W0 = o.w[0]; W1 = o.w[1]; W2 = o.w[2]; W3 = o.w[3];
            }
            lu = log(philox_u53(rdlane(W2, 0), rdlane(W3, 0)));
            const uint32_t w_site = wprev1;
            wprev1 = rdlane(W1, 0);
            bool do_swap = (double)rdlane(W0, 0) * (1.0 / 4294967296.0) < Q->tf_sw; // :577-578
            double sumw = 0.0;
            if (!do_swap) {
                sumw = current_weights();
                if (!(sumw > 0.0)) do_swap = true; // no feasible direction: canonical swap only (:604-611)
            }
            if (do_swap) {
                int s1 = 0, c1 = 0, s2 = 0, c2 = 0;
                nfl = propose_swap(pick_sublattice(rdlane(W1, 1)), w_site, 0, s1, c1, s2, c2);
                vS = lane == 0 ? s1 : lane == 1 ? s2 : 0;
                vC = lane == 0 ? c1 : lane == 1 ? c2 : 0;
                if (lane >= nfl) { vS = 0; vC = 0; }
            } else {
                // choose_section_from_partition (math.py:870-893)
                const double target = (double)rdlane(W0, 1) * (1.0 / 4294967296.0) * sumw;
                int idx;
                {   // the first direction with a weight whose running sum exceeds the target, else the last one with a weight
                    const bool in = lane < 2 * Q->tf_n;
                    const double m = in ? s_mw[lane] : 0.0, c = in ? s_cum[lane] : 0.0;
                    const unsigned long long hit = __ballot(in && m > 0.0 && target < c);
                    idx = hit ? (int)__builtin_ctzll(hit) : last_feas;
                }
                const int d = Q->tf_d;
                const int *row = Q->tf_table + (size_t)(idx >> 1) * d;
                const int sgn = (idx & 1) ? -1 : 1;
                log_priori = log_priori_of(idx, sumw);
                dir = idx;
                uint32_t tcand = 0, qdraw = 0;
                for (int sl = 0; sl < nsub; ++sl) {
                    const uint32_t nact = (uint32_t)(Q->K.sub_ptr[sl + 1] - Q->K.sub_ptr[sl]);
                    const int base = Q->K.sub_code_ptr[sl], nc = Q->K.sub_code_ptr[sl + 1] - base;
                    int ncol = 0;
                    for (int c = 0; c < nc; ++c) { // depleted species: -u sites without replacement
                        const int u = uni(sgn * row[base + c]);
                        const int want = uni(Q->K.sub_codes[base + c]);
                        int need = -u;
                        while (need > 0) {
                            // candidates tcand .. tcand + 63 at once (candidate t: word t & 3 of block 4 + t / 4);
                            // the matching ones are taken in order, repeated sites skipped
                            const uint32_t t = tcand + (uint32_t)lane;
                            const uint32_t blk = 4u + t / 4u;
                            const bool inb = blk < 64u;
                            const int src = inb ? (int)blk : 0;
                            const uint32_t a = (uint32_t)__shfl((int)W0, src), b = (uint32_t)__shfl((int)W1, src),
                                           cc = (uint32_t)__shfl((int)W2, src), dd = (uint32_t)__shfl((int)W3, src);
                            const uint32_t wd = (t & 3u) == 0 ? a : (t & 3u) == 1 ? b : (t & 3u) == 2 ? cc : dd;
                            const int site = site_of(sl, __umulhi(wd, nact));
                            const unsigned long long valid = __ballot(inb);
                            if (valid == 0ull) {
                                // beyond the batch (block 64 and later): one candidate at a time
                                const int st1 = uni(site_of(sl, __umulhi(tword(4u + tcand / 4u, tcand & 3u), nact)));
                                tcand++;
                                if (uocc_ld<OL>(occ, st1) != want) continue;
                                int dup = 0;
                                for (int z = 0; z < ncol; ++z) dup |= s_col[z] == st1;
                                if (dup) continue;
                                if (lane == 0) s_col[ncol] = st1;
                                ncol++;
                                need--;
                                continue;
                            }
                            unsigned long long m = __ballot(inb && uocc_ld<OL>(occ, site) == want);
                            const int nvalid = __builtin_popcountll(valid); // (the valid lanes are 0 .. nvalid - 1)
                            int used = nvalid;
                            while (m != 0ull) {
                                const int l = (int)__builtin_ctzll(m);
                                m &= m - 1ull;
                                const int st1 = (int)rdlane((uint32_t)site, l);
                                int dup = 0;
                                for (int z = 0; z < ncol; ++z) dup |= s_col[z] == st1;
                                if (dup) continue;
                                if (lane == 0) s_col[ncol] = st1;
                                ncol++;
                                if (--need == 0) { used = l + 1; break; }
                            }
                            tcand += (uint32_t)used;
                        }
                    }
                    for (int c = 0; c < nc; ++c) { // enriched species: random assignment (:627-631)
                        const int u = uni(sgn * row[base + c]);
                        for (int k = 0; k < u; ++k) {
                            const int rr = (int)__umulhi(tword(2u + qdraw / 4u, qdraw & 3u), (uint32_t)ncol);
                            qdraw++;
                            const int picked = s_col[rr];
                            if (lane == nfl && nfl < SMOLMC_MAX_STEP_FLIPS) {
                                vS = picked;
                                vC = Q->K.sub_codes[base + c];
                            }
                            nfl++;
                            const int moved = lane < 8 && lane >= rr && lane + 1 < ncol ? s_col[lane + 1] : 0;
                            if (lane < 8 && lane >= rr && lane + 1 < ncol) s_col[lane] = moved;
                            ncol--;
                        }
                    }
                }
            }
        }

        Q = univ_params();
        // ================= enthalpy change of the step ======================================
        // species before the STEP (mu / bias terms) and before THIS flip (sequential), lane-indexed
        const int vOrig = uocc_ld<OL>(occ, vS);
        int vOld = vOrig;
        uint32_t vQ0 = 0, vCum = 0, total = 0;
        {
            const uint32_t ru = (uint32_t)Q->rows_uniform; // (every site has ru rows: no look-up)
            const uint32_t q0 = ru ? (uint32_t)vS * ru : Q->row_ptr[vS], q1 = ru ? q0 + ru : Q->row_ptr[vS + 1];
            vQ0 = q0;
            const uint32_t len = lane < nfl ? q1 - q0 : 0u;
            for (int j = 0; j < nfl; ++j) {
                if (lane == j) vCum = total;
                total += rdlane(len, j);
                if (j + 1 < nfl) {
                    const int sj = (int)rdlane((uint32_t)vS, j), cj = (int)rdlane((uint32_t)vC, j);
                    if (lane > j && vS == sj) vOld = cj;
                }
            }
        }
        auto S_ = [&](const int f) -> int { return (int)rdlane((uint32_t)vS, f); };
        auto C_ = [&](const int f) -> int { return (int)rdlane((uint32_t)vC, f); };
        auto OLD_ = [&](const int f) -> int { return (int)rdlane((uint32_t)vOld, f); };
        auto ORIG_ = [&](const int f) -> int { return (int)rdlane((uint32_t)vOrig, f); };
        auto RD_ = [&](const double v, const int f) -> double { // lane f's value
            return __hiloint2double((int)rdlane((uint32_t)__double2hiint(v), f), (int)rdlane((uint32_t)__double2loint(v), f));
        };
        double e = 0.0, ew_part = 0.0, ew_uni = 0.0, dMu = 0.0;
        double vdq = 0.0; // potential-field mode: lane f holds the charge change of flip f
        if (total) {
            if (nfl <= 2) e = univ_step_partial<OL, K1, DL, 2>(smem, occ, lane, nfl, vS, vC, vQ0, vCum, total, s_dF, cshift);
            else e = univ_step_partial<OL, K1, DL, SMOLMC_MAX_STEP_FLIPS>(smem, occ, lane, nfl, vS, vC, vQ0, vCum, total, s_dF, cshift);
        }
        if (has_ewald) {
            const int W = Q->K.ew_W;
            if (Q->K.ew_field) {
                // O(1) per flip from the walker's potential field + the cross terms of the earlier flips; lane f works
                // on flip f (one round trip for the whole step), the terms are added in flip order
                const int ab = Q->K.ew_act_base, na = Q->K.ew_nact;
                const double *phi = Q->K.ew_phi + (size_t)r * na;
                const bool act = lane < nfl;
                const int s = act ? vS : ab;
                const double dq = act ? Q->K.ew_qs[(size_t)s * W + vC] - Q->K.ew_qs[(size_t)s * W + vOld] : 0.0;
                double pot = __hip_atomic_load(&phi[s - ab], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vdq = dq;
#pragma unroll
                for (int g = 0; g + 1 < SMOLMC_MAX_STEP_FLIPS; ++g)
                    if (g + 1 < nfl) {
                        const int sg = S_(g);
                        const double dqg = RD_(dq, g);
                        const double gg = Q->K.ew_G[(size_t)s * na + (sg - ab)];
                        if (g < lane && sg != s) pot = fma(dqg, gg, pot);
                    }
                const double term = act ? 2.0 * dq * pot + (Q->K.ew_dg[(size_t)s * W + vC] - Q->K.ew_dg[(size_t)s * W + vOld]) : 0.0;
                for (int f = 0; f < nfl; ++f) ew_uni += RD_(term, f);
            } else {
                for (int f = 0; f < nfl; ++f) {
                    const int s = S_(f), newc = C_(f), oldc = OLD_(f);
                    if (Q->K.ew_compact) {
                        const double dq = Q->K.ew_qs[(size_t)s * W + newc] - Q->K.ew_qs[(size_t)s * W + oldc];
                        ew_part += 2.0 * dq * univ_ewald_compact<OL>(Q->K, occ, lane, s);
                        ew_uni += 2.0 * dq * Q->K.ew_frozen[s] + (Q->K.ew_dg[(size_t)s * W + newc] - Q->K.ew_dg[(size_t)s * W + oldc]);
                    } else {
                        ew_part += univ_ewald_dense<OL>(Q->K, occ, lane, s, oldc, newc);
                    }
                    uocc_st<OL>(occ, s, newc, lane); // tentative (seq_occ)
                }
            }
        }
        if (has_mu) {
            const double term = lane < nfl ? Q->K.mu[(size_t)vS * Q->K.mu_W + vC] - Q->K.mu[(size_t)vS * Q->K.mu_W + vOrig] : 0.0; // ensemble.py:368-374
            for (int f = 0; f < nfl; ++f) dMu += RD_(term, f);
        }
        double dH = wave_sum(e);
        double dEw = 0.0;
        if (has_ewald) {
            dEw = uni_d(wave_sum(ew_part) + ew_uni);
            dH += Q->K.ew_coef * dEw;
        }
        if (has_mu) { dMu = uni_d(dMu); dH -= dMu; }
        // MCBias.compute_bias_change (kernel/base.py:307-311; orc_compute_bias_change): the last flip of a
        // site counts, against the species before the step
        double dB = 0.0, dq_row[SMOLMC_MAX_BIAS_ROWS] = {0.0, 0.0, 0.0, 0.0}, qrow_now[SMOLMC_MAX_BIAS_ROWS] = {0.0, 0.0, 0.0, 0.0};
        if (bias_type && nfl) {
            const int W = Q->K.bias_W;
            if (bias_type == SMOLMC_BIAS_FUGACITY) {
                for (int f = 0; f < nfl; ++f) {
                    const int s = S_(f);
                    int last = 1;
                    for (int g = f + 1; g < nfl; ++g) last &= S_(g) != s;
                    if (!last) continue;
                    dB += log(Q->K.bias_tab[(size_t)s * W + C_(f)] / Q->K.bias_tab[(size_t)s * W + ORIG_(f)]);
                }
            } else {
                double sq_new = 0.0, sq_old = 0.0;
#pragma unroll
                for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k) { // (constant indices: dq_row stays in registers)
                    if (k >= Q->K.bias_rows) continue;
                    const double *tab = Q->K.bias_tab + (size_t)k * Q->K.bias_row_stride;
                    const double c = __hip_atomic_load(&qrow[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    double cn = c;
                    for (int f = 0; f < nfl; ++f) {
                        const int s = S_(f);
                        int last = 1;
                        for (int g = f + 1; g < nfl; ++g) last &= S_(g) != s;
                        if (!last) continue;
                        cn += tab[(size_t)s * W + C_(f)] - tab[(size_t)s * W + ORIG_(f)];
                    }
                    dq_row[k] = cn - c;
                    qrow_now[k] = c;
                    sq_old += c * c;
                    sq_new += cn * cn;
                }
                dB = -Q->K.bias_pen * sq_new - (-Q->K.bias_pen * sq_old);
            }
            dB = uni_d(dB);
        }

        Q = univ_params();
        // ================= accept ========================================================
        bool accepted;
        if (!WL) {
            double exponent = -beta * dH + log_priori; // metropolis.py:41-42
            if (bias_type) exponent += dB;               // :43-44
            accepted = __ballot(exponent >= 0.0 ? true : (exponent > lu)) != 0ull;
        } else {
            const double new_h = H + dH; // wanglandau.py:188
            if (new_h < Q->K.wl_min || new_h >= Q->K.wl_max) {
                accepted = false;
            } else {
                const long long b = (long long)floordiv_exact(H - Q->K.wl_min, Q->K.wl_bin);
                const long long nb = (long long)floordiv_exact(new_h - Q->K.wl_min, Q->K.wl_bin);
                const double Sb = __hip_atomic_load(&wl_S[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double Snb = __hip_atomic_load(&wl_S[nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double exponent = Sb - Snb + log_priori; // :197-198
                accepted = __ballot(exponent >= 0.0 ? true : (exponent > lu)) != 0ull;
            }
        }

        // ================= update ==========================================================
        if (accepted && nfl) {
            if (s_dF != nullptr) {
                // the cells filled by the enthalpy pass join the walker's features
                for (int i = lane; i < Fce; i += 64) {
                    double v = 0.0;
                    for (int c = 0; c < (1 << cshift); ++c) {
                        v += s_dF[(i << cshift) + c];
                        s_dF[(i << cshift) + c] = 0.0;
                    }
                    s_acc[i] += v;
                }
                if (!seq_occ)
                    for (int f = 0; f < nfl; ++f) uocc_st_nf<OL>(occ, S_(f), C_(f), lane);
            } else {
                // pass 2: feature deltas flip by flip against the occupancy each flip saw
                if (seq_occ)
                    for (int f = nfl - 1; f >= 0; --f) uocc_st<OL>(occ, S_(f), OLD_(f), lane);
                for (int f = 0; f < nfl; ++f) {
                    univ_flip_features<OL>(*Q, occ, lane, S_(f), C_(f), feat);
                    uocc_st<OL>(occ, S_(f), C_(f), lane);
                }
            }
            if (lane == 0) {
                if (s_acc != nullptr) {
                    if (has_ewald) s_acc[Fce] += dEw;
                    if (has_mu) s_acc[Fce + (has_ewald ? 1 : 0)] += dMu;
                } else {
                    if (has_ewald) unsafeAtomicAdd(&feat[Fce], dEw);
                    if (has_mu) unsafeAtomicAdd(&feat[Fce + (has_ewald ? 1 : 0)], dMu);
                }
            }
            if (has_ewald && Q->K.ew_field) {
                // (flip after flip through the tuned sweep: one fused pass over the field with the flips in an inner loop
                // measured slower -- config 5 forced universal 73 -> 98 ms)
                double *phi = Q->K.ew_phi + (size_t)r * Q->K.ew_nact;
                for (int f = 0; f < nfl; ++f) {
                    const double dq = RD_(vdq, f);
                    if (dq != 0.0) field_apply_global(Q->K, phi, lane, S_(f), dq);
                }
            }
            if (TABLE && dir >= 0 && lane < Q->tf_d) {
                const int sgn = (dir & 1) ? -1 : 1;
                s_cnt[lane] += sgn * Q->tf_table[(size_t)(dir >> 1) * Q->tf_d + lane];
            }
            if (TABLE && dir >= 0) head_valid = false; // the counts changed
            H += dH;
            bias += dB;
            if (bias_type && bias_type != SMOLMC_BIAS_FUGACITY && lane == 0) {
#pragma unroll
                for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k)
                    if (k < Q->K.bias_rows) __hip_atomic_store(&qrow[k], qrow_now[k] + dq_row[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // lane 0's occupancy / feature / charge stores have reached L2 before the wave reads them there
            // (the readers load past the L1): a wait, no cache maintenance
            // (nothing of the kind was written -- occupancy and feature changes in LDS, no charge row, no potential
            // field --: no wait)
            if (!OL) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            else if (s_acc == nullptr || (bias_type && bias_type != SMOLMC_BIAS_FUGACITY) || (has_ewald && Q->K.ew_field))
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            nacc++;
        } else if (accepted) {
            nacc++; // the empty step is accepted (metropolis.py:46)
        } else {
            if (seq_occ)
                for (int f = nfl - 1; f >= 0; --f) uocc_st<OL>(occ, S_(f), OLD_(f), lane);
            if (s_dF != nullptr && nfl)
                for (int i = lane; i < (Fce << cshift); i += 64) s_dF[i] = 0.0;
        }
        last_acc = accepted ? 1 : 0;

        if (WL) {
            // WangLandau._do_post_step (wanglandau.py:222-266)
            const double bq = floordiv_exact(H - Q->K.wl_min, Q->K.wl_bin);
            if (bq >= 0.0 && bq < (double)Q->K.L) {
                const long long b = (long long)bq;
                wl_counter++;
                long long total = 0;
                if (lane == 0) total = wl_oc[b];
                total = ((long long)(unsigned)uni((int)(total >> 32)) << 32) | (unsigned)uni((int)(total & 0xffffffffll));
                const double inv = 1.0 / (double)(total + 1);
                for (int i = lane; i < F; i += 64) {
                    double *mf = wl_mf + (size_t)b * F + i;
                    const double cf = __hip_atomic_load(&feat[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (s_acc != nullptr ? s_acc[i] : 0.0);
                    *mf = inv * (cf + (double)total * (*mf));
                }
                if (wl_counter % Q->K.wl_update == 0 && lane == 0) {
                    wl_S[b] += wl_m;
                    wl_Hh[b] += 1;
                    wl_oc[b] = total + 1;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); // lane 0's entropy update before the next step's reads
            }
            if (Q->K.wl_check != 0 && wl_counter % Q->K.wl_check == 0) { // (check period 0: no device-side check)
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                long cnt = 0;
                double sum = 0;
                for (int i = lane; i < Q->K.L; i += 64) {
                    const double Si = __hip_atomic_load(&wl_S[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const long long Hi = __hip_atomic_load(&wl_Hh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (Si > 0) { cnt++; sum += (double)Hi; }
                }
                const double tcnt = wave_sum((double)cnt), tsum = wave_sum(sum);
                if (tcnt >= 2.0) {
                    const double thr = Q->K.wl_flat * (tsum / tcnt);
                    int bad = 0;
                    for (int i = lane; i < Q->K.L; i += 64) {
                        const double Si = __hip_atomic_load(&wl_S[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const long long Hi = __hip_atomic_load(&wl_Hh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (Si > 0 && !((double)Hi > thr)) bad = 1;
                    }
                    if (__ballot(bad) == 0ull) {
                        for (int i = lane; i < Q->K.L; i += 64)
                            __hip_atomic_store(&wl_Hh[i], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        wl_m = wl_m / Q->K.wl_div;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            }
        }
        if (replay && lane == 0) {
            const size_t k = (size_t)r * nsteps_run + it_step;
            if (Q->K.rp_acc) Q->K.rp_acc[k] = (uint8_t)last_acc;
            if (Q->K.rp_H) Q->K.rp_H[k] = H;
            if (Q->rp_lp_out) Q->rp_lp_out[k] = log_priori;
        }
        if (smp_every && --smp_countdown == 0) { // one thinned sample of this walker
            smp_countdown = smp_every;
            const size_t row = (size_t)smp_index * Q->K.R + r;
            smp_index++;
            for (int i = lane; i < F; i += 64)
                Q->K.smp.feat[row * F + i] = __hip_atomic_load(&feat[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (s_acc != nullptr ? s_acc[i] : 0.0);
            if (lane == 0) {
                Q->K.smp.H[row] = H;
                Q->K.smp.acc[row] = (uint8_t)last_acc;
            }
            if (Q->K.smp.occ) {
                uint8_t *dst = Q->K.smp.occ + row * Q->K.Npad;
                for (int i = lane; i < Q->K.Npad; i += 64) dst[i] = (uint8_t)uocc_ld<OL>(occ, i);
            }
        }
    }

    // ---- write the chain back -----------------------------------------------------------
    const UParamsKernarg Q = univ_params();
    if (s_acc != nullptr)
        for (int i = lane; i < F; i += 64) feat[i] += s_acc[i];
    if (OL) {
        uint4 *dst = (uint4 *)(Q->K.occ + (size_t)r * Q->K.Npad);
        for (int i = lane; i < Q->K.Npad / 16; i += 64) dst[i] = ((const uint4 *)occ)[i];
    }
    if (lane == 0) {
        Q->K.enthalpy[r] = H;
        Q->K.nsteps[r] = step;
        Q->K.nacc[r] = nacc;
        Q->K.last_acc[r] = (uint8_t)last_acc;
        if (bias_type) Q->K.bias[r] = bias;
        if (WL) {
            Q->K.wl_m[r] = wl_m;
            Q->K.wl_counter[r] = wl_counter;
        }
    }
}


// the instantiation of a launch: sel = occupancy in LDS << 3 | one function per record << 2 | TableFlip << 1 | more
// than two walkers per SIMD (WPS 4)
typedef void (*univ_kernel_fn)(const UParams, const int);
template <bool DL> static univ_kernel_fn univ_select(const int sel) {
    switch (sel) {
#define UNIV_CASE(n, OLv, K1v, TBv, WPv) case n: return mc_univ_kernel<OLv, K1v, TBv, WPv, DL>;
        UNIV_CASE(0, false, false, false, 2) UNIV_CASE(1, false, false, false, 4)
        UNIV_CASE(2, false, false, true, 2) UNIV_CASE(3, false, false, true, 4)
        UNIV_CASE(4, false, true, false, 2) UNIV_CASE(5, false, true, false, 4)
        UNIV_CASE(6, false, true, true, 2) UNIV_CASE(7, false, true, true, 4)
        UNIV_CASE(8, true, false, false, 2) UNIV_CASE(9, true, false, false, 4)
        UNIV_CASE(10, true, false, true, 2) UNIV_CASE(11, true, false, true, 4)
        UNIV_CASE(12, true, true, false, 2) UNIV_CASE(13, true, true, false, 4)
        UNIV_CASE(14, true, true, true, 2) UNIV_CASE(15, true, true, true, 4)
#undef UNIV_CASE
    }
    return nullptr;
}
univ_kernel_fn smolmc_univ_kernel_dict(int sel); // (univ_dict.hip)
