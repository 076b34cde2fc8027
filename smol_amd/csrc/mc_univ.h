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

__device__ __forceinline__ philox_out univ_block(unsigned long long step, uint32_t block, uint32_t k0, uint32_t k1) {
    return philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), block, 0u, k0, k1);
}

// lane-partial enthalpy change of ONE flip at site s (old -> new) against the current occupancy:
// sum over the cluster rows of s and their correlation functions of
// natural[feature] * (t[ind_f] - t[ind_i]) * size / ratio / J.
// The rows of ALL local records of the site are dealt to the lanes (row table built at create), two
// groups of 64 in flight: a row costs a chain of dependent round trips (row -> record descriptor ->
// member sites -> species -> tensor entries), and record by record that chain was all a wave did
// (round 4: 50 us per step on config 2).
// dF (or null): LDS cells of the step's feature deltas; every row adds scale * (t_k[ind_f] - t_k[ind_i])
// to the cell of its feature (committed to the walker's features when the step is accepted).
template <bool OL>
__device__ __forceinline__ double univ_flip_partial(const UParams &U, const uint8_t *occ, int lane, int s, int newc, double *dF) {
    const RefTables &T = U.T;
    double e = 0.0;
    const long long q0 = U.row_ptr[s], q1 = U.row_ptr[s + 1];
    const double *tens = T.corr_mode ? T.corr_tensors : T.interaction_tensors;
    constexpr int G = 2;
    for (long long qb = q0; qb < q1; qb += 64 * G) {
        int rec[G];
        long long off[G];
        bool live[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const long long q = qb + 64 * g + lane;
            live[g] = q < q1;
            const long long qq = live[g] ? q : q1 - 1; // (clamped: loads only)
            rec[g] = U.row_rec[qq];
            off[g] = U.row_off[qq];
        }
        URec R[G];
#pragma unroll
        for (int g = 0; g < G; ++g) R[g] = U.recs[rec[g]];
        int x[G][SMOLMC_MAX_CLUSTER_SITES];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < SMOLMC_MAX_CLUSTER_SITES; ++i) x[g][i] = i < R[g].I ? T.loc_idx[off[g] + i] : s;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            int ind_i = 0, ind_f = 0;
#pragma unroll
            for (int i = 0; i < SMOLMC_MAX_CLUSTER_SITES; ++i)
                if (i < R[g].I) {
                    const int v = uocc_ld<OL>(occ, x[g][i]);
                    const int vf = (x[g][i] == s) ? newc : v;
                    ind_i += R[g].st[i] * v;
                    ind_f += R[g].st[i] * vf;
                }
            const double *t0 = tens + R[g].t_off;
            const double *nat = U.natural + R[g].feat;
            double p = 0.0;
            for (int k = 0; k < R[g].K; ++k) {
                const double d = t0[(size_t)k * R[g].Nt + ind_f] - t0[(size_t)k * R[g].Nt + ind_i];
                p = fma(nat[k], d, p);
                if (dF != nullptr && live[g] && d != 0.0)
                    __hip_atomic_fetch_add(&dF[R[g].feat + k], R[g].scale * d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            e = fma(live[g] ? R[g].scale : 0.0, p, e);
        }
    }
    return e;
}

// feature deltas of ONE accepted flip, added to feat[] (wave-reduced per (record, function): the
// reference's p / ratio / J, x size)
template <bool OL>
__device__ __forceinline__ void univ_flip_features(const UParams &U, const uint8_t *occ, int lane, int s, int newc,
                                                   double *feat) {
    const RefTables &T = U.T;
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
template <bool OL>
__device__ __forceinline__ double univ_ewald_dense(const KParams &P, const uint8_t *occ, int lane, int s, int oldc, int newc) {
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
template <bool OL>
__device__ __forceinline__ double univ_ewald_compact(const KParams &P, const uint8_t *occ, int lane, int s) {
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

template <bool OL>
__global__ void __launch_bounds__(256) mc_univ_kernel(const UParams U, const int replay) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = uni(blockIdx.x * nwaves + wave);
    const KParams &P = U.K;
    const RefTables &T = U.T;
    if (r >= P.R) return; // (no block-wide barrier in this kernel)
    unsigned char *wp = smem + (size_t)wave * U.lds_per_wave;
    // per-wave scratch: the flips of the step, species counts, direction weights
    int *fl_site = (int *)wp;            // [8]
    int *fl_new = fl_site + 8;           // [8]
    int *fl_old = fl_new + 8;            // [8] species before THIS flip (sequential)
    int *fl_orig = fl_old + 8;           // [8] species before the STEP (mu / bias terms)
    int *s_cnt = fl_orig + 8;            // [64] species counts over the active sites ("counts" format)
    int *s_col = s_cnt + 64;             // [8] sites collected for one sublattice
    double *s_mw = (double *)(s_col + 8); // [64] masked direction weights
    double *s_dF = U.dfeat_cells ? s_mw + 64 : nullptr; // [dfeat_cells] feature deltas of the step in flight
    uint8_t *occ = OL ? (uint8_t *)(s_mw + 64 + U.dfeat_cells) : P.occ + (size_t)r * P.Npad;
    for (int i = lane; i < U.dfeat_cells; i += 64) s_dF[i] = 0.0;
    if (OL) {
        const uint4 *src = (const uint4 *)(P.occ + (size_t)r * P.Npad);
        for (int i = lane; i < P.Npad / 16; i += 64) ((uint4 *)occ)[i] = src[i];
    }
    const bool WL = U.wl != 0;
    const bool table = P.step_type == SMOLMC_STEP_TABLE_FLIP;
    const int F = P.F, Fce = P.Fce, nsub = P.nsub;
    const bool has_ewald = P.has_ewald != 0, has_mu = P.has_mu != 0;
    const int bias_type = P.bias_type;
    double *feat = P.features + (size_t)r * F;
    double H = P.enthalpy[r];
    const double beta = WL ? 0.0 : P.beta[r];
    unsigned long long step = P.nsteps[r], nacc = P.nacc[r];
    const uint32_t key0 = (uint32_t)P.seeds[r], key1 = (uint32_t)(P.seeds[r] >> 32);
    double bias = bias_type ? P.bias[r] : 0.0;
    double *qrow = P.charge + (size_t)r * SMOLMC_MAX_BIAS_ROWS; // running A_r . n - b_r (one row: net charge)
    int last_acc = 1;
    double wl_m = 0.0;
    long long wl_counter = 0;
    double *wl_S = nullptr, *wl_mf = nullptr;
    long long *wl_Hh = nullptr, *wl_oc = nullptr;
    if (WL) {
        wl_m = P.wl_m[r];
        wl_counter = P.wl_counter[r];
        wl_S = P.wl_entropy + (size_t)r * P.L;
        wl_Hh = P.wl_hist + (size_t)r * P.L;
        wl_oc = P.wl_occur + (size_t)r * P.L;
        wl_mf = P.wl_meanf + (size_t)r * P.L * F;
    }
    // species counts (table_counts, oracle): dim = sub_code_ptr[sl] + position of the code
    if (table) {
        s_cnt[lane] = 0;
        for (int sl = 0; sl < nsub; ++sl) {
            const int c0 = P.sub_code_ptr[sl], nc = P.sub_code_ptr[sl + 1] - c0;
            for (int a = P.sub_ptr[sl] + lane; a < P.sub_ptr[sl + 1]; a += 64) {
                const int v = uocc_ld<OL>(occ, P.sub_sites[a]);
                for (int c = 0; c < nc; ++c)
                    if (P.sub_codes[c0 + c] == v) atomicAdd(&s_cnt[c0 + c], 1);
            }
        }
    }
    uint32_t wprev1 = 0; // W(step - 1, 0, 1): the site word of the next step
    if (!replay) wprev1 = univ_block(step - 1ull, 0u, key0, key1).w[1];
    const long long nsteps_run = P.steps_to_run;
    const long long smp_every = P.smp.every;
    long long smp_countdown = smp_every, smp_index = 0;

    // feasibility-masked weights of the directions at counts n (+ u of direction `plus`, or -1):
    // table_masked_weights (oracle) / flip_weights_mask (math.py:832-867); returns their sum, added
    // in direction order
    auto masked_weights = [&](const int plus) -> double {
        const int d = U.tf_d, n2 = 2 * U.tf_n;
        if (lane < n2) {
            const int *row = U.tf_table + (size_t)(lane >> 1) * d;
            const int sgn = (lane & 1) ? -1 : 1;
            const int *prow = plus >= 0 ? U.tf_table + (size_t)(plus >> 1) * d : row;
            const int psgn = plus >= 0 ? ((plus & 1) ? -1 : 1) : 0;
            bool ok = true;
            for (int i = 0; i < d; ++i) {
                const int sl = U.tf_dim_sub[i];
                const int max_n = P.sub_ptr[sl + 1] - P.sub_ptr[sl]; // mcusher.py:497-501
                const int v = s_cnt[i] + psgn * prow[i] + sgn * row[i];
                if (v < 0 || v > max_n) ok = false;
            }
            s_mw[lane] = ok ? U.tf_w[lane] : 0.0;
        }
        double sum = 0.0;
        for (int idx = 0; idx < n2; ++idx) sum += s_mw[idx];
        return sum;
    };
    // compute_log_priori_factor (mcusher.py:656-711) of direction idx at the current counts; s_mw
    // must hold the masked weights at the current counts and sum_now their sum (table_log_priori, oracle)
    auto log_priori_of = [&](const int idx, const double sum_now) -> double {
        const int d = U.tf_d;
        const double w_now = s_mw[idx];
        const double sum_next = masked_weights(idx);
        const double w_back = s_mw[idx ^ 1];
        const double p_now = (1.0 - U.tf_sw) * w_now / sum_now;
        const double p_next = (1.0 - U.tf_sw) * w_back / sum_next;
        double lf = log(p_next / p_now);
        const int *row = U.tf_table + (size_t)(idx >> 1) * d;
        const int sgn = (idx & 1) ? -1 : 1;
        for (int i = 0; i < d; ++i) {
            const int u = sgn * row[i], n0 = s_cnt[i];
            for (int k = 1; k <= u; ++k) lf -= U.tf_ln[n0 + k];
            for (int k = 0; k < -u; ++k) lf += U.tf_ln[n0 - k];
        }
        return uni_d(lf);
    };
    auto site_of = [&](const int sl, const uint32_t k) -> int { // k-th active site of sublattice sl
        const int b = P.sub_base[sl];
        return b >= 0 ? b + (int)k : P.sub_sites[P.sub_ptr[sl] + k];
    };
    // Swap.propose_step (mcusher.py:176-200) on the candidate stream (propose_swap_in, oracle)
    auto propose_swap = [&](const int sl, const uint32_t w_site) -> int {
        const uint32_t nact = (uint32_t)(P.sub_ptr[sl + 1] - P.sub_ptr[sl]);
        const int site1 = site_of(sl, __umulhi(w_site, nact));
        const int sp1 = uocc_ld<OL>(occ, site1);
        int site2 = -1;
        {
            philox_out w[3];
            for (int b = 0; b < 3; ++b) w[b] = univ_block(step, 1u + (uint32_t)b, key0, key1);
            for (int t = 0; t < 12 && site2 < 0; ++t) {
                const int c = site_of(sl, __umulhi(w[t % 3].w[t / 3], nact));
                if (uocc_ld<OL>(occ, c) != sp1) site2 = c;
            }
        }
        for (uint32_t blk = 4; site2 < 0; ++blk) {
            const philox_out w = univ_block(step, blk, key0, key1);
            for (int j = 0; j < 4 && site2 < 0; ++j) {
                const int c = site_of(sl, __umulhi(w.w[j], nact));
                if (uocc_ld<OL>(occ, c) != sp1) site2 = c;
            }
            if (site2 < 0 && (blk == 4u + 63u || ((blk - 4u) & 4095u) == 4095u)) { // swap_options.size == 0 (:197-199)
                int any = 0;
                for (uint32_t a = lane; a < nact; a += 64) any |= uocc_ld<OL>(occ, site_of(sl, a)) != sp1;
                if (__ballot(any) == 0ull) return 0;
            }
        }
        site2 = uni(site2);
        const int sp2 = uocc_ld<OL>(occ, site2);
        if (lane == 0) {
            fl_site[0] = site1; fl_new[0] = sp2;
            fl_site[1] = site2; fl_new[1] = sp1;
        }
        return 2;
    };
    auto pick_sublattice = [&](const uint32_t w) -> int { // MCUsher.get_random_sublattice (mcusher.py:146-148)
        if (nsub == 1) return 0;
        const double x = (double)w * (1.0 / 4294967296.0);
        for (int s = 0; s < nsub; ++s)
            if (x < P.sub_cum[s]) return s;
        return nsub - 1;
    };

    // _get_flip_id (mcusher.py:641-654) over delta_counts_from_step (occu_utils.py:131-168) for the
    // nfl recorded flips (not applied yet): the direction index, or -1 for a canonical swap.
    // bad: 1 = not in the flip table (ValueError, :673-674), 2 = inactive site / impossible code.
    auto step_direction = [&](const int nfl, int &bad) -> int {
        const int d = U.tf_d;
        int dn = 0; // lane i < d: change of count i
        for (int f = 0; f < nfl; ++f) {
            const int s = fl_site[f], code = fl_new[f];
            int cur = uocc_ld<OL>(occ, s);
            for (int g = 0; g < f; ++g)
                if (fl_site[g] == s) cur = fl_new[g];
            int dim_ori = -1, dim_nex = -1;
            for (int sl = 0; sl < nsub; ++sl) {
                const int b = P.sub_base[sl], a0 = P.sub_ptr[sl], a1 = P.sub_ptr[sl + 1];
                int in_sl;
                if (b >= 0) in_sl = s >= b && s < b + (a1 - a0);
                else {
                    int hit = 0;
                    for (int a = a0 + lane; a < a1; a += 64) hit |= P.sub_sites[a] == s;
                    in_sl = __ballot(hit) != 0ull;
                }
                if (!in_sl) continue;
                for (int c = P.sub_code_ptr[sl]; c < P.sub_code_ptr[sl + 1]; ++c) {
                    if (P.sub_codes[c] == cur) dim_ori = c;
                    if (P.sub_codes[c] == code) dim_nex = c;
                }
            }
            if (dim_ori < 0 || dim_nex < 0) { bad = 2; return -1; }
            dn += (lane == dim_nex) - (lane == dim_ori);
        }
        if (__ballot(lane < d && dn != 0) == 0ull) return -1;
        for (int vv = 0; vv < U.tf_n; ++vv) {
            const int e = lane < d ? U.tf_table[(size_t)vv * d + lane] : 0;
            if (__ballot(lane < d && e != dn) == 0ull) return 2 * vv;
            if (__ballot(lane < d && -e != dn) == 0ull) return 2 * vv + 1;
        }
        bad = 1;
        return -1;
    };

    for (long long it_step = 0; it_step < nsteps_run; ++it_step, ++step) {
        // ================= proposal =====================================================
        int nfl = 0;
        double lu = 0.0, log_priori = 0.0;
        int dir = -1; // accepted table direction: the counts follow it
        if (replay) {
            const int *rec = P.rp_steps + ((size_t)r * nsteps_run + it_step) * SMOLMC_STEP_ROW;
            int v = lane < SMOLMC_STEP_ROW ? rec[lane] : -1;
            while (nfl < SMOLMC_MAX_STEP_FLIPS && (int)rdlane((uint32_t)v, 2 * nfl) >= 0) nfl++;
            {   // (shuffles in uniform control flow: ds_bpermute reads switched-off lanes as garbage)
                const int a = __shfl(v, 2 * (lane & 7)), b = __shfl(v, 2 * (lane & 7) + 1);
                if (lane < 8) { fl_site[lane] = a; fl_new[lane] = b; }
            }
            double u = uni_d(P.rp_u[(size_t)r * nsteps_run + it_step]);
            if (u != u) u = 0.0; // NaN: the reference accepted without drawing
            lu = log(u);
            double lp = U.rp_lp ? uni_d(U.rp_lp[(size_t)r * nsteps_run + it_step]) : __builtin_nan("");
            if (table && nfl) {
                // the table direction of the step (the counts follow it when the step is accepted) and,
                // when no factor is given, TableFlip.compute_log_priori_factor at the current counts
                int bad = 0;
                const int idx = step_direction(nfl, bad);
                if (bad) {
                    if (lane == 0) atomicOr(U.rp_err, bad);
                    nfl = 0; // (the call fails; this walker idles through the rest of it)
                    lp = 0.0;
                } else {
                    dir = idx;
                    if (lp != lp) lp = idx >= 0 ? log_priori_of(idx, masked_weights(-1)) : 0.0;
                }
            } else if (lp != lp) {
                lp = 0.0; // Flip / Swap ushers: MCUsher.compute_log_priori_factor (mcusher.py:118-134)
            }
            log_priori = lp;
        } else {
            const philox_out w0 = univ_block(step, 0u, key0, key1);
            lu = log(philox_u53(w0.w[2], w0.w[3]));
            const uint32_t w_site = wprev1;
            wprev1 = w0.w[1];
            if (!table) {
                // propose_step (oracle): Flip.propose_step (mcusher.py:154-170) / Swap
                const int sl = pick_sublattice(w0.w[0]);
                if (P.step_type == SMOLMC_STEP_FLIP) {
                    const uint32_t nact = (uint32_t)(P.sub_ptr[sl + 1] - P.sub_ptr[sl]);
                    const int site1 = site_of(sl, __umulhi(w_site, nact));
                    const int c0 = P.sub_code_ptr[sl];
                    const uint32_t nc = (uint32_t)(P.sub_code_ptr[sl + 1] - c0);
                    const uint32_t kk = __umulhi(univ_block(step, 1u, key0, key1).w[0], nc - 1u);
                    const int cur = uocc_ld<OL>(occ, site1);
                    int code = -1;
                    uint32_t seen = 0;
                    for (uint32_t c = 0; c < nc && code < 0; ++c) {
                        const int cc = P.sub_codes[c0 + c];
                        if (cc == cur) continue;
                        if (seen == kk) code = cc;
                        seen++;
                    }
                    if (lane == 0) { fl_site[0] = site1; fl_new[0] = code; }
                    nfl = 1;
                } else {
                    nfl = propose_swap(sl, w_site);
                }
            } else {
                // TableFlip.propose_step (mcusher.py:553-639): propose_table_flip (oracle)
                const philox_out w1 = univ_block(step, 1u, key0, key1);
                bool do_swap = (double)w0.w[0] * (1.0 / 4294967296.0) < U.tf_sw; // :577-578
                double sumw = 0.0;
                if (!do_swap) {
                    sumw = masked_weights(-1);
                    if (!(sumw > 0.0)) do_swap = true; // no feasible direction: canonical swap only (:604-611)
                }
                if (do_swap) {
                    nfl = propose_swap(pick_sublattice(w1.w[1]), w_site);
                } else {
                    // choose_section_from_partition (math.py:870-893)
                    const double target = (double)w1.w[0] * (1.0 / 4294967296.0) * sumw;
                    double cum = 0.0;
                    int idx = -1, last = -1;
                    for (int i = 0; i < 2 * U.tf_n && idx < 0; ++i) {
                        const double m = s_mw[i];
                        if (m <= 0.0) continue;
                        last = i;
                        cum += m;
                        if (target < cum) idx = i;
                    }
                    if (idx < 0) idx = last;
                    idx = uni(idx);
                    const int d = U.tf_d;
                    const int *row = U.tf_table + (size_t)(idx >> 1) * d;
                    const int sgn = (idx & 1) ? -1 : 1;
                    log_priori = log_priori_of(idx, sumw);
                    dir = idx;
                    uint32_t tcand = 0, qdraw = 0, wc_blk = 0xffffffffu, wd_blk = 0xffffffffu;
                    philox_out wc = w0, wd = w0;
                    for (int sl = 0; sl < nsub; ++sl) {
                        const uint32_t nact = (uint32_t)(P.sub_ptr[sl + 1] - P.sub_ptr[sl]);
                        const int base = P.sub_code_ptr[sl], nc = P.sub_code_ptr[sl + 1] - base;
                        int ncol = 0;
                        for (int c = 0; c < nc; ++c) { // depleted species: -u sites without replacement
                            const int u = sgn * row[base + c];
                            const int want = P.sub_codes[base + c];
                            for (int k = 0; k < -u; ++k) {
                                for (;;) {
                                    const uint32_t blk = 4u + tcand / 4u;
                                    if (blk != wc_blk) { wc = univ_block(step, blk, key0, key1); wc_blk = blk; }
                                    const uint32_t word = (tcand & 3u) == 0 ? wc.w[0] : (tcand & 3u) == 1 ? wc.w[1] : (tcand & 3u) == 2 ? wc.w[2] : wc.w[3];
                                    const int site = uni(site_of(sl, __umulhi(word, nact)));
                                    tcand++;
                                    if (uocc_ld<OL>(occ, site) != want) continue;
                                    int dup = 0;
                                    for (int z = 0; z < ncol; ++z) dup |= s_col[z] == site;
                                    if (dup) continue;
                                    if (lane == 0) s_col[ncol] = site;
                                    ncol++;
                                    break;
                                }
                            }
                        }
                        for (int c = 0; c < nc; ++c) { // enriched species: random assignment (:627-631)
                            const int u = sgn * row[base + c];
                            for (int k = 0; k < u; ++k) {
                                const uint32_t blk = 2u + qdraw / 4u;
                                if (blk != wd_blk) { wd = univ_block(step, blk, key0, key1); wd_blk = blk; }
                                const uint32_t word = (qdraw & 3u) == 0 ? wd.w[0] : (qdraw & 3u) == 1 ? wd.w[1] : (qdraw & 3u) == 2 ? wd.w[2] : wd.w[3];
                                const int rr = (int)__umulhi(word, (uint32_t)ncol);
                                qdraw++;
                                const int picked = s_col[rr];
                                if (lane == 0 && nfl < SMOLMC_MAX_STEP_FLIPS) {
                                    fl_site[nfl] = picked;
                                    fl_new[nfl] = P.sub_codes[base + c];
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
        }

        // ================= pass 1: enthalpy change, flips applied tentatively ==============
        double e = 0.0, ew_part = 0.0, ew_uni = 0.0, dMu = 0.0;
        for (int f = 0; f < nfl; ++f) {
            const int s = uni(fl_site[f]), newc = uni(fl_new[f]);
            const int oldc = uni(uocc_ld<OL>(occ, s));
            int orig = oldc;
            for (int g = f - 1; g >= 0; --g)
                if (fl_site[g] == s) orig = fl_orig[g];
            if (lane == 0) { fl_old[f] = oldc; fl_orig[f] = orig; }
            e += univ_flip_partial<OL>(U, occ, lane, s, newc, s_dF);
            if (has_ewald) {
                if (P.ew_field) {
                    // O(1) from the walker's potential field + the cross terms of the earlier flips
                    const KParams *Q = &P;
                    const int W = Q->ew_W, ab = Q->ew_act_base, na = Q->ew_nact;
                    const double *phi = Q->ew_phi + (size_t)r * na;
                    const double dq = Q->ew_qs[(size_t)s * W + newc] - Q->ew_qs[(size_t)s * W + oldc];
                    double pot = __hip_atomic_load(&phi[s - ab], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int g = 0; g < f; ++g) {
                        const int sg = fl_site[g];
                        if (sg == s) continue;
                        const double dqg = Q->ew_qs[(size_t)sg * W + fl_new[g]] - Q->ew_qs[(size_t)sg * W + fl_old[g]];
                        pot = fma(dqg, Q->ew_G[(size_t)s * na + (sg - ab)], pot);
                    }
                    ew_uni += 2.0 * dq * pot + (Q->ew_dg[(size_t)s * W + newc] - Q->ew_dg[(size_t)s * W + oldc]);
                } else if (P.ew_compact) {
                    const int W = P.ew_W;
                    const double dq = P.ew_qs[(size_t)s * W + newc] - P.ew_qs[(size_t)s * W + oldc];
                    ew_part += 2.0 * dq * univ_ewald_compact<OL>(P, occ, lane, s);
                    ew_uni += 2.0 * dq * P.ew_frozen[s] + (P.ew_dg[(size_t)s * W + newc] - P.ew_dg[(size_t)s * W + oldc]);
                } else {
                    ew_part += univ_ewald_dense<OL>(P, occ, lane, s, oldc, newc);
                }
            }
            if (has_mu) dMu += P.mu[(size_t)s * P.mu_W + newc] - P.mu[(size_t)s * P.mu_W + orig]; // ensemble.py:368-374
            uocc_st<OL>(occ, s, newc, lane); // tentative
        }
        double dH = wave_sum(e);
        double dEw = 0.0;
        if (has_ewald) {
            dEw = uni_d(wave_sum(ew_part) + ew_uni);
            dH += P.ew_coef * dEw;
        }
        if (has_mu) { dMu = uni_d(dMu); dH -= dMu; }
        // MCBias.compute_bias_change (kernel/base.py:307-311; orc_compute_bias_change): the last flip of a
        // site counts, against the species before the step
        double dB = 0.0, dq_row[SMOLMC_MAX_BIAS_ROWS] = {0.0, 0.0, 0.0, 0.0}, qrow_now[SMOLMC_MAX_BIAS_ROWS] = {0.0, 0.0, 0.0, 0.0};
        if (bias_type && nfl) {
            const int W = P.bias_W;
            if (bias_type == SMOLMC_BIAS_FUGACITY) {
                for (int f = 0; f < nfl; ++f) {
                    const int s = fl_site[f];
                    int last = 1;
                    for (int g = f + 1; g < nfl; ++g) last &= fl_site[g] != s;
                    if (!last) continue;
                    dB += log(P.bias_tab[(size_t)s * W + fl_new[f]] / P.bias_tab[(size_t)s * W + fl_orig[f]]);
                }
            } else {
                double sq_new = 0.0, sq_old = 0.0;
#pragma unroll
                for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k) { // (constant indices: dq_row stays in registers)
                    if (k >= P.bias_rows) continue;
                    const double *tab = P.bias_tab + (size_t)k * P.bias_row_stride;
                    const double c = __hip_atomic_load(&qrow[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    double cn = c;
                    for (int f = 0; f < nfl; ++f) {
                        const int s = fl_site[f];
                        int last = 1;
                        for (int g = f + 1; g < nfl; ++g) last &= fl_site[g] != s;
                        if (!last) continue;
                        cn += tab[(size_t)s * W + fl_new[f]] - tab[(size_t)s * W + fl_orig[f]];
                    }
                    dq_row[k] = cn - c;
                    qrow_now[k] = c;
                    sq_old += c * c;
                    sq_new += cn * cn;
                }
                dB = -P.bias_pen * sq_new - (-P.bias_pen * sq_old);
            }
            dB = uni_d(dB);
        }

        // ================= accept ========================================================
        bool accepted;
        if (!WL) {
            double exponent = -beta * dH + log_priori; // metropolis.py:41-42
            if (bias_type) exponent += dB;               // :43-44
            accepted = __ballot(exponent >= 0.0 ? true : (exponent > lu)) != 0ull;
        } else {
            const double new_h = H + dH; // wanglandau.py:188
            if (new_h < P.wl_min || new_h >= P.wl_max) {
                accepted = false;
            } else {
                const long long b = (long long)floordiv_exact(H - P.wl_min, P.wl_bin);
                const long long nb = (long long)floordiv_exact(new_h - P.wl_min, P.wl_bin);
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
                    const double v = s_dF[i];
                    if (v != 0.0) unsafeAtomicAdd(&feat[i], v);
                    s_dF[i] = 0.0;
                }
            } else {
                // pass 2: feature deltas flip by flip against the occupancy each flip saw
                for (int f = nfl - 1; f >= 0; --f) uocc_st<OL>(occ, fl_site[f], fl_old[f], lane);
                for (int f = 0; f < nfl; ++f) {
                    univ_flip_features<OL>(U, occ, lane, uni(fl_site[f]), uni(fl_new[f]), feat);
                    uocc_st<OL>(occ, fl_site[f], fl_new[f], lane);
                }
            }
            if (lane == 0) {
                if (has_ewald) unsafeAtomicAdd(&feat[Fce], dEw);
                if (has_mu) unsafeAtomicAdd(&feat[Fce + (has_ewald ? 1 : 0)], dMu);
            }
            if (has_ewald && P.ew_field) {
                const KParams *Q = &P;
                double *phi = Q->ew_phi + (size_t)r * Q->ew_nact;
                const int W = Q->ew_W;
                for (int f = 0; f < nfl; ++f) {
                    const int s = fl_site[f];
                    const double dq = Q->ew_qs[(size_t)s * W + fl_new[f]] - Q->ew_qs[(size_t)s * W + fl_old[f]];
                    if (dq != 0.0) field_apply_global(*Q, phi, lane, s, dq);
                }
            }
            if (dir >= 0 && lane < U.tf_d) {
                const int sgn = (dir & 1) ? -1 : 1;
                s_cnt[lane] += sgn * U.tf_table[(size_t)(dir >> 1) * U.tf_d + lane];
            }
            H += dH;
            bias += dB;
            if (bias_type && bias_type != SMOLMC_BIAS_FUGACITY && lane == 0) {
#pragma unroll
                for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k)
                    if (k < P.bias_rows) __hip_atomic_store(&qrow[k], qrow_now[k] + dq_row[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // lane 0's feature atomics / charge stores have reached L2 before the wave reads them there (the
            // readers load past the L1): a wait, no cache maintenance
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            nacc++;
        } else if (accepted) {
            nacc++; // the empty step is accepted (metropolis.py:46)
        } else {
            for (int f = nfl - 1; f >= 0; --f) uocc_st<OL>(occ, fl_site[f], fl_old[f], lane);
            if (s_dF != nullptr && nfl)
                for (int i = lane; i < Fce; i += 64) s_dF[i] = 0.0;
        }
        last_acc = accepted ? 1 : 0;

        if (WL) {
            // WangLandau._do_post_step (wanglandau.py:222-266)
            const double bq = floordiv_exact(H - P.wl_min, P.wl_bin);
            if (bq >= 0.0 && bq < (double)P.L) {
                const long long b = (long long)bq;
                wl_counter++;
                long long total = 0;
                if (lane == 0) total = wl_oc[b];
                total = ((long long)(unsigned)uni((int)(total >> 32)) << 32) | (unsigned)uni((int)(total & 0xffffffffll));
                const double inv = 1.0 / (double)(total + 1);
                for (int i = lane; i < F; i += 64) {
                    double *mf = wl_mf + (size_t)b * F + i;
                    const double cf = __hip_atomic_load(&feat[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *mf = inv * (cf + (double)total * (*mf));
                }
                if (wl_counter % P.wl_update == 0 && lane == 0) {
                    wl_S[b] += wl_m;
                    wl_Hh[b] += 1;
                    wl_oc[b] = total + 1;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); // lane 0's entropy update before the next step's reads
            }
            if (P.wl_check != 0 && wl_counter % P.wl_check == 0) { // (check period 0: no device-side check)
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                long cnt = 0;
                double sum = 0;
                for (int i = lane; i < P.L; i += 64) {
                    const double Si = __hip_atomic_load(&wl_S[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const long long Hi = __hip_atomic_load(&wl_Hh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (Si > 0) { cnt++; sum += (double)Hi; }
                }
                const double tcnt = wave_sum((double)cnt), tsum = wave_sum(sum);
                if (tcnt >= 2.0) {
                    const double thr = P.wl_flat * (tsum / tcnt);
                    int bad = 0;
                    for (int i = lane; i < P.L; i += 64) {
                        const double Si = __hip_atomic_load(&wl_S[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const long long Hi = __hip_atomic_load(&wl_Hh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (Si > 0 && !((double)Hi > thr)) bad = 1;
                    }
                    if (__ballot(bad) == 0ull) {
                        for (int i = lane; i < P.L; i += 64)
                            __hip_atomic_store(&wl_Hh[i], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        wl_m = wl_m / P.wl_div;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            }
        }
        if (replay && lane == 0) {
            const size_t k = (size_t)r * nsteps_run + it_step;
            if (P.rp_acc) P.rp_acc[k] = (uint8_t)last_acc;
            if (P.rp_H) P.rp_H[k] = H;
            if (U.rp_lp_out) U.rp_lp_out[k] = log_priori;
        }
        if (smp_every && --smp_countdown == 0) { // one thinned sample of this walker
            smp_countdown = smp_every;
            const size_t row = (size_t)smp_index * P.R + r;
            smp_index++;
            for (int i = lane; i < F; i += 64)
                P.smp.feat[row * F + i] = __hip_atomic_load(&feat[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) {
                P.smp.H[row] = H;
                P.smp.acc[row] = (uint8_t)last_acc;
            }
            if (P.smp.occ) {
                uint8_t *dst = P.smp.occ + row * P.Npad;
                for (int i = lane; i < P.Npad; i += 64) dst[i] = (uint8_t)uocc_ld<OL>(occ, i);
            }
        }
    }

    // ---- write the chain back -----------------------------------------------------------
    if (OL) {
        uint4 *dst = (uint4 *)(P.occ + (size_t)r * P.Npad);
        for (int i = lane; i < P.Npad / 16; i += 64) dst[i] = ((const uint4 *)occ)[i];
    }
    if (lane == 0) {
        P.enthalpy[r] = H;
        P.nsteps[r] = step;
        P.nacc[r] = nacc;
        P.last_acc[r] = (uint8_t)last_acc;
        if (bias_type) P.bias[r] = bias;
        if (WL) {
            P.wl_m[r] = wl_m;
            P.wl_counter[r] = wl_counter;
        }
    }
}

