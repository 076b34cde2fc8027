// mc_lean_multi.h -- lean Metropolis kernel for models with several site classes / several
// active sublattices (disordered cation AND anion sublattices, symmetry-distinct sites) and
// for up to 512 clusters per site.  Same design as mc_lean_kernel (one wavefront per walker,
// occupancy + delta tables in LDS, lane-packed index rows, float32 accept pre-test, Ewald
// potential field), with the per-class state moved from registers to LDS:
//   * slot records {table offset, strides, weight} per (class, slot, lane), read per flip;
//   * feature accumulators per (class, slot, lane) as LDS cells (ds_add_f64 on accept);
//   * per-sublattice site range / species count / mu / charge rows.
// Proposal stream identical to the oracle's: sublattice from W(step, 0, 0) and the cumulative
// sublattice probabilities (mcusher.py:146-148), site word W(step - 1, 0, 1), swap candidates
// inside the chosen sublattice.  The sites of a batch of 16 steps are formed at batch time, so
// the next step's row is prefetched except across a batch boundary (one exposed fetch per 16).
#pragma once
#include "mc_lean.h"

#ifdef SMOLMC_NO_TABLE_FAST // A/B switch: every table step through the full candidate scan
#define SMOLMC_TABLE_MULTI_FAST false
#else
#define SMOLMC_TABLE_MULTI_FAST true
#endif

// element k (< 4) of a kernel-argument array without dynamic indexing (which would go through scratch)
__device__ __forceinline__ int sel4(const int (&a)[4], int k) {
    return k == 0 ? a[0] : (k == 1 ? a[1] : (k == 2 ? a[2] : a[3]));
}

// (potential field kept in HBM, ew_field == 2 -- it would cost too much LDS: field_apply /
// field_apply2 of mc_lean.h are called with the global pointer; the update is then bound by HBM /
// Infinity-Cache traffic: the G rows plus one read-modify-write of phi)
struct MultiRec { // slot record, 24 bytes
    uint32_t doff8;
    uint32_t st8[3];
    double w;
};

// ---- Wang-Landau on the multi-class layout (WLK, round 5) -----------------------------------------
// kernel/wanglandau.py:186-266 for every model class this kernel takes (several site classes = active
// sublattices, up to 512 clusters per site, mu rows, the Ewald field in LDS or HBM) and for any
// update_period.  Per walker in LDS:
//   * S[L]      float64 entropies (the accept test reads S[bin], S[new bin]; += m per counted step);
//   * cnt[L]    uint32: counted steps per bin SINCE the launch started / the last successful flatness
//               check -- histogram and occurrences gain one together (wanglandau.py:241-245), so one
//               delta serves both: the HBM arrays keep the base values and take the deltas when the
//               histogram is reset and when the launch ends (8 + 4 bytes per bin instead of 24);
//   * occb[L]   float64 occurrences at launch start, ONLY when update_period > 1 (the running-mean
//               recurrence :235-239 then needs total = occurrences[bin] on every step);
//   * rows      update_period 1 (per-bin feature SUMS): a log of WL_ROWS finished runs {bin, run_n * features}
//               (a run = the post-steps spent in one state), written to the rows in HBM in bursts;
//               update_period > 1 (running means): a direct-mapped cache of WL_ROWS rows.
// The current feature vector lives in the lanes of one register (lane f < F), updated on accepted
// steps through the shadow cells of mc_wl.h; the enthalpy is carried as the reference carries it
// (_current_enthalpy += delta, :216-218) and every decision is the exact float64 one.
// (wl_multi_wave_bytes, the size of that state, is in smolmc_common.h: the host sizes the launch with it)
// a cached row changes its bin: the old bin's row goes to HBM (sums: added, the rows of a walker are
// touched by its own wave only; means: stored) and the new bin's comes in (means only; a row of sums
// starts from zero).  Out of line: global-memory instructions that exist on some paths of the step loop
// only make the compiler's s_waitcnt insertion conservative on every path (NOTES.md).
__device__ __noinline__ void wl_multi_row_swap(double *grows, double *crow, int old_bin, int new_bin, int F, int lane,
                                               int sum_mode) {
    if (lane < F) {
        if (old_bin >= 0) {
            if (sum_mode) unsafeAtomicAdd(grows + (size_t)old_bin * F + lane, crow[lane]);
            else grows[(size_t)old_bin * F + lane] = crow[lane];
        }
        crow[lane] = sum_mode ? 0.0 : grows[(size_t)new_bin * F + lane];
    }
}
// (wl_multi_flatness_check, the flatness check on the compact records, is in mc_lean.h: mc_wl_kernel shares it)
#define WLM_LOG SMOLMC_WLM_LOG // entries of the log of finished runs (sums)
// ONE: a single site class (the 257..512-clusters-per-site models): the slot records stay in
// registers for the whole launch instead of being re-read from LDS for every flip.
// BIAS: FugacityBias / SquareChargeBias (bias.py:96-287) with one bias row per sublattice,
// P.bias_pair[sub][old * 8 + new]; biased walkers always take the exact decision path (as in
// mc_lean_kernel).  Separate instantiations (multi_bias_n*.hip).
// EWM: 0 = no Ewald term, 1 = potential field in LDS, 2 = potential field in HBM.  A template
// parameter: with the placement as a runtime flag the field reads are memory instructions "on some
// paths only", which the compiler's s_waitcnt insertion cannot count (conservative waits at the
// index-row fetches, see mc_lean.h).
// REPLAY: proposals and uniforms from the host in the reference's draw order (see mc_lean_kernel;
// instantiated in multi_replay_n*.hip).
// WLK: the Wang-Landau kernel on this layout (see above; multi_wl_n*.hip).  WLK == 2 (KFW, multi_wl_kf_n*.hip): ... with
// several correlation functions per orbit (evaluator.pyx:211-265): the decision reads the slot's FOLDED table
// E = sum_k coef_k ct_k as every other slot does; an accepted step reads the slot's K function tables (global memory,
// LeanParams::dtk, at the table index the decision computed) and adds K feature deltas.
template <int NSLOT, int MM, int STEP, bool HAS_MU, int EWM, bool ONE = false, bool BIAS = false, bool REPLAY = false, int WLK = 0>
__global__ void __launch_bounds__(512) mc_lean_multi_kernel(const LeanParams P) {
    static_assert(!(WLK && BIAS), "Cannot apply bias to Wang-Landau simulation (wanglandau.py:127-128)");
    constexpr bool KFW = WLK == 2;
    constexpr bool HAS_EW = EWM != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int r = uni(blockIdx.x * nwaves + wave);
    const int NC = P.m_ncls, NS = P.m_nsub;
    // block-shared: delta tables | mu rows [4][8] | q rows [4][8] | dg rows [4][8] | slot records
    double *s_dt = (double *)smem;
    double *s_mu = s_dt + P.dt_len;
    double *s_q = s_mu + 32, *s_dg = s_mu + 64;
    MultiRec *s_rec = (MultiRec *)(s_mu + 96);
    const int nrec = NC * NSLOT * 64;
    // (WLK) feature scale and feature index of every slot record, read on accepted steps only
    double *s_fs = (double *)(s_rec + nrec);
    uint32_t *s_ft = (uint32_t *)(s_fs + (WLK ? nrec : 0));
    unsigned char *shared_end = (unsigned char *)(s_ft + (WLK ? ((nrec + 3) & ~3) : 0));
    // per wave: occupancy [Nlds] | zero pad 64 | feature scratch [64] | acc cells [NC][NSLOT][64] | phi
    // (WLK: S [L] | cnt [L] | occb [L] (update_period > 1) | rows [WL_ROWS][F] instead of the acc cells)
    const int wl_sum_mode = WLK ? P.wl.sum_mode : 1;
    const size_t state_bytes = WLK ? wl_multi_wave_bytes(P.wl.L, P.F, wl_sum_mode) : (size_t)nrec * 8;
    const size_t per_wave = (size_t)P.Nlds + 64 + 64 * 8 + state_bytes + ((EWM == 1) ? (size_t)P.ew_nact * 8 : 0);
    unsigned char *wbase = shared_end + (size_t)wave * per_wave;
    uint8_t *occ = wbase;
    double *s_feat = (double *)(wbase + P.Nlds + 64);
    double *s_acc = s_feat + 64;
    double *wl_S = s_acc;                                                   // WLK
    uint32_t *wl_cnt = (uint32_t *)(wl_S + (WLK ? P.wl.L : 0));
    double *wl_occb = (double *)((unsigned char *)wl_cnt + (WLK ? (((size_t)P.wl.L * 4 + 7) & ~(size_t)7) : 0));
    double *s_rows = wl_occb + ((WLK && !wl_sum_mode) ? P.wl.L : 0);
    // Ewald potential field: LDS copy (ew_field 1) or the walker's HBM array itself (ew_field 2)
    constexpr bool phi_lds = EWM == 1;
    double *phi = phi_lds ? (double *)((unsigned char *)s_acc + state_bytes) : P.ew_phi + (size_t)r * P.ew_nact;
    const int swa = P.swz_a, swm = P.swz_m, swb = P.swz_b;
    for (int i = threadIdx.x; i < P.dt_len; i += blockDim.x) s_dt[i] = P.dt[i];
    if (threadIdx.x < 32) {
        s_mu[threadIdx.x] = HAS_MU ? P.m_mu[threadIdx.x] : 0.0;
        s_q[threadIdx.x] = HAS_EW ? P.m_q[threadIdx.x] : 0.0;
        s_dg[threadIdx.x] = HAS_EW ? P.m_dg[threadIdx.x] : 0.0;
    }
    for (int i = threadIdx.x; i < nrec; i += blockDim.x) {
        const LeanSlot sl = P.slots[i];
        MultiRec rec;
        rec.doff8 = sl.doff8;
        rec.st8[0] = sl.stride8[0]; rec.st8[1] = sl.stride8[1]; rec.st8[2] = sl.stride8[2];
        rec.w = sl.w;
        s_rec[i] = rec;
        if (WLK) {
            s_fs[i] = sl.live ? sl.fs : 0.0;
            s_ft[i] = KFW ? (sl.feat | (sl.live << 16)) : sl.feat; // (KFW: + the number of functions of the slot's orbit)
        }
    }
    const bool live = r < P.R;
    if (live) {
        const uint32_t *src = (const uint32_t *)(P.occ + (size_t)r * P.Npad);
        for (int i = lane; i < P.Npad / 4; i += 64)
            *(uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb)) = src[i];
        s_feat[lane] = 0.0;
        if (!WLK)
            for (int i = lane; i < nrec; i += 64) s_acc[i] = 0.0;
        if (WLK) {
            for (int i = lane; i < P.wl.L; i += 64) {
                wl_S[i] = P.wl.entropy[(size_t)r * P.wl.L + i];
                wl_cnt[i] = 0u;
                if (!wl_sum_mode) wl_occb[i] = (double)P.wl.occur[(size_t)r * P.wl.L + i];
            }
            if (!wl_sum_mode)
                for (int i = lane; i < SMOLMC_WL_ROWS * P.F; i += 64) s_rows[i] = 0.0;
        }
        if (phi_lds)
            for (int j = lane; j < P.ew_nact; j += 64) phi[j] = P.ew_phi[(size_t)r * P.ew_nact + j];
    }
    __syncthreads();
    if (!live) return;

    double H = P.enthalpy[r];
    const double nbeta = -P.beta[r];
    unsigned long long step = P.nsteps[r];
    uint32_t nacc_add = 0, nacc_before = 0;
    const uint32_t key0 = (uint32_t)P.seeds[r], key1 = (uint32_t)(P.seeds[r] >> 32);
    const uint32_t nt8 = P.nt8, snt8 = P.snt8;
    const int abase = P.ew_act_base;
    double acc_mu = 0.0, acc_ew = 0.0;
    constexpr bool FAST = !HAS_EW && !BIAS && !WLK; // float32 accept pre-test (Ewald / biased / Wang-Landau variants take the exact path)
    const int btype = BIAS ? P.bias_type : 0;
    // (running sums of the quadratic biases, one per row: see mc_lean_kernel)
    const int brows = (btype && btype != SMOLMC_BIAS_FUGACITY) ? P.bias_rows : 0;
    double bias_acc = 0.0, chg[SMOLMC_MAX_BIAS_ROWS];
#pragma unroll
    for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k) chg[k] = k < brows ? P.charge[(size_t)r * SMOLMC_MAX_BIAS_ROWS + k] : 0.0;
    float thr_lo = 0.0f, thr_hi = 0.0f;
    double *featp = P.features + (size_t)r * P.F;
    const double base_feat = lane < P.F ? featp[lane] : 0.0;
    // ---- Wang-Landau state (WLK) ----
    double fcur = base_feat;                  // lane f < F: the walker's current feature vector (_current_features)
    double Hcur = H;                          // _current_enthalpy (wanglandau.py:216-218)
    const double wl_vmin = WLK ? P.wl.vmin : 0.0, wl_bin = WLK ? P.wl.bin : 1.0, wl_inv_bin = 1.0 / wl_bin;
    double wl_m = WLK ? P.wl.m[r] : 0.0;
    int wb = 0;                               // current bin (walkers start inside the window: smolmc_set_state)
    if (WLK) wb = min(max(uni((int)floordiv_exact(Hcur - wl_vmin, wl_bin)), 0), P.wl.L - 1);
    const long long wl_counter0 = WLK ? P.wl.counter[r] : 0;
    const uint32_t wl_check = WLK ? (uint32_t)P.wl.check : 0u, wl_upd = WLK ? (uint32_t)P.wl.update : 1u;
    // counters modulo the periods (the host refuses periods >= 2^31; check period 0 = no device-side check:
    // the remainder starts at 1 and cannot wrap to 0 inside a launch of < 2^30 steps)
    uint32_t wl_rem_check = (WLK && wl_check) ? (uint32_t)uni((int)(wl_counter0 % (long long)wl_check)) : 1u;
    uint32_t wl_rem_upd = WLK ? (uint32_t)uni((int)(wl_counter0 % (long long)wl_upd)) : 0u;
    uint32_t wl_run_n = 0;                    // sums: post-steps of the current (bin, features) state not yet in its row
    int vtag = -1;                            // lane i < WL_ROWS: the bin cached in row i (-1: none)
    const int f_ew = P.Fce, f_mu = P.Fce + (HAS_EW ? 1 : 0);
    // shadow copies of the feature cells for the accepted steps' deltas (see mc_wl.h): lane l adds into
    // copy l % wl_k, a reader sums the copies; cell 63 is never written (the address of "no copy")
    const int wl_k = max(1, min(8, 63 / max(P.F, 1)));
    const uint32_t wl_shadow = (uint32_t)((lane % wl_k) * P.F);
    const int wl_rd0 = lane < P.F ? lane : 63, wl_rdstep = lane < P.F ? P.F : 0; // copy k of feature `lane`: cell wl_rd0 + k wl_rdstep
    const bool wl_zero_lane = lane < wl_k * P.F;
    // the cached row of a bin (evicting what the slot held)
    auto wl_row_of = [&](const int bin) -> double * {
        const int slot = bin & (SMOLMC_WL_ROWS - 1);
        const int tag = (int)rdlane((uint32_t)vtag, slot);
        double *crow = s_rows + (uint32_t)slot * (uint32_t)P.F;
        if (tag != bin) {
            const LeanParamsKernarg Q = rare_params();
            wl_multi_row_swap(Q->wl.meanf + (size_t)r * Q->wl.L * Q->F, crow, tag, bin, Q->F, lane, wl_sum_mode);
            vtag = lane == slot ? bin : vtag;
        }
        return crow;
    };
    // Sums (update_period 1): the post-steps spent in the current state (bin, features) add run_n * features to
    // the bin's row when the state ends.  The rows are NOT cached by bin: with an Ewald term a swap moves the
    // enthalpy by several bins' worth, the walker revisits a bin after hundreds of others, and every miss of a
    // direct-mapped cache was a global atomic in the step loop -- vmcnt counts in order, so the next index-row
    // wait sat on its acknowledgement (~1000 cycles per accepted step, measured).  Finished runs are LOGGED in
    // LDS instead, WL_ROWS entries {bin, run_n * features}, and the log goes to the rows in HBM in one burst of
    // fire-and-forget atomics (the rows of a walker are touched by its own wave only).
    int wl_nlog = 0;
    // (64 / F entries per atomic instruction: lane l carries feature l % F of entry base + l / F)
    const int wl_epi = max(1, 64 / max(P.F, 1)), wl_lane_e = lane / max(P.F, 1), wl_lane_f = lane - wl_lane_e * P.F;
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
            if (lane < P.F) s_rows[(uint32_t)wl_nlog * (uint32_t)P.F + lane] = (double)wl_run_n * fcur;
            vtag = lane == wl_nlog ? wb : vtag;
            wl_run_n = 0u;
            if (++wl_nlog == WLM_LOG) wl_log_flush();
        }
    };
    uint32_t smp_countdown = P.smp.every ? (uint32_t)P.smp.every : 0xffffffffu; // (off: cannot reach zero in a launch)
    long long smp_index = 0;
    constexpr int ROW = NSLOT * MM;
    constexpr int NW = ROW / 2;
    constexpr uint32_t SITE_BYTES = 64u * ROW * 2u;
    const __amdgpu_buffer_rsrc_t idx_rs =
        __builtin_amdgcn_make_buffer_rsrc((void *)P.idx, 0, 0x7fffffff, 0x00020000);
    const uint32_t lane_voff = (uint32_t)lane * (ROW * 2u);

    // per-step values of the current batch, lane-indexed (lane 4 k holds step base + k)
    uint32_t W0 = 0, W1 = 0;
    double logu = 0.0;
    int vsite = 0, vaddr = 0, vsub = 0;
    int cand[4] = {0, 0, 0, 0}, canda[4] = {0, 0, 0, 0};
    double vGc[4] = {0.0, 0.0, 0.0, 0.0}; // swap + Ewald: cross terms G[candidate j of the lane][site of the lane's step]
    unsigned long long batch_base = ~0ull;
    RowWords<NW> row1;

    // Pending field updates (field in HBM + translation-compressed site kernel): an accepted flip is
    // not swept into phi at once but kept in a list of up to four (table offset S8[s], charge
    // change); a proposal at site j adds sum_p dq_p G[s_p][j] -- one gather from the compressed
    // tables by one lane per entry -- to the stored phi[j], and a full list goes into phi in ONE
    // pass (field_sweep_gx_multi<4>): phi, the traffic that bounds these kernels on large cells, is
    // read and written once per four accepted flips instead of once per flip / swap.  G[s][s] = 0
    // in the tables, so an entry never corrects its own site.  (Replay keeps the immediate update.)
    constexpr bool PEND = EWM == 2 && !REPLAY;
    const unsigned char *gxp = PEND ? (const unsigned char *)P.ew_gx : nullptr;
#ifdef SMOLMC_NO_EWALD_PENDING // A/B switch
    const bool pend_on = false;
#else
    // (worth it from about 32 groups of 64 field entries on: LiNiO2 8^3, 16 groups, loses 15 %)
    const bool pend_on = PEND && gxp != nullptr && P.ew_nact >= 2048;
#endif
    int npend = 0;
    uint32_t vps8 = 0;  // lane l: table offset of entry l & 3
    double vpdq = 0.0;  //         its charge change
    uint32_t vE8s = 0, vE8c[4] = {0, 0, 0, 0}; // per batch: E8 of the lane's site / swap candidates
    auto flush_pending = [&]() {
        const LeanParamsKernarg Q = rare_params();
        uint32_t s8[4];
        double dq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s8[k] = rdlane(vps8, k < npend ? k : 0);
            const double d = __hiloint2double((int)rdlane((uint32_t)__double2hiint(vpdq), k),
                                              (int)rdlane((uint32_t)__double2loint(vpdq), k));
            dq[k] = k < npend ? d : 0.0;
        }
        // (flip kernels: small batches inside their 128-register budget, four waves per SIMD; the swap
        // kernels run two waves per SIMD anyway and take batches of nine groups)
        field_sweep_gx_multi<4, STEP == SMOLMC_STEP_SWAP ? 0 : -1>(P.ew_phi + (size_t)r * Q->ew_nact, Q->ew_E8, gxp, lane, Q->ew_nact, s8, dq);
        npend = 0;
    };
    auto push_pending = [&](const int s, const double dq) {
        const LeanParamsKernarg Q = rare_params();
        const uint32_t s8 = Q->ew_S8[s - Q->ew_act_base];
        if ((lane & 3) == npend) { vps8 = s8; vpdq = dq; }
        npend++;
    };

    // Wang-Landau with the Ewald field in LDS on a field of exactly 16 groups (the reference's LiNiO2 model in an 8^3
    // cell, config 11): three steps of four are accepted there and every accepted step swept the field in three
    // dependent round trips -- S8 / E8, then two batches of gathers -- on a wave that has its SIMD to itself.  The E8
    // offsets of a lane's 16 entries are constants of the launch: kept in registers (round 6; the kernel runs one or two
    // waves per SIMD, registers are not what it is short of), the S8 offsets of a step's sites come with the proposal
    // batch, and ALL gathers of the sweep are issued at once: one round trip.
#ifdef SMOLMC_NO_WL_E16 // A/B switch (tools/build_variant.sh)
    constexpr bool WLE16 = false;
#else
    constexpr bool WLE16 = WLK != 0 && EWM == 1 && !REPLAY; // (EWM 1: the field in LDS)
#endif
    uint32_t e0w[16]; // (dead -- and removed by the compiler -- in the instantiations that do not take this sweep)
    bool have_e0w = false;
    uint32_t vS8s = 0, vS8c[4] = {0, 0, 0, 0}; // per batch: S8 of the lane's site / swap candidates
    if (WLE16 && phi_lds) {
        const LeanParamsKernarg Q = rare_params();
        if (Q->ew_gx != nullptr && Q->ew_nact == 1024) {
            const uint32_t *pE8 = Q->ew_E8;
#pragma unroll
            for (int u = 0; u < 16; ++u) e0w[u] = pE8[64 * u + lane];
            have_e0w = true;
        }
    }
    MultiRec rcs[ONE ? NSLOT : 1];
    if (ONE) {
#pragma unroll
        for (int it = 0; it < NSLOT; ++it) rcs[it] = s_rec[it * 64 + lane];
    }
    auto sub_of = [&](uint32_t w0) -> int { // MCUsher.get_random_sublattice (mcusher.py:146-148)
        if (NS == 1) return 0;
        const double x = (double)w0 * (1.0 / 4294967296.0);
        int sl = NS - 1; // (constant indices: a dynamic index into kernel arguments goes through scratch)
        if (NS > 3 && x < P.m_cum[2]) sl = 2;
        if (NS > 2 && x < P.m_cum[1]) sl = 1;
        if (x < P.m_cum[0]) sl = 0;
        return sl;
    };

    const uint32_t nsteps32 = (uint32_t)P.steps;
    // replay: record i of this walker = (site1, code1, site2, code2), -1 = no flip (an empty step
    // "flips" the first site of the first sublattice to its own species)
    double lu_rp = 0.0;
    int rp_bad = 0;
    auto rp_site = [&](const uint32_t i) -> int {
        const int v = uni(P.rp_steps[((size_t)r * nsteps32 + i) * 4]);
        return v >= 0 ? v : P.m_sbase[0];
    };
    if (REPLAY && nsteps32) row1 = load_row<NW>(idx_rs, lane_voff, (uint32_t)rp_site(0u) * SITE_BYTES);
#ifdef SMOLMC_EXP_PHASES // experiment: shader cycles per phase of a step (walker 0 prints the averages)
    long long mph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long mph_t = clock64();
#define MULTI_PHASE(i) { const long long tn = clock64(); mph[i] += tn - mph_t; mph_t = tn; }
#else
#define MULTI_PHASE(i)
#endif
    for (uint32_t it_step = 0; it_step < nsteps32; ++it_step, ++step) {
        const unsigned long long base = step & ~15ull;
        if (!REPLAY && base != batch_base) {
            // site word of the batch's first step: W(base - 1, 0, 1)
            uint32_t carry;
            if (batch_base == base - 16) {
                carry = rdlane(W1, 60);
            } else {
                const unsigned long long sp = base - 1ull;
                carry = (uint32_t)uni((int)philox4x32_10((uint32_t)sp, (uint32_t)(sp >> 32), 0u, 0u, key0, key1).w[1]);
            }
            batch_base = base;
            const unsigned long long st = base + (unsigned)(lane >> 2);
            const philox_out o = philox4x32_10((uint32_t)st, (uint32_t)(st >> 32), (uint32_t)(lane & 3), 0u,
                                               key0, key1);
            W0 = o.w[0]; W1 = o.w[1];
            logu = log(philox_u53(o.w[2], o.w[3]));
            if (FAST) {
                const double thr = logu / nbeta;
                const double eps = P.fast_eps + 1e-6 * fabs(thr);
                thr_lo = P.fast_eps > 0.0 ? (float)(thr - eps) : -INFINITY;
                thr_hi = P.fast_eps > 0.0 ? (float)(thr + eps) : INFINITY;
            }
            // sublattice / site / candidates of the lane's step
            const uint32_t w0_blk0 = (uint32_t)__shfl((int)W0, lane & ~3);
            const uint32_t w1_prev = (uint32_t)__shfl((int)W1, (lane & ~3) - 4);
            const uint32_t w_site = lane < 4 ? carry : w1_prev;
            vsub = sub_of(w0_blk0);
            const int sb = sel4(P.m_sbase, vsub);
            const uint32_t na = (uint32_t)sel4(P.m_nact, vsub);
            vsite = sb + (int)__umulhi(w_site, na);
            vaddr = lean_swz(vsite, swa, swm, swb);
            if (STEP == SMOLMC_STEP_SWAP) {
                cand[0] = sb + (int)__umulhi(o.w[0], na);
                cand[1] = sb + (int)__umulhi(o.w[1], na);
                cand[2] = sb + (int)__umulhi(o.w[2], na);
                cand[3] = sb + (int)__umulhi(o.w[3], na);
#pragma unroll
                for (int j = 0; j < 4; ++j) canda[j] = lean_swz(cand[j], swa, swm, swb);
                if (HAS_EW) {
                    // Cross terms of the lane's FOUR candidates with its step's site, a batch ahead.  From the
                    // translation-compressed tables when the model has them (entry (s, j) at E8[j] + S8[s]: 108 KB,
                    // L2-resident); else from the rows of the site kernel.  (Until round 5 only the first-round
                    // candidate's term came with the batch; a later candidate -- 4 steps of 10 -- read G[s2][s1]
                    // from a 16 MB matrix in the step: an HBM round trip in front of every such decision.)
                    const LeanParamsKernarg Q = rare_params();
                    const unsigned char *gxq = (const unsigned char *)Q->ew_gx;
                    if (gxq != nullptr) {
                        const uint32_t *qS8 = Q->ew_S8;
                        const uint32_t e1 = Q->ew_E8[vsite - abase];
                        uint32_t s8c[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) s8c[j] = qS8[cand[j] - abase];
#pragma unroll
                        for (int j = 0; j < 4; ++j) vGc[j] = *(const double *)(gxq + (size_t)(e1 + s8c[j]));
                        if (WLE16) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) vS8c[j] = s8c[j];
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) vGc[j] = P.ew_G[(size_t)cand[j] * P.ew_nact + (vsite - abase)];
                    }
                }
            }
            if (WLE16 && have_e0w) vS8s = rare_params()->ew_S8[vsite - abase];
            if (pend_on) {
                const uint32_t *pE8 = rare_params()->ew_E8;
                vE8s = pE8[vsite - abase];
                if (STEP == SMOLMC_STEP_SWAP) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) vE8c[j] = pE8[cand[j] - abase];
                }
            }
            // index row of the batch's first step (the other steps' rows are prefetched one step
            // ahead, see below).  Every step issues the same loads in the same order whatever
            // path it takes: a load that exists on some paths only makes the compiler's s_waitcnt
            // insertion wait with vmcnt(0) -- for the prefetch and the partner's row -- before
            // the first gathers (mc_lean.h, NOTES.md).
            row1 = load_row<NW>(idx_rs, lane_voff, rdlane((uint32_t)vsite, (int)(step & 15ull) * 4) * SITE_BYTES);
        }
        const int l4 = (int)(step & 15ull) * 4;
#ifndef SMOLMC_NO_SETPRIO
        if (!ONE) __builtin_amdgcn_s_setprio(1); // wave priority rises through the step (see mc_lean_kernel; no gain for ONE)
#endif
        int s1, a1, sub1;
        int rq1 = 0, rq2 = -1, rq3 = 0;
        bool rp_empty = false;
        if (REPLAY) {
            const int *rec = P.rp_steps + ((size_t)r * nsteps32 + it_step) * 4;
            const int q0 = uni(rec[0]);
            rq1 = uni(rec[1]); rq2 = uni(rec[2]); rq3 = uni(rec[3]);
            rp_empty = q0 < 0;
            s1 = rp_empty ? P.m_sbase[0] : q0;
            a1 = lean_swz(s1, swa, swm, swb);
            sub1 = 0; // the active sublattice that holds the site
            if (NS > 1 && s1 >= P.m_sbase[1] && s1 < P.m_sbase[1] + P.m_nact[1]) sub1 = 1;
            if (NS > 2 && s1 >= P.m_sbase[2] && s1 < P.m_sbase[2] + P.m_nact[2]) sub1 = 2;
            if (NS > 3 && s1 >= P.m_sbase[3] && s1 < P.m_sbase[3] + P.m_nact[3]) sub1 = 3;
            rp_bad |= (s1 < sel4(P.m_sbase, sub1) || s1 >= sel4(P.m_sbase, sub1) + sel4(P.m_nact, sub1)) ? 1 : 0;
            double u = uni_d(P.rp_u[(size_t)r * nsteps32 + it_step]);
            if (u != u) u = 0.0; // NaN: the reference accepted without drawing a number
            lu_rp = log(u);
            if (FAST) {
                const double thr = lu_rp / nbeta, eps = P.fast_eps + 1e-6 * fabs(thr);
                thr_lo = P.fast_eps > 0.0 ? (float)(thr - eps) : -INFINITY;
                thr_hi = P.fast_eps > 0.0 ? (float)(thr + eps) : INFINITY;
            }
        } else {
            s1 = (int)rdlane((uint32_t)vsite, l4);
            a1 = (int)rdlane((uint32_t)vaddr, l4);
            sub1 = (int)rdlane((uint32_t)vsub, l4);
        }
        const int cls1 = sel4(P.m_cls, sub1);
        // prefetch the next step's row while this one runs (not across a batch boundary).  ONE:
        // issued after the gathers of flip 1, straight into row1 (no second register set, no
        // copy per step); with several classes the earlier issue is worth more than the copy.
        // (last step of a batch: the next site is not known yet -- its sublattice comes from the
        // next batch's words --, the own row is fetched once more instead and dropped)
        const uint32_t nsite_pf = REPLAY ? (uint32_t)(it_step + 1u < nsteps32 ? rp_site(it_step + 1u) : s1)
                                         : rdlane((uint32_t)vsite, l4 < 60 ? l4 + 4 : l4);
        RowWords<NW> rown = row1;
        if (!ONE) rown = load_row<NW>(idx_rs, lane_voff, nsite_pf * SITE_BYTES);

        const int o1 = uni((int)occ[a1]);
        int nfl, s2, a2, n1, n2 = 0, o2 = 0; // (swap: s2 / a2 / o2 are set by every proposal outcome)
        double cross_b = 0.0;  // cross term G[s2][s1] of a partner that came from the batch's candidates
        bool have_cross = false;
        uint32_t e8_2 = 0xffffffffu; // E8 of the swap partner when it came from the batch's candidates
        uint32_t s8_2 = 0xffffffffu; // ... and its S8 (WLE16)
        if (STEP != SMOLMC_STEP_SWAP) { s2 = s1; a2 = a1; }
        if (REPLAY) { // the recorded proposal
            if (STEP == SMOLMC_STEP_FLIP) {
                nfl = rp_empty ? 0 : 1;
                n1 = rp_empty ? o1 : rq1;
                rp_bad |= (rq2 >= 0) ? 1 : 0;
            } else if (!rp_empty && rq2 >= 0) {
                nfl = 2;
                s2 = rq2;
                a2 = lean_swz(s2, swa, swm, swb);
                o2 = uni((int)occ[a2]);
                n1 = rq1;
                n2 = rq3;
                // both sites of a swap lie on one sublattice (they share the slot records)
                rp_bad |= (s2 < sel4(P.m_sbase, sub1) || s2 >= sel4(P.m_sbase, sub1) + sel4(P.m_nact, sub1)) ? 1 : 0;
            } else {
                nfl = 0; s2 = s1; a2 = a1; o2 = o1; n2 = o1; n1 = o1;
                rp_bad |= (!rp_empty || rq2 >= 0) ? 1 : 0;
            }
        } else if (STEP == SMOLMC_STEP_FLIP) {
            const uint32_t kk = __umulhi(rdlane(W0, l4 + 1), (uint32_t)(sel4(P.m_ncodes, sub1) - 1));
            n1 = (int)kk + ((int)kk >= o1 ? 1 : 0);
            nfl = 1;
        } else {
            // (nested if / else: one conditional + one unconditional branch on the common
            // first-candidate hit; see mc_lean_kernel)
            nfl = 2;
            n2 = o1;
#define SMOLMC_CAND_MASK(J)                                                                        \
    const int v##J = (int)occ[canda[J]];                                                           \
    const unsigned long long m##J = __ballot(v##J != o1) & (0xEull << l4);
#define SMOLMC_CAND_TAKE(J)                                                                        \
    {                                                                                              \
        const int b = __ffsll((long long)m##J) - 1;                                                \
        s2 = (int)rdlane((uint32_t)cand[J], b);                                                    \
        a2 = (int)rdlane((uint32_t)canda[J], b);                                                   \
        o2 = (int)rdlane((uint32_t)v##J, b);                                                       \
        if (PEND) e8_2 = rdlane(vE8c[J], b);                                                       \
        if (WLE16) s8_2 = rdlane(vS8c[J], b);                                                      \
        if (HAS_EW) {                                                                              \
            cross_b = __hiloint2double((int)rdlane((uint32_t)__double2hiint(vGc[J]), b),           \
                                       (int)rdlane((uint32_t)__double2loint(vGc[J]), b));          \
            have_cross = true;                                                                     \
        }                                                                                          \
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
                            const int sb = sel4(P.m_sbase, sub1);
                            const uint32_t na = (uint32_t)sel4(P.m_nact, sub1);
                            for (uint32_t q = 0;; ++q) {
                                const philox_out o = philox4x32_10((uint32_t)step, (uint32_t)(step >> 32),
                                                                   4u + 64u * q + (uint32_t)lane, 0u, key0, key1);
                                int selsite = -1, selv = 0;
#pragma unroll
                                for (int j = 3; j >= 0; --j) {
                                    const int cs = sb + (int)__umulhi(o.w[j], na);
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
                                    for (uint32_t a = lane; a < na; a += 64)
                                        any |= ((int)occ[lean_swz(sb + (int)a, swa, swm, swb)] != o1);
                                    if (__ballot(any) == 0ull) break;
                                }
                            }
                            if (!hit) { nfl = 0; s2 = s1; a2 = a1; o2 = o1; }
                        }
                    }
                }
            }
#undef SMOLMC_CAND_MASK
#undef SMOLMC_CAND_TAKE
            n1 = o2;
        }
#ifndef SMOLMC_NO_SETPRIO
        if (!ONE) __builtin_amdgcn_s_setprio(2);
#endif
        MULTI_PHASE(0)
        // potential at the two sites (HBM copy of the field: global loads, issued AHEAD of the
        // partner's row so that the wait for that row does not cover them as well)
        double p1 = 0.0, p2 = 0.0;
        if (HAS_EW) {
            p1 = phi_lds ? phi[s1 - abase]
                         : __hip_atomic_load(&phi[s1 - abase], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (STEP == SMOLMC_STEP_SWAP)
                p2 = phi_lds ? phi[s2 - abase]
                             : __hip_atomic_load(&phi[s2 - abase], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        double corr1 = 0.0, corr2 = 0.0; // what the pending flips add to the stored potentials
        if (pend_on) {
            const uint32_t e1 = rdlane(vE8s, l4);
            uint32_t e2 = e1;
            if (STEP == SMOLMC_STEP_SWAP) e2 = e8_2 != 0xffffffffu ? e8_2 : rare_params()->ew_E8[s2 - abase];
            const bool act = lane < 8 && (lane & 3) < npend;
            const uint32_t off = act ? ((lane & 4) ? e2 : e1) + vps8 : 0u;
            const double g = *(const double *)(gxp + off);
            double t = act ? vpdq * g : 0.0;
            t = dpp_xor_add<0xB1>(t); // sums over the quads 0-3 (site 1) and 4-7 (site 2)
            t = dpp_xor_add<0x4E>(t);
            corr1 = __hiloint2double((int)rdlane((uint32_t)__double2hiint(t), 0), (int)rdlane((uint32_t)__double2loint(t), 0));
            corr2 = __hiloint2double((int)rdlane((uint32_t)__double2hiint(t), 4), (int)rdlane((uint32_t)__double2loint(t), 4));
        }
        RowWords<NW> row2 = row1;
        if (STEP == SMOLMC_STEP_SWAP) row2 = load_row<NW>(idx_rs, lane_voff, (uint32_t)s2 * SITE_BYTES);

        // -------- enthalpy delta: slot records of the site's class from LDS ------------------
        const MultiRec *rec1 = s_rec + ((size_t)cls1 * NSLOT) * 64 + lane;
        double e = 0.0, d1[NSLOT], d2[NSLOT];
        uint32_t ad1[KFW ? NSLOT : 1], ad2[KFW ? NSLOT : 1]; // KFW: byte offsets of the two decision reads
        {
            const uint32_t pair1 = (uint32_t)o1 * snt8 + (uint32_t)n1 * nt8;
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) {
                const MultiRec rc = ONE ? rcs[ONE ? it : 0] : rec1[it * 64];
                uint32_t a = rc.doff8;
#pragma unroll
                for (int m = 0; m < MM; ++m) a += __umul24(rc.st8[m], (uint32_t)occ[row_entry<NW>(row1, it * MM + m)]);
                d1[it] = *(const double *)((const unsigned char *)s_dt + (a + pair1));
                if (KFW) ad1[KFW ? it : 0] = a + pair1;
                e = fma(rc.w, d1[it], e);
            }
        }
        if (ONE) // (see above: row1's last use, the gathers of flip 1, has been issued)
            row1 = load_row<NW>(idx_rs, lane_voff, nsite_pf * SITE_BYTES);
        double ew_uni = 0.0, dq1 = 0.0, dq2 = 0.0;
        if (HAS_EW) {
            dq1 = s_q[sub1 * 8 + n1] - s_q[sub1 * 8 + o1];
            ew_uni = 2.0 * dq1 * (p1 + corr1) + (s_dg[sub1 * 8 + n1] - s_dg[sub1 * 8 + o1]);
        }
        if (STEP == SMOLMC_STEP_SWAP) {
            occ[a1] = (uint8_t)n1; // tentative: the second flip sees the first (expansion.py:217-229)
            const uint32_t pair2 = (uint32_t)o2 * snt8 + (uint32_t)n2 * nt8;
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) {
                const MultiRec rc = ONE ? rcs[ONE ? it : 0] : rec1[it * 64]; // both sites of a swap share the sublattice / class
                uint32_t a = rc.doff8;
#pragma unroll
                for (int m = 0; m < MM; ++m) a += __umul24(rc.st8[m], (uint32_t)occ[row_entry<NW>(row2, it * MM + m)]);
                d2[it] = *(const double *)((const unsigned char *)s_dt + (a + pair2));
                if (KFW) ad2[KFW ? it : 0] = a + pair2;
                e = fma(rc.w, d2[it], e);
            }
            if (HAS_EW) {
                dq2 = s_q[sub1 * 8 + n2] - s_q[sub1 * 8 + o2];
                const double cross = have_cross ? cross_b : P.ew_G[(size_t)s2 * P.ew_nact + (s1 - abase)];
                ew_uni += 2.0 * dq2 * ((p2 + corr2) + dq1 * cross) + (s_dg[sub1 * 8 + n2] - s_dg[sub1 * 8 + o2]);
            }
        }
        double dMu = 0.0;
        if (HAS_MU && nfl >= 1) {
            dMu = s_mu[sub1 * 8 + n1] - s_mu[sub1 * 8 + o1];
            if (nfl == 2) dMu += s_mu[sub1 * 8 + n2] - s_mu[sub1 * 8 + o2];
        }
#ifndef SMOLMC_NO_SETPRIO
        if (!ONE) __builtin_amdgcn_s_setprio(3);
#endif
        // compute_bias_change against the original occupancy (kernel/base.py:307-311; bias.py)
        double dB = 0.0, dQ[SMOLMC_MAX_BIAS_ROWS] = {0.0, 0.0, 0.0, 0.0};
        if (BIAS && nfl >= 1) {
            if (btype == SMOLMC_BIAS_FUGACITY) {
                const double *bp = P.bias_pair + sub1 * 64;
                dB = bp[o1 * 8 + n1];
                if (nfl == 2) dB += bp[o2 * 8 + n2];
            } else {
                double sq_new = 0.0, sq_old = 0.0;
#pragma unroll
                for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k)
                    if (k < brows) {
                        const double *bp = P.bias_pair + k * P.bias_row_stride + sub1 * 64;
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
        double dH = 0.0, dEw = HAS_EW ? ew_uni : 0.0;
        bool accepted = false, decided = false;
        int wnb = wb;
        MULTI_PHASE(1)
        if (WLK) { // WangLandau._accept_step (wanglandau.py:186-202): exact float64 delta, exact floor division
            dH = wave_sum_all(e);
            if (HAS_EW) dH += P.ew_coef * dEw;
            if (HAS_MU) dH -= dMu;
            const double lu = REPLAY ? lu_rp
                                     : __hiloint2double((int)rdlane((uint32_t)__double2hiint(logu), l4),
                                                        (int)rdlane((uint32_t)__double2loint(logu), l4));
            const double new_h = Hcur + dH;
            if (__ballot(!(new_h < wl_vmin || new_h >= P.wl.vmax)) != 0ull) {
                wnb = uni((int)floordiv_exact_inv(new_h - wl_vmin, wl_bin, wl_inv_bin));
                const double ex = wl_S[wb] - wl_S[wnb] + 0.0; // (+ log a-priori factor: 0 for flips / swaps)
                accepted = __ballot((ex >= 0.0) || (ex > lu)) != 0ull;
            }
            decided = true;
        }
        if (FAST) { // float32 pre-test (see mc_lean_kernel)
            const float ef = (float)((HAS_MU && lane == 0) ? e - dMu : e);
            const float S = wave_sum_f32_uniform(ef);
            const unsigned long long bit = 1ull << (REPLAY ? 0 : l4); // (replay: the thresholds are uniform)
            const bool ca = (__ballot(S < thr_lo) & bit) != 0ull;
            const bool cr = (__ballot(S > thr_hi) & bit) != 0ull;
            decided = ca | cr;
            accepted = ca;
        }
        if (!decided) {
            dH = wave_sum_all(e);
            if (HAS_EW) dH += P.ew_coef * dEw;
            if (HAS_MU) dH -= dMu;
            const double lu = REPLAY ? lu_rp
                                     : __hiloint2double((int)rdlane((uint32_t)__double2hiint(logu), l4),
                                                        (int)rdlane((uint32_t)__double2loint(logu), l4));
            const double exponent = nbeta * dH + 0.0 + dB; // metropolis.py:41-44
            accepted = __ballot((exponent >= 0.0) || (exponent > lu)) != 0ull;
        }
        nacc_before = nacc_add;
        MULTI_PHASE(2)
        if (accepted) {
            bias_acc += dB;
#pragma unroll
            for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k) chg[k] += dQ[k];
            if (WLK) {
                // the state (bin, features) ends here: its post-steps go to the bin's row (sums)
                if (wl_sum_mode) wl_flush_run();
                MULTI_PHASE(6)
                // _do_accept_step (wanglandau.py:204-220): features and enthalpy follow the step
                const double *fsp = s_fs + ((size_t)cls1 * NSLOT) * 64 + lane;
                const uint32_t *ftp = s_ft + ((size_t)cls1 * NSLOT) * 64 + lane;
                if (KFW) {
                    // K function deltas per slot from the function tables of the slot's group, which starts at
                    // doff8 * SMOLMC_LEAN_MAX_KF in LeanParams::dtk (as mc_lean_kernel's KF variant, mc_lean.h)
                    const unsigned char *dtk = (const unsigned char *)P.dtk;
                    const uint32_t ktab8 = P.ktab8;
#pragma unroll
                    for (int it = 0; it < NSLOT; ++it) {
                        const uint32_t ft = ftp[it * 64], K = ft >> 16, f0 = ft & 0xffffu;
                        const uint32_t doff8 = (ONE ? rcs[ONE ? it : 0] : rec1[it * 64]).doff8;
                        const uint32_t g1 = doff8 * (uint32_t)(SMOLMC_LEAN_MAX_KF - 1) + ad1[KFW ? it : 0];
                        const uint32_t g2 = doff8 * (uint32_t)(SMOLMC_LEAN_MAX_KF - 1) + ad2[KFW ? it : 0];
                        const double fsc = fsp[it * 64];
#pragma unroll
                        for (int k = 0; k < SMOLMC_LEAN_MAX_KF; ++k) {
                            const bool on = (uint32_t)k < K;
                            double dk = *(const double *)(dtk + (g1 + (on ? (uint32_t)k * ktab8 : 0u)));
                            if (STEP == SMOLMC_STEP_SWAP) dk += *(const double *)(dtk + (g2 + (on ? (uint32_t)k * ktab8 : 0u)));
                            __hip_atomic_fetch_add(&s_feat[f0 + (on ? (uint32_t)k : 0u) + wl_shadow], on ? fsc * dk : 0.0, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WAVEFRONT);
                        }
                    }
                } else {
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) {
                    const double d = STEP == SMOLMC_STEP_SWAP ? d1[it] + d2[it] : d1[it];
                    __hip_atomic_fetch_add(&s_feat[ftp[it * 64] + wl_shadow], fsp[it * 64] * d, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
                }
                double df = 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) df += s_feat[k < wl_k ? wl_rd0 + k * wl_rdstep : 63];
                if (wl_zero_lane) s_feat[lane] = 0.0;
                if (HAS_EW) df += lane == f_ew ? dEw : 0.0;
                if (HAS_MU) df += lane == f_mu ? dMu : 0.0;
                fcur += df;
                Hcur += dH;
                wb = wnb;
            } else {
                double *cell = s_acc + ((size_t)cls1 * NSLOT) * 64 + lane;
#pragma unroll
                for (int it = 0; it < NSLOT; ++it)
                    cell[it * 64] += STEP == SMOLMC_STEP_SWAP ? d1[it] + d2[it] : d1[it];
            }
            if (STEP == SMOLMC_STEP_FLIP) occ[a1] = (uint8_t)n1;
            if (STEP == SMOLMC_STEP_SWAP) occ[a2] = (uint8_t)n2;
            MULTI_PHASE(3)
            if (HAS_EW) {
                bool swept = false;
                if constexpr (WLE16) {
                  if (have_e0w) {
                    swept = true;
                    // all 16 groups in one batch: the E8 offsets from registers, S8 from the proposal batch (a swap partner
                    // that did not come from the batch's candidates -- rare -- reads its own)
                    const LeanParamsKernarg Q = rare_params();
                    const unsigned char *gxa = (const unsigned char *)Q->ew_gx;
                    const uint32_t s8a = rdlane(vS8s, l4);
                    if (STEP == SMOLMC_STEP_SWAP) {
                        if (dq1 != 0.0 || dq2 != 0.0) {
                            const uint32_t s8b = s8_2 != 0xffffffffu ? s8_2 : Q->ew_S8[s2 - abase];
                            const uint32_t s8[2] = {s8a, s8b};
                            const double dq[2] = {dq1, dq2};
                            field_sweep_gx_groups<2, 16>(phi + lane, e0w, gxa, s8, dq, 0);
                        }
                    } else if (dq1 != 0.0) {
                        const uint32_t s8[1] = {s8a};
                        const double dq[1] = {dq1};
                        field_sweep_gx_groups<1, 16>(phi + lane, e0w, gxa, s8, dq, 0);
                    }
                  }
                }
                if (swept) {
                } else if (phi_lds) {
                    if (STEP == SMOLMC_STEP_SWAP) {
                        if (dq1 != 0.0 || dq2 != 0.0) field_apply2<1>(P, phi, lane, s1, dq1, s2, dq2);
                    } else if (dq1 != 0.0) {
                        field_apply<1>(P, phi, lane, s1, dq1);
                    }
                } else if (pend_on) {
                    const int nnew = (dq1 != 0.0 ? 1 : 0) + ((STEP == SMOLMC_STEP_SWAP && dq2 != 0.0) ? 1 : 0);
                    if (npend + nnew > 4) flush_pending();
                    if (dq1 != 0.0) push_pending(s1, dq1);
                    if (STEP == SMOLMC_STEP_SWAP && dq2 != 0.0) push_pending(s2, dq2);
                } else {
                    double *phi_g = P.ew_phi + (size_t)r * P.ew_nact;
                    if (STEP == SMOLMC_STEP_SWAP) {
                        if (dq1 != 0.0 || dq2 != 0.0) field_apply2<0>(P, phi_g, lane, s1, dq1, s2, dq2);
                    } else if (dq1 != 0.0) {
                        field_apply<0>(P, phi_g, lane, s1, dq1);
                    }
                }
            }
            acc_mu += dMu;
            acc_ew += dEw;
            nacc_add++;
            MULTI_PHASE(4)
        } else if (STEP == SMOLMC_STEP_SWAP) {
            occ[a1] = (uint8_t)o1;
        }
        if (!ONE) row1 = rown;
        if (WLK) { // WangLandau._do_post_step (wanglandau.py:222-266), accepted or not
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
        if (!ONE) __builtin_amdgcn_s_setprio(0);
#endif
        MULTI_PHASE(5)
        if (REPLAY && WLK) {
            if (lane == 0) {
                const size_t k = (size_t)r * nsteps32 + it_step;
                P.rp_acc[k] = (uint8_t)(nacc_add != nacc_before);
                P.rp_H[k] = Hcur;
            }
        } else if (REPLAY) { // accept flag and running enthalpy of every step (what smolmc_replay returns)
            double lane_e = 0.0;
            for (int i = lane; i < nrec; i += 64) lane_e = fma(s_rec[i].w, s_acc[i], lane_e);
            const double Hnow = H + (wave_sum_all(lane_e) - acc_mu + (HAS_EW ? P.ew_coef * acc_ew : 0.0));
            if (lane == 0) {
                const size_t k = (size_t)r * nsteps32 + it_step;
                P.rp_acc[k] = (uint8_t)(nacc_add != nacc_before);
                P.rp_H[k] = Hnow;
            }
        }

        if (--smp_countdown == 0) {
            const LeanParamsKernarg Q = rare_params(); // (sampling parameters re-read from the kernel arguments)
            const int qF = Q->F, qFce = Q->Fce;
            double *const q_feat = Q->smp.feat;
            const LeanSlot *const q_slots = Q->slots;
            smp_countdown = (uint32_t)Q->smp.every;
            const size_t row = (size_t)smp_index * Q->R + r;
            smp_index++;
            double Hnow;
            if (WLK) {
                if (lane < qF) q_feat[row * qF + lane] = fcur;
                Hnow = Hcur;
            } else {
            s_feat[lane] = 0.0;
            double lane_e = 0.0;
            for (int i = lane; i < nrec; i += 64) {
                const LeanSlot sl = q_slots[i];
                const double v = s_acc[i];
                lane_e = fma(sl.w, v, lane_e);
                if (sl.live && v != 0.0)
                    __hip_atomic_fetch_add(&s_feat[sl.feat], sl.fs * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            if (lane < qFce) q_feat[row * qF + lane] = base_feat + s_feat[lane];
            if (HAS_EW && lane == qFce) q_feat[row * qF + lane] = base_feat + acc_ew;
            if (HAS_MU && lane == qFce + (HAS_EW ? 1 : 0)) q_feat[row * qF + lane] = base_feat + acc_mu;
            Hnow = H + (wave_sum_all(lane_e) - acc_mu + (HAS_EW ? Q->ew_coef * acc_ew : 0.0));
            }
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

#ifdef SMOLMC_EXP_PHASES
    if (r == 0 && lane == 0)
        printf("multi phases (cycles per step): skeleton+proposal %.0f | gathers+tables %.0f | decision %.0f | accept: features+occupancy %.0f | "
               "field %.0f | post-step+rest %.0f | (row flush %.0f)\n",
               (double)mph[0] / (double)nsteps32, (double)mph[1] / (double)nsteps32, (double)mph[2] / (double)nsteps32,
               (double)mph[3] / (double)nsteps32, (double)mph[4] / (double)nsteps32, (double)mph[5] / (double)nsteps32, (double)mph[6] / (double)nsteps32);
#endif
    // ---- write back ---------------------------------------------------------------
    if (pend_on && npend > 0) flush_pending(); // (phi in HBM is complete between launches)
    if (phi_lds)
        for (int j = lane; j < P.ew_nact; j += 64) P.ew_phi[(size_t)r * P.ew_nact + j] = phi[j];
    {
        uint32_t *dst = (uint32_t *)(P.occ + (size_t)r * P.Npad);
        for (int i = lane; i < P.Npad / 4; i += 64)
            dst[i] = *(const uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb));
    }
    if (WLK) {
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
        H = Hcur;
        if (lane == 0) {
            P.wl.m[r] = wl_m;
            P.wl.counter[r] = wl_counter0 + (long long)nsteps32;
        }
    } else {
    s_feat[lane] = 0.0;
    double lane_e = 0.0;
    for (int i = lane; i < nrec; i += 64) {
        const LeanSlot sl = P.slots[i];
        const double v = s_acc[i];
        lane_e = fma(sl.w, v, lane_e);
        if (sl.live && v != 0.0)
            __hip_atomic_fetch_add(&s_feat[sl.feat], sl.fs * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    if (lane < P.Fce) featp[lane] = base_feat + s_feat[lane];
    H += wave_sum_all(lane_e) - acc_mu + (HAS_EW ? P.ew_coef * acc_ew : 0.0);
    }
    if (btype && lane == 0) {
        P.bias[r] += bias_acc;
#pragma unroll
        for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k)
            if (k < brows) P.charge[(size_t)r * SMOLMC_MAX_BIAS_ROWS + k] = chg[k];
    }
    if (lane == 0) {
        if (HAS_EW && !WLK) featp[P.Fce] += acc_ew;
        if (HAS_MU && !WLK) featp[P.Fce + (HAS_EW ? 1 : 0)] += acc_mu;
        P.enthalpy[r] = H;
        P.nsteps[r] = step;
        P.nacc[r] += nacc_add;
        if (nsteps32) P.last_acc[r] = (uint8_t)(nacc_add != nacc_before);
    }
    if (REPLAY && lane == 0 && rp_bad) atomicOr(P.rp_err, 1);
}

template <int NSLOT, int MM, int STEP, bool MU, int EW, bool BIAS = false, bool REPLAY = false, int WLK = 0>
static int launch_multi_inst(smolmc_handle *h, const LeanParams &lp) {
    const unsigned wpb = (unsigned)h->waves_per_block_lean;
    const unsigned grid = (unsigned)((h->R + wpb - 1) / wpb);
    // one site class with more than 256 clusters per site: slot records in registers
    auto kern = (NSLOT == 8 && lp.m_ncls == 1) ? mc_lean_multi_kernel<NSLOT, MM, STEP, MU, EW, NSLOT == 8, BIAS, REPLAY, WLK>
                                               : mc_lean_multi_kernel<NSLOT, MM, STEP, MU, EW, false, BIAS, REPLAY, WLK>;
    if (h->lean_lds > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h->lean_lds));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wpb), h->lean_lds, h->stream, lp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return 0;
}
template <int NSLOT, int MM, int STEP, bool BIAS> static int launch_multi_me(smolmc_handle *h, const LeanParams &lp) {
    const bool mu = lp.m_mu != nullptr;
    if (lp.ew_field == 1)
        return mu ? launch_multi_inst<NSLOT, MM, STEP, true, 1, BIAS>(h, lp) : launch_multi_inst<NSLOT, MM, STEP, false, 1, BIAS>(h, lp);
    if (lp.ew_field == 2)
        return mu ? launch_multi_inst<NSLOT, MM, STEP, true, 2, BIAS>(h, lp) : launch_multi_inst<NSLOT, MM, STEP, false, 2, BIAS>(h, lp);
    return mu ? launch_multi_inst<NSLOT, MM, STEP, true, 0, BIAS>(h, lp) : launch_multi_inst<NSLOT, MM, STEP, false, 0, BIAS>(h, lp);
}
template <int NSLOT, int MM, bool BIAS = false> static int launch_multi_nm(smolmc_handle *h, const LeanParams &lp) {
    if (h->cfg.step_type == SMOLMC_STEP_SWAP) return launch_multi_me<NSLOT, MM, SMOLMC_STEP_SWAP, BIAS>(h, lp);
    return launch_multi_me<NSLOT, MM, SMOLMC_STEP_FLIP, BIAS>(h, lp);
}
// Wang-Landau variants (multi_wl_n*.hip, multi_wl_replay_n*.hip; W = 2: several correlation functions per orbit, multi_wl_kf_n*.hip)
template <int NSLOT, int MM, int STEP, bool REPLAY, int W> static int launch_multi_wl_me(smolmc_handle *h, const LeanParams &lp) {
    const bool mu = lp.m_mu != nullptr;
    if (lp.ew_field == 1)
        return mu ? launch_multi_inst<NSLOT, MM, STEP, true, 1, false, REPLAY, W>(h, lp) : launch_multi_inst<NSLOT, MM, STEP, false, 1, false, REPLAY, W>(h, lp);
    if (lp.ew_field == 2)
        return mu ? launch_multi_inst<NSLOT, MM, STEP, true, 2, false, REPLAY, W>(h, lp) : launch_multi_inst<NSLOT, MM, STEP, false, 2, false, REPLAY, W>(h, lp);
    return mu ? launch_multi_inst<NSLOT, MM, STEP, true, 0, false, REPLAY, W>(h, lp) : launch_multi_inst<NSLOT, MM, STEP, false, 0, false, REPLAY, W>(h, lp);
}
template <int NSLOT, bool REPLAY = false, int W = 1> static int launch_multi_wl_nslot(smolmc_handle *h, const LeanParams &lp) {
    const bool swap = h->cfg.step_type == SMOLMC_STEP_SWAP;
    if (h->lean_mm == 2)
        return swap ? launch_multi_wl_me<NSLOT, 2, SMOLMC_STEP_SWAP, REPLAY, W>(h, lp) : launch_multi_wl_me<NSLOT, 2, SMOLMC_STEP_FLIP, REPLAY, W>(h, lp);
    return swap ? launch_multi_wl_me<NSLOT, 3, SMOLMC_STEP_SWAP, REPLAY, W>(h, lp) : launch_multi_wl_me<NSLOT, 3, SMOLMC_STEP_FLIP, REPLAY, W>(h, lp);
}
// replay variants (multi_replay_n*.hip)
template <int NSLOT, int MM, int STEP> static int launch_multi_replay_me(smolmc_handle *h, const LeanParams &lp) {
    const bool mu = lp.m_mu != nullptr;
    if (lp.ew_field == 1)
        return mu ? launch_multi_inst<NSLOT, MM, STEP, true, 1, false, true>(h, lp) : launch_multi_inst<NSLOT, MM, STEP, false, 1, false, true>(h, lp);
    if (lp.ew_field == 2)
        return mu ? launch_multi_inst<NSLOT, MM, STEP, true, 2, false, true>(h, lp) : launch_multi_inst<NSLOT, MM, STEP, false, 2, false, true>(h, lp);
    return mu ? launch_multi_inst<NSLOT, MM, STEP, true, 0, false, true>(h, lp) : launch_multi_inst<NSLOT, MM, STEP, false, 0, false, true>(h, lp);
}
template <int NSLOT> static int launch_multi_replay_nslot(smolmc_handle *h, const LeanParams &lp) {
    const bool swap = h->cfg.step_type == SMOLMC_STEP_SWAP;
    if (h->lean_mm == 2)
        return swap ? launch_multi_replay_me<NSLOT, 2, SMOLMC_STEP_SWAP>(h, lp) : launch_multi_replay_me<NSLOT, 2, SMOLMC_STEP_FLIP>(h, lp);
    return swap ? launch_multi_replay_me<NSLOT, 3, SMOLMC_STEP_SWAP>(h, lp) : launch_multi_replay_me<NSLOT, 3, SMOLMC_STEP_FLIP>(h, lp);
}
// biased replay variants (multi_bias_replay_n*.hip)
template <int NSLOT, int MM, int STEP> static int launch_multi_bias_replay_me(smolmc_handle *h, const LeanParams &lp) {
    const bool mu = lp.m_mu != nullptr;
    if (lp.ew_field == 1)
        return mu ? launch_multi_inst<NSLOT, MM, STEP, true, 1, true, true>(h, lp) : launch_multi_inst<NSLOT, MM, STEP, false, 1, true, true>(h, lp);
    if (lp.ew_field == 2)
        return mu ? launch_multi_inst<NSLOT, MM, STEP, true, 2, true, true>(h, lp) : launch_multi_inst<NSLOT, MM, STEP, false, 2, true, true>(h, lp);
    return mu ? launch_multi_inst<NSLOT, MM, STEP, true, 0, true, true>(h, lp) : launch_multi_inst<NSLOT, MM, STEP, false, 0, true, true>(h, lp);
}
template <int NSLOT> static int launch_multi_bias_replay_nslot(smolmc_handle *h, const LeanParams &lp) {
    const bool swap = h->cfg.step_type == SMOLMC_STEP_SWAP;
    if (h->lean_mm == 2)
        return swap ? launch_multi_bias_replay_me<NSLOT, 2, SMOLMC_STEP_SWAP>(h, lp) : launch_multi_bias_replay_me<NSLOT, 2, SMOLMC_STEP_FLIP>(h, lp);
    return swap ? launch_multi_bias_replay_me<NSLOT, 3, SMOLMC_STEP_SWAP>(h, lp) : launch_multi_bias_replay_me<NSLOT, 3, SMOLMC_STEP_FLIP>(h, lp);
}
template <int NSLOT> static int launch_multi_bias_nslot(smolmc_handle *h, const LeanParams &lp) {
    return h->lean_mm == 2 ? launch_multi_nm<NSLOT, 2, true>(h, lp) : launch_multi_nm<NSLOT, 3, true>(h, lp);
}


// ----------------------------------------------------------------------------
// TableFlip for several active sublattices (mcusher.py:397-711): flip vectors span the
// species of ALL active sublattices ("counts" dimensions d = sum of their species, <= 16),
// e.g. Li+ + F- <-> Mn3+ + O2- style charge-neutral exchanges between the cation and anion
// sublattices.  Same algorithm as mc_table_kernel (feasibility by compare + ballot on
// lane-indexed counts, scalar ballot scan of the candidate stream, a-priori factor, sequential
// evaluation with tentative LDS writes) on the multi-class state layout of
// mc_lean_multi_kernel; sites of the depleted species are drawn sublattice by sublattice from
// ONE candidate stream, exactly as the oracle does.
// ----------------------------------------------------------------------------
// EWM: 0 = no Ewald term, 1 = potential field in LDS, 2 = in HBM (compile time, see mc_lean_multi_kernel)
// REPLAY: host-provided step records (see mc_table_kernel)
// WLT (round 6; multi_table_wl_n*.hip): the Wang-Landau kernel with TableFlip proposals (any update_period; see
// mc_table_kernel): the accept rule S[bin] - S[new bin] + a-priori factor (wanglandau.py:197-198), the per-walker
// state of mc_lean_multi_kernel's WLK variant in place of the accumulator cells.
// BIAS (round 6; multi_table_bias_n*.hip): an MCBias term in the exponent (metropolis.py:43-44), one pair table per
// sublattice and bias row, bias_pair[row][sublattice][old * 8 + new] (see mc_table_kernel).
template <int NSLOT, int MM, int EWM, bool REPLAY = false, bool WLT = false, bool BIAS = false>
__global__ void __launch_bounds__(512) mc_table_multi_kernel(const LeanParams P) {
    static_assert(!(WLT && REPLAY), "Wang-Landau TableFlip replays take the universal kernel");
    static_assert(!(WLT && BIAS), "Cannot apply bias to Wang-Landau simulation (wanglandau.py:127-128)");
    static_assert(!(BIAS && REPLAY), "biased TableFlip replays take the universal kernel");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int r = uni(blockIdx.x * nwaves + wave);
    const int NC = P.m_ncls, NS = P.m_nsub;
    const bool has_mu = P.m_mu != nullptr;
    constexpr bool has_ew = EWM != 0;
    // block-shared: dt | mu [4][8] | q [4][8] | dg [4][8] | weights [16] | flip table [8][16] ints | records
    double *s_dt = (double *)smem;
    double *s_mu = s_dt + P.dt_len;
    double *s_q = s_mu + 32, *s_dg = s_mu + 64;
    double *s_tfw = s_mu + 96;
    int *s_tf = (int *)(s_tfw + 16);
    MultiRec *s_rec = (MultiRec *)(s_tfw + 16 + 64);
    const int nrec = NC * NSLOT * 64;
    // (WLT) feature scale and feature index of every slot record, read on accepted steps only
    double *s_fs = (double *)(s_rec + nrec);
    uint32_t *s_ft = (uint32_t *)(s_fs + (WLT ? nrec : 0));
    unsigned char *shared_end = (unsigned char *)(s_ft + (WLT ? ((nrec + 3) & ~3) : 0));
    // per wave: occupancy | 64 B (species counts) | feature scratch [64] | acc cells | pending cells | phi
    // (WLT: S [L] | counted steps [L] | log of finished runs [SMOLMC_WLM_LOG][F] instead of the acc cells)
    constexpr bool phi_lds = EWM == 1;
    const int wl_sum_mode = WLT ? P.wl.sum_mode : 1;
    const size_t state_bytes = WLT ? wl_multi_wave_bytes(P.wl.L, P.F, wl_sum_mode) : (size_t)nrec * 8;
    const size_t per_wave = (size_t)P.Nlds + 64 + 64 * 8 + state_bytes + (size_t)nrec * 8 + (phi_lds ? (size_t)P.ew_nact * 8 : 0);
    unsigned char *wbase = shared_end + (size_t)wave * per_wave;
    uint8_t *occ = wbase;
    int *s_cnt = (int *)(wbase + P.Nlds);
    double *s_feat = (double *)(wbase + P.Nlds + 64);
    double *s_acc = s_feat + 64;
    double *wl_S = s_acc;                                                   // WLT
    uint32_t *wl_cnt = (uint32_t *)(wl_S + (WLT ? P.wl.L : 0));
    double *wl_occb = (double *)((unsigned char *)wl_cnt + (WLT ? (((size_t)P.wl.L * 4 + 7) & ~(size_t)7) : 0));
    double *s_rows = wl_occb + ((WLT && !wl_sum_mode) ? P.wl.L : 0);
    double *s_pend = (double *)((unsigned char *)s_acc + state_bytes);
    double *phi = phi_lds ? s_pend + nrec : P.ew_phi + (size_t)r * P.ew_nact;
    const int swa = P.swz_a, swm = P.swz_m, swb = P.swz_b;
    const int D = P.m_ndims;
    for (int i = threadIdx.x; i < P.dt_len; i += blockDim.x) s_dt[i] = P.dt[i];
    if (threadIdx.x < 32) {
        s_mu[threadIdx.x] = has_mu ? P.m_mu[threadIdx.x] : 0.0;
        s_q[threadIdx.x] = has_ew ? P.m_q[threadIdx.x] : 0.0;
        s_dg[threadIdx.x] = has_ew ? P.m_dg[threadIdx.x] : 0.0;
    }
    for (int i = threadIdx.x; i < 2 * P.tf_n; i += blockDim.x) s_tfw[i] = P.tf_w[i];
    for (int i = threadIdx.x; i < P.tf_n * D; i += blockDim.x) s_tf[i] = P.tf_table[i];
    for (int i = threadIdx.x; i < nrec; i += blockDim.x) {
        const LeanSlot sl = P.slots[i];
        MultiRec rec;
        rec.doff8 = sl.doff8;
        rec.st8[0] = sl.stride8[0]; rec.st8[1] = sl.stride8[1]; rec.st8[2] = sl.stride8[2];
        rec.w = sl.w;
        s_rec[i] = rec;
        if (WLT) {
            s_fs[i] = sl.live ? sl.fs : 0.0;
            s_ft[i] = sl.feat;
        }
    }
    const bool live = r < P.R;
    if (live) {
        const uint32_t *src = (const uint32_t *)(P.occ + (size_t)r * P.Npad);
        for (int i = lane; i < P.Npad / 4; i += 64)
            *(uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb)) = src[i];
        s_feat[lane] = 0.0;
        if (lane < 16) s_cnt[lane] = 0;
        if (WLT) {
            for (int i = lane; i < nrec; i += 64) s_pend[i] = 0.0;
            for (int i = lane; i < P.wl.L; i += 64) {
                wl_S[i] = P.wl.entropy[(size_t)r * P.wl.L + i];
                wl_cnt[i] = 0u;
                if (!wl_sum_mode) wl_occb[i] = (double)P.wl.occur[(size_t)r * P.wl.L + i];
            }
            if (!wl_sum_mode)
                for (int i = lane; i < SMOLMC_WL_ROWS * P.F; i += 64) s_rows[i] = 0.0;
        } else {
            for (int i = lane; i < 2 * nrec; i += 64) s_acc[i] = 0.0; // acc + pending cells
        }
        if (phi_lds)
            for (int j = lane; j < P.ew_nact; j += 64) phi[j] = P.ew_phi[(size_t)r * P.ew_nact + j];
    }
    __syncthreads();
    if (!live) return;
    const uint32_t nt8 = P.nt8, snt8 = P.snt8;
    const int abase = P.ew_act_base;
    // "counts" dimension of this lane: sublattice, species code, first dimension of the sublattice
    int dim_sub = 0;
    {
        int b = 0;
        for (int k = 0; k < NS; ++k) {
            const int nck = sel4(P.m_ncodes, k);
            if (lane >= b && lane < b + nck) dim_sub = k;
            b += nck;
        }
    }
    const int dim_max = lane < D ? sel4(P.m_nact, dim_sub) : 0;
    for (int k = 0; k < NS; ++k) {
        const int sb = sel4(P.m_sbase, k), na = sel4(P.m_nact, k);
        int b = 0;
        for (int q = 0; q < k; ++q) b += sel4(P.m_ncodes, q);
        for (int a = lane; a < na; a += 64) atomicAdd(&s_cnt[b + (int)occ[lean_swz(sb + a, swa, swm, swb)]], 1);
    }
    int vcnt = lane < D ? s_cnt[lane] : 0;
    int vtf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) vtf[i] = (i < P.tf_n && lane < D) ? s_tf[i * D + lane] : 0;
    auto feasible = [&](const int vc) -> unsigned { // flip_weights_mask (math.py:832-867)
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < P.tf_n) {
                const int vp = vc + vtf[i], vm = vc - vtf[i];
                if (__ballot(vp < 0 || vp > dim_max) == 0ull) m |= 1u << (2 * i);
                if (__ballot(vm < 0 || vm > dim_max) == 0ull) m |= 2u << (2 * i);
            }
        return m;
    };
    const double vw = lane < 2 * P.tf_n ? s_tfw[lane] : 0.0;
    auto weight_of = [&](const int idx) -> double {
        return __hiloint2double((int)rdlane((uint32_t)__double2hiint(vw), idx),
                                (int)rdlane((uint32_t)__double2loint(vw), idx));
    };
    auto masked_sum = [&](const unsigned m) -> double {
        double sw = 0.0;
        for (int idx = 0; idx < 2 * P.tf_n; ++idx)
            if ((m >> idx) & 1u) sw += weight_of(idx);
        return sw;
    };
    auto sub_of = [&](uint32_t w0) -> int {
        if (NS == 1) return 0;
        const double x = (double)w0 * (1.0 / 4294967296.0);
        int sl = NS - 1; // (constant indices: a dynamic index into kernel arguments goes through scratch)
        if (NS > 3 && x < P.m_cum[2]) sl = 2;
        if (NS > 2 && x < P.m_cum[1]) sl = 1;
        if (x < P.m_cum[0]) sl = 0;
        return sl;
    };

    double H = P.enthalpy[r];
    double H_rp = H; // replay: running enthalpy reported per step
    const double nbeta = -P.beta[r];
    unsigned long long step = P.nsteps[r];
    uint32_t nacc_add = 0, nacc_before = 0;
    const uint32_t key0_ = (uint32_t)P.seeds[r], key1_ = (uint32_t)(P.seeds[r] >> 32);
    // (opaque at every Philox call: the twenty loop-invariant round keys otherwise live in SGPRs
    // across the step loop, see mc_table_kernel)
#define key0 opaque_u32(key0_)
#define key1 opaque_u32(key1_)
    double acc_mu = 0.0, acc_ew = 0.0;
    // MCBias: running bias and, for the square biases, the running A_k . n - b_k of every hyperplane
    const int tb_type = BIAS ? P.bias_type : 0;
    const int tb_rows = (tb_type && tb_type != SMOLMC_BIAS_FUGACITY) ? P.bias_rows : 0;
    double tb_acc = 0.0, tb_chg[SMOLMC_MAX_BIAS_ROWS];
#pragma unroll
    for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k) tb_chg[k] = (BIAS && k < tb_rows) ? P.charge[(size_t)r * SMOLMC_MAX_BIAS_ROWS + k] : 0.0;
    double *featp = P.features + (size_t)r * P.F;
    const double base_feat = lane < P.F ? featp[lane] : 0.0;
    // ---- Wang-Landau state (WLT; mc_lean_multi_kernel's WLK with update_period 1: see there) ----
    double fcur = base_feat;  // lane f < F: the walker's current feature vector; H is carried as _current_enthalpy
    double wl_m = WLT ? P.wl.m[r] : 0.0;
    int wb = 0;
    if (WLT) wb = min(max(uni((int)floordiv_exact(H - P.wl.vmin, P.wl.bin)), 0), P.wl.L - 1);
    const long long wl_counter0 = WLT ? P.wl.counter[r] : 0;
    const uint32_t wl_check = WLT ? (uint32_t)P.wl.check : 0u, wl_upd = WLT ? (uint32_t)P.wl.update : 1u;
    uint32_t wl_rem_check = (WLT && wl_check) ? (uint32_t)uni((int)(wl_counter0 % (long long)wl_check)) : 1u; // (0 = no check)
    uint32_t wl_rem_upd = WLT ? (uint32_t)uni((int)(wl_counter0 % (long long)wl_upd)) : 0u;
    uint32_t wl_run_n = 0;
    int vtag = -1, wl_nlog = 0;
    auto wl_row_of = [&](const int bin) -> double * { // running means: the cached row of a bin (see mc_lean_multi_kernel)
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
    const int wl_F = WLT ? P.F : 1;
    const int wl_k = max(1, min(8, 63 / max(wl_F, 1)));
    const uint32_t wl_shadow = (uint32_t)((lane % wl_k) * wl_F);
    const int wl_rd0 = lane < wl_F ? lane : 63, wl_rdstep = lane < wl_F ? wl_F : 0;
    const bool wl_zero_lane = lane < wl_k * wl_F;
    const int wl_epi = max(1, 64 / max(wl_F, 1)), wl_lane_e = lane / max(wl_F, 1), wl_lane_f = lane - wl_lane_e * wl_F;
    auto wl_log_flush = [&]() {
        const LeanParamsKernarg Q = rare_params();
        double *grows = Q->wl.meanf + (size_t)r * Q->wl.L * Q->F;
        const int qF = Q->F;
        for (int base = 0; base < wl_nlog; base += wl_epi) {
            const int e = base + wl_lane_e;
            const int bin = __shfl(vtag, e & 63); // (uniform control flow)
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
            if (++wl_nlog == WLM_LOG) wl_log_flush();
        }
    };
    uint32_t smp_countdown = P.smp.every ? (uint32_t)P.smp.every : 0xffffffffu; // (off: cannot reach zero in a launch)
    uint32_t smp_index = 0;
    uint32_t W0 = 0, W1 = 0, W2 = 0, W3 = 0;
    uint32_t batch_base = ~0u; // low word of the batch's first step
    uint32_t w_site_carry = 0;
    constexpr int ROW = NSLOT * MM;
    constexpr int NW = ROW / 2;
    constexpr uint32_t SITE_BYTES = 64u * ROW * 2u;
    const __amdgpu_buffer_rsrc_t idx_rs =
        __builtin_amdgcn_make_buffer_rsrc((void *)P.idx, 0, 0x7fffffff, 0x00020000);
    const uint32_t lane_voff = (uint32_t)lane * (ROW * 2u);
    const int nf2 = 2 * P.tf_n;
    bool head_valid = false; // feasibility mask / weight sum / a-priori factors cached between count changes
    unsigned feas_now = 0, lp_valid = 0;
    double sumw = 0.0, vlp = 0.0, vcum = 0.0;
    int last_feas = -1;
    // Candidate block (see mc_table_kernel): the first 32 words of the candidate stream of TWO
    // steps, one per lane, from one Philox call per two steps.  A sublattice maps the same words to
    // its own sites; a step that finds all its picks among them skips the 256-candidate rounds.
    uint32_t cblk_base = ~0u, cb_word = 0u;

    // ---- proposal batch (see mc_table_kernel: the proposals of 64 consecutive steps at once, lane
    // l <-> step (step & ~63) + l, stale marking after accepted steps, the step-at-a-time code below
    // as the definition and the fallback).  Here the picks of a direction run over the active
    // sublattices in turn: lane idx (< 2 tf_n) holds, as nibbles in pick order, the (local) species
    // and the sublattice of every depleted pick, and the same for the enriched entries in the
    // order the assignment draws them; a candidate word maps to a site of the sublattice of the
    // pick it is examined for.
    uint32_t vdep_c = 0, vdep_s = 0, venr_c = 0, venr_s = 0;
    int vncol = 0;
    if (lane < 2 * P.tf_n) {
        const int sgn = (lane & 1) ? -1 : 1;
        int ne = 0, db = 0;
        for (int sl = 0; sl < NS; ++sl) {
            const int ncod = sel4(P.m_ncodes, sl);
            for (int c = 0; c < ncod; ++c) {
                const int u = sgn * s_tf[(lane >> 1) * D + db + c];
                for (int z = 0; z < -u; ++z, ++vncol)
                    if (vncol < 8) { vdep_c |= (uint32_t)c << (4 * vncol); vdep_s |= (uint32_t)sl << (4 * vncol); }
                for (int z = 0; z < u; ++z, ++ne)
                    if (ne < 8) { venr_c |= (uint32_t)c << (4 * ne); venr_s |= (uint32_t)sl << (4 * ne); }
            }
            db += ncod;
        }
    }
    int vmaxu = 0; // lane d: the largest change of count dimension d in any direction
#pragma unroll
    for (int i = 0; i < 8; ++i) vmaxu = max(vmaxu, vtf[i] < 0 ? -vtf[i] : vtf[i]);
    auto all_feasible = [&](const int vc) -> bool {
        return __ballot(lane < D && (vc - vmaxu < 0 || vc + vmaxu > dim_max)) == 0ull;
    };
    const uint32_t lane4 = (uint32_t)(lane & 7) * 4u;
    uint32_t q_base = ~0u;           // low word of the batch's first step
    uint32_t q_meta = 0;             // bit 0 covered | bit 1 swap | bits 2-4 flips | bits 5-8 direction
    uint32_t q_s01 = 0, q_s23 = 0;   // sites of the flips (u16 each)
    uint32_t q_pack = 0;             // old species of flip f: nibble f; new species: nibble 4 + f
    uint32_t q_sub = 0;              // sublattice of flip f: nibble f
    uint32_t q_w1 = 0;               // W(step, 0, 1): the site word of the NEXT step
    double q_logu = 0.0;             // log of the acceptance uniform
    uint32_t q_c[16];                // examined candidate sites (u16 each; 0xffff: none)
#pragma unroll
    for (int i = 0; i < 16; ++i) q_c[i] = 0xffffffffu;
    unsigned long long q_stale = ~0ull;
    auto compute_head = [&]() {
        feas_now = feasible(vcnt);
        sumw = masked_sum(feas_now);
        head_valid = true;
        lp_valid = 0u;
        // running sums of the feasible weights, lane idx <-> direction idx (mc_table_kernel)
        double c = 0.0;
        last_feas = -1;
        for (int idx = 0; idx < nf2; ++idx)
            if ((feas_now >> idx) & 1u) {
                c += weight_of(idx);
                if (lane == idx) vcum = c;
                last_feas = idx;
            }
    };
    auto vsel4 = [&](const int a0, const int a1, const int a2, const int a3, const int k) -> int { // per-lane choice
        return k == 0 ? a0 : k == 1 ? a1 : k == 2 ? a2 : a3;
    };
    auto propose_batch = [&](const unsigned long long b0) { // b0: first step of the block
        const LeanParamsKernarg Q = rare_params();
        const int sb0 = Q->m_sbase[0], sb1 = Q->m_sbase[1], sb2 = Q->m_sbase[2], sb3 = Q->m_sbase[3];
        const int na0 = Q->m_nact[0], na1 = Q->m_nact[1], na2 = Q->m_nact[2], na3 = Q->m_nact[3];
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
        const double target = (double)o1.w[0] * (1.0 / 4294967296.0) * sumw;
        int d = -1;
        for (int idx = 0; idx < nf2; ++idx) {
            const double c = __hiloint2double((int)rdlane((uint32_t)__double2hiint(vcum), idx),
                                              (int)rdlane((uint32_t)__double2loint(vcum), idx));
            if (((feas_now >> idx) & 1u) && d < 0 && target < c) d = idx;
        }
        if (d < 0) d = last_feas;
        const uint32_t dep_c = (uint32_t)__shfl((int)vdep_c, d), dep_s = (uint32_t)__shfl((int)vdep_s, d);
        const uint32_t enr_c = (uint32_t)__shfl((int)venr_c, d), enr_s = (uint32_t)__shfl((int)venr_s, d);
        const int ncol = __shfl(vncol, d);
#pragma unroll
        for (int i = 0; i < 16; ++i) q_c[i] = 0xffffffffu;
        uint32_t meta = 0u, pack = 0u, subs = 0u;
        int col0 = -1, col1 = -1, col2 = -1, col3 = -1;
        if (is_swap) {
            // Swap.propose_step inside the sublattice picked by W(step, 1, 1); the first 12
            // candidates c_t = W(step, 1 + t % 3, t / 3)
            const philox_out o3 = philox_call(c0, c1, 3u, key0, key1);
            const int sl = sub_of(o1.w[1]);
            const int sb = vsel4(sb0, sb1, sb2, sb3, sl);
            const uint32_t na = (uint32_t)vsel4(na0, na1, na2, na3, sl);
            const int s1 = sb + (int)__umulhi(wsite, na);
            const int sp1 = (int)occ[lean_swz(s1, swa, swm, swb)];
            q_c[0] = 0xffff0000u | (uint32_t)s1;
            int found = -1, fo = 0;
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                const uint32_t w = (t % 3 == 0 ? o1 : t % 3 == 1 ? o2 : o3).w[t / 3];
                const int cs = sb + (int)__umulhi(w, na);
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
                subs = (uint32_t)sl | ((uint32_t)sl << 4);
                meta = 1u | 2u | (2u << 2);
            }
        }
        // table steps: one pass over the candidate stream c_t = W(step, 4 + t / 4, t % 4); the next
        // pick says which species is wanted and which sublattice maps the word to a site
        int k = 0;
        bool scanning = !is_swap && ncol >= 1 && ncol <= 4;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (__ballot(scanning) != 0ull) {
                const philox_out o = philox_call(c0, c1, 4u + (uint32_t)b, key0, key1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = 4 * b + j;
                    const int want = (int)((dep_c >> (4 * k)) & 15u), wsl = (int)((dep_s >> (4 * k)) & 15u);
                    const int cs = vsel4(sb0, sb1, sb2, sb3, wsl) + (int)__umulhi(o.w[j], (uint32_t)vsel4(na0, na1, na2, na3, wsl));
                    const int v = (int)occ[lean_swz(scanning ? cs : 0, swa, swm, swb)];
                    if (scanning) {
                        q_c[t >> 1] = (t & 1) ? ((q_c[t >> 1] & 0x0000ffffu) | ((uint32_t)cs << 16))
                                              : ((q_c[t >> 1] & 0xffff0000u) | (uint32_t)cs);
                        const bool dup = cs == col0 || cs == col1 || cs == col2 || cs == col3;
                        if (v == want && !dup) {
                            col0 = k == 0 ? cs : col0;
                            col1 = k == 1 ? cs : col1;
                            col2 = k == 2 ? cs : col2;
                            col3 = k == 3 ? cs : col3;
                            pack |= (uint32_t)v << (4 * k);
                            subs |= (uint32_t)wsl << (4 * k);
                            k++;
                            scanning = k < ncol;
                        }
                    }
                }
            }
        }
        if (!is_swap && ncol >= 1 && ncol <= 4 && k == ncol) {
            // the random assignment (mcusher.py:627-631), sublattice by sublattice: draw q takes the
            // rr-th pick still available AMONG THE PICKS OF ITS SUBLATTICE
            uint32_t avail = (1u << ncol) - 1u;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < ncol) {
                    const uint32_t esl = (enr_s >> (4 * q)) & 15u;
                    uint32_t msl = 0u; // picks of that sublattice
#pragma unroll
                    for (int z = 0; z < 4; ++z) msl |= (z < ncol && ((subs >> (4 * z)) & 15u) == esl) ? 1u << z : 0u;
                    uint32_t m = avail & msl;
                    const uint32_t rr = __umulhi(o2.w[q], (uint32_t)__popc(m));
#pragma unroll
                    for (int z = 0; z < 3; ++z) m = z < (int)rr ? (m & (m - 1u)) : m;
                    const int pj = __ffs((int)m) - 1;
                    avail &= ~(1u << pj);
                    pack |= ((enr_c >> (4 * q)) & 15u) << (16 + 4 * pj);
                }
            meta = 1u | ((uint32_t)ncol << 2) | ((uint32_t)d << 5);
        }
        q_meta = meta;
        q_s01 = ((uint32_t)col0 & 0xffffu) | ((uint32_t)col1 << 16);
        q_s23 = ((uint32_t)col2 & 0xffffu) | ((uint32_t)col3 << 16);
        q_pack = pack;
        q_sub = subs;
    };

#ifdef SMOLMC_EXP_PHASES // experiment: shader cycles per phase of a step (walker 0 prints the averages)
    long long ph_acc[5] = {0, 0, 0, 0, 0}, ph_cov = 0;
    long long ph_t = clock64();
#endif
    const uint32_t nsteps32 = (uint32_t)P.steps;
    for (uint32_t it_step = 0; it_step < nsteps32; ++it_step, ++step) {
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
        const int l4 = (int)(step & 15ull) * 4;
        // feasibility mask / weight sums after the counts changed (one copy of the code); the batch's
        // directions assume the mask they were chosen under
        if (__builtin_expect(!head_valid, 0)) {
            const unsigned feas_old = feas_now;
            compute_head();
            if (feas_now != feas_old) q_stale = ~0ull;
        }
        if (!REPLAY && __builtin_expect((uint32_t)(step & ~63ull) != q_base, 0)) propose_batch(step & ~63ull);
        const int l6 = (int)(step & 63ull);
        const uint32_t q_m = rdlane(q_meta, l6);
#ifdef SMOLMC_NO_TABLE_BATCH // A/B switch: every step through the step-at-a-time proposal
        const bool covered = false;
#else
        const bool covered = !REPLAY && (q_m & 1u) != 0u && ((q_stale >> l6) & 1ull) == 0ull;
#endif
        double lu_rp = 0.0; // replay: log of the recorded uniform
#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); ph_acc[0] += tn - ph_t; ph_t = tn; if (covered) ph_cov++; }
#endif
        // flips of this step, lane-indexed: lane f holds flip f (site, new / old code, sublattice)
        int vsite = 0, vnew = 0, vold = 0, vfsub = 0;
        int nfl = 0, dir = -1;
        bool fast_ok = false;
        int vu = 0; // table step: lane d holds the change of count dimension d
        double log_priori = 0.0;
        if (covered) {
            // the batch's proposal of this step (see propose_batch)
            nfl = (int)((q_m >> 2) & 7u);
            const uint32_t a01 = rdlane(q_s01, l6), a23 = rdlane(q_s23, l6), pk = rdlane(q_pack, l6), sb4 = rdlane(q_sub, l6);
            vsite = lane == 0 ? (int)(a01 & 0xffffu) : lane == 1 ? (int)(a01 >> 16)
                  : lane == 2 ? (int)(a23 & 0xffffu) : (int)(a23 >> 16);
            vold = (int)((pk >> lane4) & 15u);
            vnew = (int)(((pk >> 16) >> lane4) & 15u);
            vfsub = (int)((sb4 >> lane4) & 15u);
            if (!(q_m & 2u)) {
                dir = (int)((q_m >> 5) & 15u);
#pragma unroll
                for (int i = 0; i < 8; ++i) vu = (i == (dir >> 1)) ? vtf[i] : vu;
                vu *= (dir & 1) ? -1 : 1;
            }
        } else if (REPLAY) {
            // the recorded step: lane f <-> flip f (see mc_table_kernel)
            const LeanParamsKernarg Q = rare_params();
            const size_t krec = (size_t)r * nsteps32 + it_step;
            const int *rec = Q->rp_steps + krec * SMOLMC_STEP_ROW;
            const int v = lane < SMOLMC_STEP_ROW ? rec[lane] : -1;
            while (nfl < SMOLMC_MAX_STEP_FLIPS && (int)rdlane((uint32_t)v, 2 * nfl) >= 0) nfl++;
            const int ra = __shfl(v, 2 * (lane & 7)), rb = __shfl(v, 2 * (lane & 7) + 1); // (uniform control flow)
            vsite = lane < nfl ? ra : sel4(P.m_sbase, 0);
            vnew = lane < nfl ? rb : 0;
            // sublattice of every flip, its dimension base; sites distinct inside a step
            int fsub = -1, fbase = 0;
            {
                int b = 0;
                for (int k = 0; k < NS; ++k) {
                    const int sb = sel4(P.m_sbase, k), na = sel4(P.m_nact, k), nck = sel4(P.m_ncodes, k);
                    if (vsite >= sb && vsite < sb + na) { fsub = k; fbase = b; if (vnew < 0 || vnew >= nck) fsub = -1; }
                    b += nck;
                }
            }
            bool bad = lane < nfl && fsub < 0;
            for (int f = 0; f < nfl; ++f) bad |= lane < nfl && lane != f && vsite == (int)rdlane((uint32_t)vsite, f);
            int rbad = __ballot(bad) != 0ull ? 2 : 0;
            if (rbad) { nfl = 0; vsite = sel4(P.m_sbase, 0); vnew = 0; fsub = 0; fbase = 0; }
            vfsub = lane < nfl ? fsub : 0;
            vold = lane < nfl ? (int)occ[lean_swz(vsite, swa, swm, swb)] : 0;
            for (int f = 0; f < nfl; ++f) {
                const int fb = (int)rdlane((uint32_t)fbase, f);
                vu += (lane == fb + (int)rdlane((uint32_t)vnew, f)) - (lane == fb + (int)rdlane((uint32_t)vold, f));
            }
            if (__ballot(lane < D && vu != 0) != 0ull) { // _get_flip_id (mcusher.py:641-654)
                const int tfn = Q->tf_n;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (i < tfn && dir < 0) {
                        if (__ballot(lane < D && vtf[i] != vu) == 0ull) dir = 2 * i;
                        else if (__ballot(lane < D && -vtf[i] != vu) == 0ull) dir = 2 * i + 1;
                    }
                if (dir < 0) { rbad |= 1; nfl = 0; vu = 0; } // "Step ... is not in flip table." (:673-674)
            }
            if (rbad && lane == 0) atomicOr(Q->rp_err, rbad);
            double u = uni_d(Q->rp_u[krec]);
            if (u != u) u = 0.0; // NaN: the reference accepted without drawing
            lu_rp = log(u);
        } else {
        word_batch();
        const uint32_t w_site = l4 == 0 ? w_site_carry : rdlane(W1, l4 - 4);
        bool do_swap = (double)rdlane(W0, l4) * (1.0 / 4294967296.0) < P.tf_sw;
        if (!do_swap) {
            if (!(sumw > 0.0)) do_swap = true;
        }
        if (do_swap) {
            // Swap.propose_step inside the sublattice picked by W(step, 1, 1) (mcusher.py:176-200)
            const int sl = sub_of(rdlane(W1, l4 + 1));
            const int sb = sel4(P.m_sbase, sl);
            const uint32_t na = (uint32_t)sel4(P.m_nact, sl);
            const int s1 = sb + (int)__umulhi(w_site, na);
            const int o1 = uni((int)occ[lean_swz(s1, swa, swm, swb)]);
            int found = -1, fo = 0;
            const uint32_t ws[4] = {W0, W1, W2, W3};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (found < 0) {
                    const int cs = sb + (int)__umulhi(ws[j], na);
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
                        const int cs = sb + (int)__umulhi(o.w[j], na);
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
                        for (uint32_t a = lane; a < na; a += 64)
                            any |= ((int)occ[lean_swz(sb + (int)a, swa, swm, swb)] != o1);
                        if (__ballot(any) == 0ull) break;
                    }
                }
            }
            if (found >= 0) {
                nfl = 2;
                vsite = lane == 0 ? s1 : found;
                vnew = lane == 0 ? fo : o1;
                vold = lane == 0 ? o1 : fo;
                vfsub = sl;
            }
        } else {
            // choose_section_from_partition (math.py:870-893) with W(step, 1, 0)
            const double target = (double)rdlane(W0, l4 + 1) * (1.0 / 4294967296.0) * sumw;
            {
                const uint32_t hit = (uint32_t)__ballot(target < vcum) & feas_now; // first feasible idx with target < running sum
                dir = hit ? __ffs((int)hit) - 1 : last_feas;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) vu = (i == (dir >> 1)) ? vtf[i] : vu;
            vu *= (dir & 1) ? -1 : 1;
#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); ph_acc[1] += tn - ph_t; ph_t = tn; }
#endif
            // sites of the depleted species, sublattice by sublattice, from the candidate stream
            // c_t = W(step, 4 + t / 4, t % 4) (256 candidates per wave round, position kept across
            // species AND sublattices); then the random assignment to the enriched species
            if ((uint32_t)(step & ~1ull) != cblk_base) {
                cblk_base = (uint32_t)(step & ~1ull);
                const unsigned long long sl2 = (step & ~1ull) + (unsigned)(lane >> 5);
                const uint32_t tt = (uint32_t)lane & 31u;
                const philox_out o = philox4x32_10((uint32_t)sl2, (uint32_t)(sl2 >> 32), 4u + (tt >> 2), 0u, key0, key1);
                cb_word = (tt & 3u) == 0u ? o.w[0] : (tt & 3u) == 1u ? o.w[1] : (tt & 3u) == 2u ? o.w[2] : o.w[3];
            }
            const int g32 = (int)(step & 1ull) * 32; // first lane of this step's candidates in the block
            bool fast = SMOLMC_TABLE_MULTI_FAST; // first the candidate block; after a miss the full scan
            for (bool done = false; !done; fast = false) {
            nfl = 0;
            fast_ok = fast;
            uint32_t fpos = 0; // block path: next stream position (kept across species and sublattices)
            uint32_t tpos = 0, round = 0, ow[4] = {0, 0, 0, 0};
            bool have_round = false;
            int qdraw = 0, dbase = 0;
            for (int sl = 0; sl < NS && (fast_ok || !fast); ++sl) {
                const int sb = sel4(P.m_sbase, sl), ncod = sel4(P.m_ncodes, sl);
                const uint32_t na = (uint32_t)sel4(P.m_nact, sl);
                int vcol = 0, vcsp = 0, ncol = 0; // sites collected in this sublattice and their (depleted) species, lane-indexed
                int cs[4] = {0, 0, 0, 0}, cv[4] = {0, 0, 0, 0};
                bool have_sites = false; // cs / cv valid for (round, sl)
                if (fast) {
                    // lane-parallel picks among the block's candidates (mc_table_kernel)
                    uint32_t dep = (uint32_t)__ballot(vu < 0) & (((1u << ncod) - 1u) << dbase);
                    if (dep != 0u) {
                        const int site_l = sb + (int)__umulhi(cb_word, na);
                        const int cvl = (int)occ[lean_swz(site_l, swa, swm, swb)];
                        int vdst = -1;
                        while (dep != 0u && fast_ok) {
                            const int d = __ffs((int)dep) - 1;
                            dep &= dep - 1u;
                            const int need = -(int)rdlane((uint32_t)vu, d);
                            uint32_t m = (uint32_t)(__ballot(cvl == d - dbase) >> g32);
                            m = fpos < 32u ? (m >> fpos) << fpos : 0u;
                            if (__popc(m) < need) { fast_ok = false; break; }
                            const unsigned long long m64 = (unsigned long long)m << g32;
                            const int rnk = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m64 >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)m64, 0u));
                            const bool mine = ((m64 >> lane) & 1ull) != 0ull && rnk < need;
                            if (mine) vdst = ncol + rnk;
                            if (lane >= ncol && lane < ncol + need) vcsp = d - dbase;
                            const unsigned long long lastm = __ballot(mine && rnk == need - 1);
                            fpos = (uint32_t)(__ffsll((long long)lastm) - 1 - g32) + 1u;
                            ncol += need;
                        }
                        if (fast_ok) {
                            int *scr = (int *)s_feat; // (per-wave scratch, free between sample rows)
                            if (vdst >= 0) scr[vdst] = site_l;
                            vcol = scr[lane];
                            bool dup = false; // a site named twice: leave the step to the full scan
#define SMOLMC_DUP_SHIFT(D, CTRL)                                                                  \
    if (ncol > D) {                                                                                \
        const int other = __builtin_amdgcn_update_dpp(0, vcol, CTRL, 0xf, 0xf, false);             \
        dup |= lane >= D && lane < ncol && vcol == other;                                          \
    }
                            SMOLMC_DUP_SHIFT(1, 0x111)
                            SMOLMC_DUP_SHIFT(2, 0x112)
                            SMOLMC_DUP_SHIFT(3, 0x113)
                            SMOLMC_DUP_SHIFT(4, 0x114)
                            SMOLMC_DUP_SHIFT(5, 0x115)
                            SMOLMC_DUP_SHIFT(6, 0x116)
                            SMOLMC_DUP_SHIFT(7, 0x117)
#undef SMOLMC_DUP_SHIFT
                            if (ncol > 8 || __ballot(dup) != 0ull) fast_ok = false;
                        }
                    }
                    if (!fast_ok) break;
                }
                for (int c = 0; c < ncod && !fast; ++c) {
                    int need = -(int)rdlane((uint32_t)vu, dbase + c);
                    unsigned long long B[4] = {0ull, 0ull, 0ull, 0ull};
                    bool have_masks = false;
                    while (need > 0) {
                        if (!have_round) {
                            const philox_out o = philox4x32_10((uint32_t)step, (uint32_t)(step >> 32),
                                                               4u + 64u * round + (uint32_t)lane, 0u, key0, key1);
                            ow[0] = o.w[0]; ow[1] = o.w[1]; ow[2] = o.w[2]; ow[3] = o.w[3];
                            have_round = true;
                            have_sites = false;
                            tpos = 0;
                        }
                        if (!have_sites) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                cs[j] = sb + (int)__umulhi(ow[j], na);
                                cv[j] = (int)occ[lean_swz(cs[j], swa, swm, swb)];
                            }
                            have_sites = true;
                            have_masks = false;
                        }
                        if (!have_masks) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) B[j] = __ballot(cv[j] == c);
                            have_masks = true;
                        }
                        uint32_t best = 0xffffffffu;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t l0 = tpos > (uint32_t)j ? (tpos - (uint32_t)j + 3u) >> 2 : 0u;
                            const unsigned long long m = l0 < 64u ? (B[j] >> l0) << l0 : 0ull;
                            if (m) best = min(best, 4u * (uint32_t)(__ffsll((long long)m) - 1) + (uint32_t)j);
                        }
                        if (best == 0xffffffffu) { round++; have_round = false; continue; }
                        tpos = best + 1u;
                        const int bl = (int)(best >> 2), bj = (int)(best & 3u);
                        const int picked = (int)rdlane((uint32_t)(bj == 0 ? cs[0] : bj == 1 ? cs[1] : bj == 2 ? cs[2] : cs[3]), bl);
                        if (__ballot(lane < ncol && vcol == picked) != 0ull) continue;
                        if (lane == ncol) { vcol = picked; vcsp = c; }
                        ncol++;
                        need--;
                    }
                }
                for (int c = 0; c < ncod; ++c) { // random assignment (:627-631)
                    const int u = (int)rdlane((uint32_t)vu, dbase + c);
                    for (int k = 0; k < u; ++k) {
                        const int wl = l4 + 2 + (qdraw >> 2);
                        const int wj = qdraw & 3;
                        const uint32_t word = rdlane(wj == 0 ? W0 : wj == 1 ? W1 : wj == 2 ? W2 : W3, wl);
                        qdraw++;
                        const int rr = (int)__umulhi(word, (uint32_t)ncol);
                        const int site = (int)rdlane((uint32_t)vcol, rr);
                        const int od = (int)rdlane((uint32_t)vcsp, rr); // (known from the scan: no LDS read)
                        if (lane == nfl) { vsite = site; vnew = c; vold = od; vfsub = sl; }
                        nfl++;
                        // list.remove keeps the order of the rest: lanes >= rr take their right neighbour
                        // (DPP wave shift, not a cross-lane LDS permute)
                        const int nxt = __builtin_amdgcn_update_dpp(0, vcol, 0x130, 0xf, 0xf, false);
                        const int nxs = __builtin_amdgcn_update_dpp(0, vcsp, 0x130, 0xf, 0xf, false);
                        if (lane >= rr) { vcol = nxt; vcsp = nxs; }
                        ncol--;
                    }
                }
                dbase += ncod;
            }
            done = !fast || fast_ok;
            } // (block attempt, then full scan)
        }
        } // (step-at-a-time proposal)
        if (dir >= 0) {
            if (__builtin_expect(!((lp_valid >> dir) & 1u), 0)) { // compute_log_priori_factor (mcusher.py:656-711), cached per direction
                // (all directions feasible now and after the step: the same weight sum, no mask to form)
                const double sum_next = (feas_now == (1u << nf2) - 1u && all_feasible(vcnt + vu)) ? sumw : masked_sum(feasible(vcnt + vu));
                double lf = 0.0;
                const double w_now = weight_of(dir), w_back = weight_of(dir ^ 1);
                if (!(w_now == w_back && sum_next == sumw)) {
                    const double tsw = rare_params()->tf_sw;
                    const double p_now = (1.0 - tsw) * w_now / sumw;
                    const double p_next = (1.0 - tsw) * w_back / sum_next;
                    lf = log(p_next / p_now);
                }
                lf += table_log_count_ratio(rare_params()->tf_ln, vu, vcnt, D);
                lf = uni_d(lf);
                if (lane == dir) vlp = lf;
                lp_valid |= 1u << dir;
            }
            log_priori = __hiloint2double((int)rdlane((uint32_t)__double2hiint(vlp), dir),
                                          (int)rdlane((uint32_t)__double2loint(vlp), dir));
        }
        if (REPLAY) { // a given a-priori factor replaces the kernel's own
            const LeanParamsKernarg Q = rare_params();
            const double given = Q->rp_lp ? uni_d(Q->rp_lp[(size_t)r * nsteps32 + it_step]) : __builtin_nan("");
            if (given == given) log_priori = nfl ? given : 0.0;
        }

#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); ph_acc[2] += tn - ph_t; ph_t = tn; }
#endif
        // -------- sequential evaluation of the flips of this step -----------------------
        double e = 0.0, ew_uni = 0.0, dMu = 0.0;
        double vdq = 0.0;
        double vG = 0.0;
        if (has_ew && nfl > 1) {
            const int pi = lane >> 3, pj = lane & 7;
            const int si = __shfl(vsite, pi), sj = __shfl(vsite, pj);
            if (pj < pi && pi < nfl) vG = P.ew_G[(size_t)si * P.ew_nact + (sj - abase)];
        }
        unsigned cmask = 0; // classes with pending feature deltas
        RowWords<NW> rows[4];
#pragma unroll
        for (int f = 0; f < 4; ++f)
            if (f < nfl) rows[f] = load_row<NW>(idx_rs, lane_voff, rdlane((uint32_t)vsite, f) * SITE_BYTES);
        auto eval_flip = [&](const int f, const RowWords<NW> &row) {
            const int s = (int)rdlane((uint32_t)vsite, f), nw = (int)rdlane((uint32_t)vnew, f);
            const int od = (int)rdlane((uint32_t)vold, f), sl = (int)rdlane((uint32_t)vfsub, f);
            const int cls = sel4(P.m_cls, sl);
            cmask |= 1u << cls;
            const uint32_t pair = (uint32_t)od * snt8 + (uint32_t)nw * nt8;
            const MultiRec *rec = s_rec + ((size_t)cls * NSLOT) * 64 + lane;
            double *pend = s_pend + ((size_t)cls * NSLOT) * 64 + lane;
#pragma unroll
            for (int it = 0; it < NSLOT; ++it) {
                const MultiRec rc = rec[it * 64];
                uint32_t a = rc.doff8;
#pragma unroll
                for (int m = 0; m < MM; ++m) a += __umul24(rc.st8[m], (uint32_t)occ[row_entry<NW>(row, it * MM + m)]);
                const double d = *(const double *)((const unsigned char *)s_dt + (a + pair));
                e = fma(rc.w, d, e);
                pend[it * 64] += d;
            }
            if (has_ew) {
                const double dq = s_q[sl * 8 + nw] - s_q[sl * 8 + od];
                double pot = phi_lds ? phi[s - abase]
                                     : __hip_atomic_load(&phi[s - abase], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int m = 0; m < f; ++m) {
                    const double dqm = __hiloint2double((int)rdlane((uint32_t)__double2hiint(vdq), m),
                                                        (int)rdlane((uint32_t)__double2loint(vdq), m));
                    const double gfm = __hiloint2double((int)rdlane((uint32_t)__double2hiint(vG), 8 * f + m),
                                                        (int)rdlane((uint32_t)__double2loint(vG), 8 * f + m));
                    pot = fma(dqm, gfm, pot);
                }
                ew_uni += 2.0 * dq * pot + (s_dg[sl * 8 + nw] - s_dg[sl * 8 + od]);
                if (lane == f) vdq = dq;
            }
            if (has_mu) dMu += s_mu[sl * 8 + nw] - s_mu[sl * 8 + od];
            occ[lean_swz(s, swa, swm, swb)] = (uint8_t)nw; // tentative (every lane, same byte)
        };
        // Steps of two or three flips on the two-group kernels in two phases (see mc_table_kernel): (A)
        // every flip's slot records, occupancy gathers, potential read and tentative write ISSUED back
        // to back -- the LDS executes in order, flip f + 1 sees flip f --, (B) table reads and
        // arithmetic; the deltas stay in registers until the decision (no pending cells in LDS).
        constexpr int TPF = NSLOT <= 2 ? 3 : 0; // flips the two-phase form covers (four: 256 VGPRs and scratch)
        double tpd[TPF ? TPF : 1][NSLOT];
        int tpcls[TPF ? TPF : 1];
        bool two_phased = false;
        auto two_phase = [&](auto nflips) {
            constexpr int NF = decltype(nflips)::value;
            int fsite[NF], fnw[NF], fod[NF], fsl[NF];
            uint32_t g[NF][NSLOT * MM];
            MultiRec rc[NF][NSLOT];
            double fpot[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                fsite[f] = (int)rdlane((uint32_t)vsite, f);
                fnw[f] = (int)rdlane((uint32_t)vnew, f);
                fod[f] = (int)rdlane((uint32_t)vold, f);
                fsl[f] = (int)rdlane((uint32_t)vfsub, f);
                tpcls[f] = sel4(P.m_cls, fsl[f]);
                const MultiRec *rec = s_rec + ((size_t)tpcls[f] * NSLOT) * 64 + lane;
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) rc[f][it] = rec[it * 64];
#pragma unroll
                for (int q = 0; q < NSLOT * MM; ++q) g[f][q] = (uint32_t)occ[row_entry<NW>(rows[f], q)];
                fpot[f] = !has_ew ? 0.0
                          : (phi_lds ? phi[fsite[f] - abase]
                                     : __hip_atomic_load(&phi[fsite[f] - abase], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                occ[lean_swz(fsite[f], swa, swm, swb)] = (uint8_t)fnw[f]; // tentative (every lane, same byte)
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int nw = fnw[f], od = fod[f], sl = fsl[f];
                const uint32_t pair = (uint32_t)od * snt8 + (uint32_t)nw * nt8;
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) {
                    uint32_t a = rc[f][it].doff8;
#pragma unroll
                    for (int m = 0; m < MM; ++m) a += __umul24(rc[f][it].st8[m], g[f][it * MM + m]);
                    const double d = *(const double *)((const unsigned char *)s_dt + (a + pair));
                    e = fma(rc[f][it].w, d, e);
                    tpd[f][it] = d;
                }
                if (has_ew) {
                    const double dq = s_q[sl * 8 + nw] - s_q[sl * 8 + od];
                    double pot = fpot[f];
#pragma unroll
                    for (int m = 0; m < f; ++m) {
                        const double dqm = __hiloint2double((int)rdlane((uint32_t)__double2hiint(vdq), m),
                                                            (int)rdlane((uint32_t)__double2loint(vdq), m));
                        const double gfm = __hiloint2double((int)rdlane((uint32_t)__double2hiint(vG), 8 * f + m),
                                                            (int)rdlane((uint32_t)__double2loint(vG), 8 * f + m));
                        pot = fma(dqm, gfm, pot);
                    }
                    ew_uni += 2.0 * dq * pot + (s_dg[sl * 8 + nw] - s_dg[sl * 8 + od]);
                    if (lane == f) vdq = dq;
                }
                if (has_mu) dMu += s_mu[sl * 8 + nw] - s_mu[sl * 8 + od];
            }
            two_phased = true;
        };
#ifndef SMOLMC_NO_MULTI_TWO_PHASE
        if (TPF && nfl == 2) two_phase(std::integral_constant<int, 2>{});
        else if (TPF && nfl == 3) two_phase(std::integral_constant<int, 3>{});
        else
#endif
        {
#pragma unroll
            for (int f = 0; f < 4; ++f)
                if (f < nfl) eval_flip(f, rows[f]);
            for (int f = 4; f < nfl; ++f)
                eval_flip(f, load_row<NW>(idx_rs, lane_voff, rdlane((uint32_t)vsite, f) * SITE_BYTES));
        }
#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); ph_acc[3] += tn - ph_t; ph_t = tn; }
#endif
        double dH = wave_sum_all(e);
        const double dEw = has_ew ? ew_uni : 0.0;
        if (has_ew) dH += P.ew_coef * dEw;
        if (has_mu) dH -= dMu;
        const double lu = REPLAY ? lu_rp
                                 : __hiloint2double((int)rdlane((uint32_t)__double2hiint(q_logu), l6),
                                                    (int)rdlane((uint32_t)__double2loint(q_logu), l6));
        bool accepted;
        int wnb = wb;
        double dB = 0.0, dQ[SMOLMC_MAX_BIAS_ROWS] = {0.0, 0.0, 0.0, 0.0};
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
            // compute_bias_change of the step against the occupancy before it (kernel/base.py:307-311, bias.py:75-93):
            // the flips of a table step touch distinct sites; lane f reads the pair entry of flip f
            if (BIAS && tb_type && nfl >= 1) {
                const LeanParamsKernarg Q = rare_params();
                const uint32_t pidx = lane < nfl ? (uint32_t)(vfsub * 64 + vold * 8 + vnew) : 0u;
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
            const double exponent = nbeta * dH + log_priori + dB; // metropolis.py:41-44
            accepted = __ballot((exponent >= 0.0) || (exponent > lu)) != 0ull;
        }
        if (BIAS && accepted) {
            tb_acc += dB;
#pragma unroll
            for (int k = 0; k < SMOLMC_MAX_BIAS_ROWS; ++k) tb_chg[k] += dQ[k];
        }
        nacc_before = nacc_add;
        if (WLT && accepted && wl_sum_mode) wl_flush_run(); // the state (bin, features) ends here: its post-steps go to the bin's row
        if (REPLAY) { // what smolmc_replay returns per step (the enthalpy follows the accepted changes)
            if (accepted) H_rp += dH;
            if (lane == 0) {
                const LeanParamsKernarg Q = rare_params();
                const size_t krec = (size_t)r * nsteps32 + it_step;
                Q->rp_acc[krec] = (uint8_t)(accepted ? 1 : 0);
                Q->rp_H[krec] = H_rp;
                if (Q->rp_lp_out) Q->rp_lp_out[krec] = log_priori;
            }
        }
        // pending feature deltas of the touched classes: keep (accept) or drop (reject)
        for (int cls = 0; cls < NC; ++cls)
            if ((cmask >> cls) & 1u) {
                double *pend = s_pend + ((size_t)cls * NSLOT) * 64 + lane;
                double *cell = s_acc + ((size_t)cls * NSLOT) * 64 + lane;
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) {
                    if (WLT) { // (the feature deltas of the step: into the shadow cells, see below)
                        if (accepted)
                            __hip_atomic_fetch_add(&s_feat[s_ft[((size_t)cls * NSLOT + it) * 64 + lane] + wl_shadow],
                                                   s_fs[((size_t)cls * NSLOT + it) * 64 + lane] * pend[it * 64], __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WAVEFRONT);
                    } else if (accepted) cell[it * 64] += pend[it * 64];
                    pend[it * 64] = 0.0;
                }
            }
        if (accepted && two_phased) { // feature deltas of the two-phase form: registers -> accumulator cells
#pragma unroll
            for (int f = 0; f < (TPF ? TPF : 1); ++f)
                if (f < nfl) {
                    double *cell = s_acc + ((size_t)tpcls[f] * NSLOT) * 64 + lane;
#pragma unroll
                    for (int it = 0; it < NSLOT; ++it) {
                        if (WLT)
                            __hip_atomic_fetch_add(&s_feat[s_ft[((size_t)tpcls[f] * NSLOT + it) * 64 + lane] + wl_shadow],
                                                   s_fs[((size_t)tpcls[f] * NSLOT + it) * 64 + lane] * tpd[f][it], __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WAVEFRONT);
                        else cell[it * 64] += tpd[f][it];
                    }
                }
        }
        if (WLT && accepted) { // _do_accept_step (wanglandau.py:204-220): features, enthalpy and bin follow the step
            double df = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) df += s_feat[k < wl_k ? wl_rd0 + k * wl_rdstep : 63];
            if (wl_zero_lane) s_feat[lane] = 0.0;
            if (has_ew) df += lane == P.Fce ? dEw : 0.0;
            if (has_mu) df += lane == P.Fce + (has_ew ? 1 : 0) ? dMu : 0.0;
            fcur += df;
            H += dH;
            wb = wnb;
        }
        if (accepted) {
            if (dir >= 0) {
                // the mask of the feasible directions is recomputed unless all were feasible and still are
                const bool was_all = all_feasible(vcnt);
                vcnt += vu;
                lp_valid = 0u;
                if (!(was_all && all_feasible(vcnt))) head_valid = false;
            }
            if (!REPLAY) {
                // batch lanes whose scan examined a site that has just changed are stale
                uint32_t hit = 0u;
                for (int f = 0; f < nfl; ++f) {
                    const uint32_t sf = rdlane((uint32_t)vsite, f);
                    const uint32_t pat = sf | (sf << 16);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const uint32_t dd = q_c[i] ^ pat; // a zero half-word <=> that kept site is sf
                        hit |= (dd - 0x00010001u) & ~dd & 0x80008000u;
                    }
                }
                q_stale |= __ballot(hit != 0u);
            }
            if (has_ew) field_apply_flips<0>(phi_lds ? phi : P.ew_phi + (size_t)r * P.ew_nact, lane, nfl, vsite, vdq);
            acc_mu += dMu;
            acc_ew += dEw;
            nacc_add++;
        } else {
            for (int f = nfl - 1; f >= 0; --f) { // undo the tentative flips
                const int s = (int)rdlane((uint32_t)vsite, f), od = (int)rdlane((uint32_t)vold, f);
                occ[lean_swz(s, swa, swm, swb)] = (uint8_t)od;
            }
        }
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

#ifdef SMOLMC_EXP_PHASES
        { const long long tn = clock64(); ph_acc[4] += tn - ph_t; ph_t = tn; }
#endif
        if (--smp_countdown == 0) { // (sampling parameters: re-read from the kernel arguments, see rare_params)
            const LeanParamsKernarg Q = rare_params();
            smp_countdown = (uint32_t)Q->smp.every;
            const size_t row = (size_t)smp_index * Q->R + r;
            smp_index++;
            const int qF = Q->F, qFce = Q->Fce;
            double *const q_feat = Q->smp.feat;
            double Hnow;
            if (WLT) {
                if (lane < qF) q_feat[row * qF + lane] = fcur;
                Hnow = H;
            } else {
            s_feat[lane] = 0.0;
            double lane_e = 0.0;
            const LeanSlot *q_slots = Q->slots;
            for (int i = lane; i < nrec; i += 64) {
                const LeanSlot sl = q_slots[i];
                const double v = s_acc[i];
                lane_e = fma(sl.w, v, lane_e);
                if (sl.live && v != 0.0)
                    __hip_atomic_fetch_add(&s_feat[sl.feat], sl.fs * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            if (lane < qFce) q_feat[row * qF + lane] = base_feat + s_feat[lane];
            if (has_ew && lane == qFce) q_feat[row * qF + lane] = base_feat + acc_ew;
            if (has_mu && lane == qFce + (has_ew ? 1 : 0)) q_feat[row * qF + lane] = base_feat + acc_mu;
            Hnow = H + (wave_sum_all(lane_e) - acc_mu + (has_ew ? Q->ew_coef * acc_ew : 0.0));
            }
            if (lane == 0) {
                Q->smp.H[row] = Hnow;
                Q->smp.acc[row] = (uint8_t)(nacc_add != nacc_before);
                if (BIAS && Q->smp_bias_off) (Q->smp.H + Q->smp_bias_off)[row] = Q->bias[r] + tb_acc; // trace.bias
            }
            if (Q->smp.occ) {
                const int qNpad = Q->Npad;
                uint32_t *dst = (uint32_t *)(Q->smp.occ + row * qNpad);
                for (int i = lane; i < qNpad / 4; i += 64)
                    dst[i] = *(const uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb));
            }
        }
    }

#ifdef SMOLMC_EXP_PHASES
    if (r == 0 && lane == 0)
        printf("multi phases (cycles per step): covered %.3f | skeleton %.0f | head %.0f | picks+assign/swap %.0f | eval %.0f | decide+update %.0f\n", (double)ph_cov / (double)P.steps,
               (double)ph_acc[0] / (double)P.steps, (double)ph_acc[1] / (double)P.steps, (double)ph_acc[2] / (double)P.steps,
               (double)ph_acc[3] / (double)P.steps, (double)ph_acc[4] / (double)P.steps);
#endif
    if (phi_lds)
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
            P.wl.counter[r] = wl_counter0 + (long long)nsteps32;
        }
    } else {
    s_feat[lane] = 0.0;
    double lane_e = 0.0;
    for (int i = lane; i < nrec; i += 64) {
        const LeanSlot sl = P.slots[i];
        const double v = s_acc[i];
        lane_e = fma(sl.w, v, lane_e);
        if (sl.live && v != 0.0)
            __hip_atomic_fetch_add(&s_feat[sl.feat], sl.fs * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    if (lane < P.Fce) featp[lane] = base_feat + s_feat[lane];
    H += wave_sum_all(lane_e) - acc_mu + (has_ew ? P.ew_coef * acc_ew : 0.0);
    }
    if (lane == 0) {
        if (has_ew && !WLT) featp[P.Fce] += acc_ew;
        if (has_mu && !WLT) featp[P.Fce + (has_ew ? 1 : 0)] += acc_mu;
        P.enthalpy[r] = H;
        P.nsteps[r] = step;
        P.nacc[r] += nacc_add;
        if (nsteps32) P.last_acc[r] = (uint8_t)(nacc_add != nacc_before);
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
template <int NSLOT, int MM, bool REPLAY = false, bool WLT = false, bool BIAS = false> static int launch_table_multi_inst(smolmc_handle *h, const LeanParams &lp) {
    const unsigned wpb = (unsigned)h->waves_per_block_lean;
    const unsigned grid = (unsigned)((h->R + wpb - 1) / wpb);
    auto kern = lp.ew_field == 1 ? mc_table_multi_kernel<NSLOT, MM, 1, REPLAY, WLT, BIAS>
                                 : (lp.ew_field == 2 ? mc_table_multi_kernel<NSLOT, MM, 2, REPLAY, WLT, BIAS> : mc_table_multi_kernel<NSLOT, MM, 0, REPLAY, WLT, BIAS>);
    if (h->lean_lds > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h->lean_lds));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wpb), h->lean_lds, h->stream, lp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return 0;
}

// (instantiated in multi_table_bias_n*.hip only)
template <int NSLOT> static int launch_table_multi_bias_nslot(smolmc_handle *h, const LeanParams &lp) {
    return h->lean_mm == 2 ? launch_table_multi_inst<NSLOT, 2, false, false, true>(h, lp) : launch_table_multi_inst<NSLOT, 3, false, false, true>(h, lp);
}
// (instantiated in multi_table_wl_n*.hip only)
template <int NSLOT> static int launch_table_multi_wl_nslot(smolmc_handle *h, const LeanParams &lp) {
    return h->lean_mm == 2 ? launch_table_multi_inst<NSLOT, 2, false, true>(h, lp) : launch_table_multi_inst<NSLOT, 3, false, true>(h, lp);
}
// (instantiated in multi_table_replay_n*.hip only)
template <int NSLOT> static int launch_table_multi_replay_nslot(smolmc_handle *h, const LeanParams &lp) {
    return h->lean_mm == 2 ? launch_table_multi_inst<NSLOT, 2, true>(h, lp) : launch_table_multi_inst<NSLOT, 3, true>(h, lp);
}
template <int NSLOT> static int launch_multi_nslot(smolmc_handle *h, const LeanParams &lp) {
    if (h->cfg.step_type == SMOLMC_STEP_TABLE_FLIP)
        return h->lean_mm == 2 ? launch_table_multi_inst<NSLOT, 2>(h, lp) : launch_table_multi_inst<NSLOT, 3>(h, lp);
    return h->lean_mm == 2 ? launch_multi_nm<NSLOT, 2>(h, lp) : launch_multi_nm<NSLOT, 3>(h, lp);
}
