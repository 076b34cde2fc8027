// mc_wl.h -- lean Wang-Landau kernel (smol/moca/kernel/wanglandau.py:175-266) for the model class of
// mc_lean_kernel: one site class, one contiguous active sublattice, cluster-interaction (or K = 1
// correlation) features, no mu / Ewald term, update_period 1.  BASELINE config 4 runs 1024 such
// walkers per GPU = ONE wave per SIMD, so a step lasts as long as its chain of dependent
// operations; this kernel exists to keep the Wang-Landau bookkeeping off that chain (round 2 ran it
// as a variant of mc_lean_kernel: 1990 cycles per step against 950 for a Metropolis step).
//
//  * Bin of the proposed enthalpy from a float32 wave sum (8 VALU) instead of the float64 butterfly
//    (22 dependent VALU) + exact floor division: H is carried as an approximation with a RIGOROUS
//    error bound (float32-sum bound P.fast_eps per accepted step since the last exact rebuild); when
//    the proposed enthalpy lies further from every bin edge and window end than that bound, the
//    bin -- all the accept test needs, wanglandau.py:186-202 -- is the exact one.  Otherwise, and
//    every time the bound has grown to WL_RESYNC_FRAC of a bin, the step takes the exact path:
//    float64 reductions of the step's delta and of the per-slot accumulators (the exact current
//    enthalpy), exact floor division.  ~1 % of the steps.
//  * The entropies of the bins around the current one are read at the top of the step (the bin
//    moves by at most one on ~95 % of the accepted steps), so the accept test waits for no LDS read.
//  * Per-bin feature sums (wanglandau.py:235-239: mean = sum / occurrences, formed when read) without
//    a current feature vector in LDS: the features are linear in the per-slot accumulators
//    acc (sums of accepted deltas since the launch started), so the sum over a RUN of consecutive
//    steps in one bin is n f0 + sum_lanes,slots fs * A with A = sum over the run of acc -- two
//    float64 adds per step in registers.  Only when the bin changes (30 % of config 4's steps)
//    the run is flushed: LDS atomics into shadow copies (see below), read back and added to the
//    bin's row.  Round 2 paid two 6-way serialised LDS atomics on every accepted step (82 %) and
//    eight LDS reads on every step for the same sums.
//  * No global-memory update in the step loop.  Round 2 (and the first version of this kernel)
//    added the sums to the bin's row in HBM with one global atomic per step.  vmcnt counts in
//    order, so the next step's wait for its index row also waits for the atomic's acknowledgement
//    from L2 -- ~1900 cycles, i.e. the whole step: with everything else gone the kernel ran exactly
//    as fast as before (16.45 against 16.57 ms per 2e4 steps).  The rows now live in a direct-mapped
//    LDS cache of WL_ROWS bins (slot = bin mod WL_ROWS, tags in the lanes of one VGPR); a walker's
//    enthalpy diffuses, so a row is evicted (the only global atomic left, in an out-of-line
//    function) once per several hundred steps, and all of them when the launch ends.
#pragma once
#include "mc_lean.h"

#define WL_ROWS SMOLMC_WL_ROWS // per-bin feature-sum rows cached in LDS (a power of two <= 64)
#ifndef WL_RESYNC_FRAC
#define WL_RESYNC_FRAC 0.005 // the carried error bound, in bins, at which the enthalpy is rebuilt exactly
#endif

// LDS access by 32-bit address (pointer arithmetic on generic pointers is done in 64 bits)
#if defined(__HIP_DEVICE_COMPILE__)
#define WL_LDS_F64P(a) ((lds_f64_t *)(a))
#define WL_LDS_I64P(a) ((__attribute__((address_space(3))) long long *)(a))
#define WL_LDS_U32P(a) ((__attribute__((address_space(3))) uint32_t *)(a))
#else
#define WL_LDS_F64P(a) ((double *)(uintptr_t)(a))
#define WL_LDS_I64P(a) ((long long *)(uintptr_t)(a))
#define WL_LDS_U32P(a) ((uint32_t *)(uintptr_t)(a))
#endif

// Eviction of one cached row of per-bin feature sums: added to the bin's row in HBM (the rows of
// a walker are touched by its own wave only; the atomic is used for its fire-and-forget form) and
// cleared.  Out of line: a global-memory instruction that exists on some paths of the step loop
// only would make the compiler's s_waitcnt insertion conservative on every path (NOTES.md).
__device__ __noinline__ void wl_evict_row(double *grow, double *crow, int F, int lane) {
    if (lane < F) {
        unsafeAtomicAdd(grow + lane, crow[lane]);
        crow[lane] = 0.0;
    }
}

// REPLAY: proposals and uniforms from the host in the reference's draw order (see mc_lean_kernel).
// HAS_MU: semigrand Wang-Landau -- one chemical-potential row on the active sublattice; the chemical
// work joins the enthalpy (ensemble.py:368-374) and is a feature of its own, carried as a uniform
// scalar beside the per-slot accumulators (its run-integrated sum goes to the bin rows like theirs).
// EWF: Ewald term from the walker's potential field in LDS: O(1) per proposal, one field
// sweep per accepted step; these steps take the exact float64 decision (the Ewald delta is of the
// order of eV and would have to be carried through the float32 error bound).
template <int NSLOT, int MM, int STEP, bool REPLAY = false, bool HAS_MU = false, bool EWF = false>
__global__ void __launch_bounds__(256) mc_wl_kernel(const LeanParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = uni((int)(threadIdx.x >> 6));
    const int nwaves = blockDim.x >> 6;
    const int slot = uni(blockIdx.x * nwaves + wave);
    // (group rotation, see launch_wl_kern: the walker of a launch slot)
    const int r = P.launch_slots ? ((P.rot_j + slot / P.rot_s) % P.rot_g) * P.rot_s + slot % P.rot_s : slot;
    const size_t per_wave = (size_t)P.Nlds + 64 * 8 + 64 + wl_lean_bins_bytes(P.wl.L) + (size_t)WL_ROWS * P.F * 8 +
                            (EWF ? (size_t)P.ew_nact * 8 : 0);
    double *s_dt = (double *)smem;
    double *s_mu = s_dt + P.dt_len, *s_q = s_mu + 8, *s_dg = s_mu + 16; // [8] each: mu / charge / diagonal term per code
    unsigned char *wbase = (unsigned char *)(s_dt + P.dt_len + 24) + (size_t)wave * per_wave;
    uint8_t *occ = wbase; // indexed by SWIZZLED site address
    double *s_cell = (double *)(wbase + P.Nlds); // 64 doubles: shadow copies of a run's feature sums
    // Per-bin state (round 5): the entropies S [L] as float64 -- with one guard entry before bin 0 and one behind
    // bin L-1 (read, never selected) -- and ONE uint32 count per bin: the steps spent there since the launch started
    // / the last successful flatness check.  Histogram and occurrences gain one together on every step
    // (wanglandau.py:241-245, update_period 1), so the HBM arrays keep the base values and take the count when the
    // histogram is reset and when the launch ends: 12 bytes per bin instead of round 3's 24-byte records
    // {entropy, histogram, occurrences} -- config 4 fits three workgroups per CU instead of two, and the post-step is
    // two LDS atomics instead of three.
    double *wl_S = s_cell + 64 + 1;
    const uint32_t rec0 = (uint32_t)(uintptr_t)wl_S; // LDS byte address of bin 0's entropy
    uint32_t *wl_cnt = (uint32_t *)(wl_S + P.wl.L + 1);
    const uint32_t cnt0 = (uint32_t)(uintptr_t)wl_cnt;
    double *s_rows = (double *)((unsigned char *)(s_cell + 64) + wl_lean_bins_bytes(P.wl.L)); // cached rows of per-bin feature sums [WL_ROWS][F]
    double *phi = s_rows + (size_t)WL_ROWS * P.F;       // EWF: Ewald potential field [ew_nact]
    const int swa = P.swz_a, swm = P.swz_m, swb = P.swz_b;
    for (int i = threadIdx.x; i < P.dt_len; i += blockDim.x) s_dt[i] = P.dt[i];
    if (threadIdx.x < 8) {
        s_mu[threadIdx.x] = (HAS_MU && threadIdx.x < P.ncodes) ? P.mu_row[threadIdx.x] : 0.0;
        s_q[threadIdx.x] = EWF ? P.ew_qrow[threadIdx.x] : 0.0;
        s_dg[threadIdx.x] = EWF ? P.ew_dgrow[threadIdx.x] : 0.0;
    }
    const bool live = P.launch_slots ? slot < P.launch_slots : r < P.R;
    if (live) {
        const uint32_t *src = (const uint32_t *)(P.occ + (size_t)r * P.Npad);
        for (int i = lane; i < P.Npad / 4; i += 64)
            *(uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb)) = src[i];
        s_cell[lane] = 0.0;
        for (int i = lane; i < P.wl.L; i += 64) {
            wl_S[i] = P.wl.entropy[(size_t)r * P.wl.L + i];
            wl_cnt[i] = 0u;
        }
        if (lane < 2) wl_S[lane ? P.wl.L : -1] = 0.0;
        for (int i = lane; i < WL_ROWS * P.F; i += 64) s_rows[i] = 0.0;
        if (EWF)
            for (int j = lane; j < P.ew_nact; j += 64) phi[j] = P.ew_phi[(size_t)r * P.ew_nact + j];
    }
    __syncthreads();
    if (!live) return;
    // uniform feature scalars: accepted chemical work / Ewald energy since the launch started, and their
    // sums over the steps of the current run (what the per-slot acc / run are for the cluster features)
    double acc_mu = 0.0, acc_ew = 0.0, run_mu = 0.0, run_ew = 0.0;
    const int f_ew = P.Fce, f_mu = P.Fce + (EWF ? 1 : 0);
    const double ew_coef = EWF ? P.ew_coef : 0.0;

    // per-lane slot constants (registers for the whole launch)
    uint32_t doff8[NSLOT], st8[NSLOT][MM], sfeat[NSLOT];
    double wgt[NSLOT], sfs[NSLOT];
    double acc[NSLOT]; // sum of the accepted steps' deltas since the launch started
    double run[NSLOT]; // sum of acc over the steps of the current run (consecutive steps in one bin)
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
        run[it] = 0.0;
    }
    // Shadow copies: lanes of one orbit add into the same feature cell, and the LDS serialises a
    // ds_add_f64 per address (~40 lanes per cell on the headline model).  The 64 doubles of s_cell
    // hold WLK = min(8, 63 / F) copies of the F cells back to back; lane l adds into copy l % WLK
    // and a reader sums the copies.  Cell 63 is never written: the address of "no copy".
    const int wl_stride = P.F;
    const int wl_k = max(1, min(8, 63 / max(wl_stride, 1)));
    uint32_t scell[NSLOT];
#pragma unroll
    for (int it = 0; it < NSLOT; ++it) scell[it] = sfeat[it] + (uint32_t)((lane % wl_k) * wl_stride);
    int wl_rd[8]; // per-lane read addresses of the copies of feature `lane` (doubles)
#pragma unroll
    for (int k = 0; k < 8; ++k) wl_rd[k] = (k < wl_k && lane < wl_stride) ? lane + k * wl_stride : 63;
    const bool zero_lane = lane < wl_k * wl_stride;

    double *featp = P.features + (size_t)r * P.F;
    const double f0 = lane < P.F ? featp[lane] : 0.0; // feature vector at launch start
    const double H0 = P.enthalpy[r];
    const double vmin = P.wl.vmin, span = P.wl.vmax - P.wl.vmin, bin = P.wl.bin;
    const double inv_bin = 1.0 / bin;
    const int Lm1 = P.wl.L - 1;
    double wl_m = P.wl.m[r];
    // current bin (walkers start inside the window: smolmc_set_state refuses anything else);
    // uniform values are kept visibly uniform (readfirstlane / ballot) so that the bookkeeping
    // below compiles to scalar code and scalar branches
    int b = uni((int)floordiv_exact(H0 - vmin, bin));
    b = min(max(b, 0), Lm1);
    // Carried enthalpy relative to the window start (float64: gains the float32 sum of every accepted
    // step) and its position in bins as a float32.  The pre-test works in float32 bins against ONE
    // fixed tolerance tolb: the bound on the carried enthalpy at its largest (resync_after accepted
    // steps of e1 each since the last exact rebuild, then an exact step is forced), the float32 sum
    // of the step being tested, and the float32 roundings of the position itself -- the position
    // (< L) and each of the two thresholds it is compared with are rounded once, 2^-24 L each, and
    // the multiply-add once more.
    double hoff = H0 - vmin;
    const double e1 = P.fast_eps + 1e-13;     // float32 sum of one step + the float64 add
    const double e1b = e1 * inv_bin, tol0 = e1b + 1e-9 + ((double)P.wl.L + 4.0) * ldexp(1.0, -22);
    const double span_b = span * inv_bin;
    const uint32_t resync_after = (uint32_t)uni((int)fmin(1.0e9, fmax(1.0, (WL_RESYNC_FRAC - tol0) / e1b)));
    const double tolb = tol0 + (double)resync_after * e1b;
    const float invbin32 = uni_f((float)inv_bin);
    const float tol32 = uni_f((float)tolb * 1.000001f), omt32 = uni_f(1.0f - tol32);
    const float hi32 = uni_f((float)span_b - tol32), hiout32 = uni_f((float)span_b + tol32);
    float xb32 = (float)(hoff * inv_bin);
    uint32_t since_sync = 0;
    const bool never_fast = __ballot(!(P.fast_eps > 0.0)) != 0ull; // SMOLMC_NO_FAST_ACCEPT: every step exact
    bool force_exact = never_fast;
    const long long wl_counter0 = P.wl.counter[r];
    // counter modulo the check period (the host refuses periods >= 2^31)
    const uint32_t wl_check = (uint32_t)P.wl.check;
    // (check period 0 = no device-side check: the remainder starts at 1 and cannot wrap to 0 inside a launch of < 2^30 steps)
    uint32_t wl_rem_check = wl_check ? (uint32_t)uni((int)(wl_counter0 % P.wl.check)) : 1u;
    unsigned long long step = P.nsteps[r];
    uint32_t nacc_add = 0, nacc_before = 0;
    const uint32_t key0_ = (uint32_t)P.seeds[r], key1_ = (uint32_t)(P.seeds[r] >> 32);
    // (the keys are made opaque at every Philox call: the compiler otherwise parks the twenty
    // loop-invariant round keys in SGPRs across the step loop, which then spills)
#define key0 opaque_u32(key0_)
#define key1 opaque_u32(key1_)
    const uint32_t nact = (uint32_t)P.nact, nt8 = P.nt8, snt8 = P.snt8;
    const int sbase = P.sbase;
    uint32_t smp_countdown = P.smp.every ? (uint32_t)P.smp.every : 0xffffffffu;
    long long smp_index = 0;
    uint32_t W0 = 0;
    int nsite = 0, naddr = 0;
    int cand[4] = {0, 0, 0, 0}, canda[4] = {0, 0, 0, 0};
    double logu = 0.0;
    unsigned long long batch64_base = ~0ull, batch_base = ~0ull;
    constexpr int ROW = NSLOT * MM;
    constexpr int NW = ROW / 2;
    constexpr uint32_t SITE_BYTES = 64u * ROW * 2u;
    const __amdgpu_buffer_rsrc_t idx_rs = __builtin_amdgcn_make_buffer_rsrc((void *)P.idx, 0, 0x7fffffff, 0x00020000);
    const uint32_t lane_voff = (uint32_t)lane * (ROW * 2u);

    // pending flush of a finished run: the shadow cells as they were read (summed in the next step,
    // so that nobody waits for the LDS atomics), the number of steps of the run and its bin
    double pend_c[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double pend_n = 0.0;
    int pend_bin = 0;
    bool pending = false;
    uint32_t run_n = 0;  // steps in the current run
    int vtag = -1;       // lane i < WL_ROWS: the bin cached in row i (-1: none)
    const uint32_t rowF = (uint32_t)P.F;
    // add the pending run's sums to its bin's cached row (uniform branch: nothing is pending on
    // ~70 % of the steps); a row that holds another bin is evicted first (rare)
    auto pend_commit = [&]() {
        if (!pending) return;
        pending = false;
        const int slot = pend_bin & (WL_ROWS - 1);
        const int tag = (int)rdlane((uint32_t)vtag, slot);
        double *crow = s_rows + (uint32_t)slot * rowF;
        if (tag != pend_bin) {
            if (tag >= 0) {
                const LeanParamsKernarg Q = rare_params();
                wl_evict_row(Q->wl.meanf + ((size_t)r * Q->wl.L + tag) * Q->F, crow, Q->F, lane);
            }
            vtag = lane == slot ? pend_bin : vtag;
        }
        const double s = ((pend_c[0] + pend_c[1]) + (pend_c[2] + pend_c[3])) + ((pend_c[4] + pend_c[5]) + (pend_c[6] + pend_c[7]));
        if (lane < (int)rowF) crow[lane] += fma(pend_n, f0, s);
    };
    // finish the current run: its sums go to the shadow cells and are read back into pend_c
    auto flush_run = [&](const int bin_of_run) {
#pragma unroll
        for (int it = 0; it < NSLOT; ++it) {
            __hip_atomic_fetch_add(&s_cell[scell[it]], sfs[it] * run[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            run[it] = 0.0;
        }
        if ((HAS_MU || EWF) && lane == 0) { // (copy 0 of the scalar features' cells)
            if (EWF) __hip_atomic_fetch_add(&s_cell[f_ew], run_ew, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (HAS_MU) __hip_atomic_fetch_add(&s_cell[f_mu], run_mu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        run_mu = 0.0;
        run_ew = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) pend_c[k] = s_cell[wl_rd[k]];
        if (zero_lane) s_cell[lane] = 0.0;
        pend_n = (double)run_n;
        pending = true;
        run_n = 0;
        pend_bin = bin_of_run;
    };
    auto exact_enthalpy = [&]() -> double { // H0 + sum over lanes and slots of w * acc (+ Ewald, - chemical work)
        double le = 0.0;
#pragma unroll
        for (int it = 0; it < NSLOT; ++it) le = fma(wgt[it], acc[it], le);
        return H0 + (wave_sum_all(le) + ew_coef * acc_ew - acc_mu);
    };
    // current features through the (zero) shadow cells: f0 + sum over lanes and slots of fs * acc, + the scalars
    auto current_features = [&]() -> double {
#pragma unroll
        for (int it = 0; it < NSLOT; ++it)
            __hip_atomic_fetch_add(&s_cell[scell[it]], sfs[it] * acc[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if ((HAS_MU || EWF) && lane == 0) {
            if (EWF) __hip_atomic_fetch_add(&s_cell[f_ew], acc_ew, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (HAS_MU) __hip_atomic_fetch_add(&s_cell[f_mu], acc_mu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        double fcur = f0;
#pragma unroll
        for (int k = 0; k < 8; ++k) fcur += s_cell[wl_rd[k]];
        if (zero_lane) s_cell[lane] = 0.0;
        return fcur;
    };

    int s1, a1;
    RowWords<NW> row1;
    uint32_t ridx = 0; // replay: record of this step
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
            const uint32_t w = (uint32_t)uni((int)philox4x32_10((uint32_t)sp, (uint32_t)(sp >> 32), 0u, 0u, key0, key1).w[1]);
            s1 = sbase + (int)__umulhi(w, nact);
        }
        a1 = lean_swz(s1, swa, swm, swb);
        row1 = load_row<NW>(idx_rs, lane_voff, (uint32_t)s1 * SITE_BYTES);
    }

#ifdef SMOLMC_EXP_PHASES // experiment: shader cycles per phase of a step (walker 0 prints the averages)
    long long lph[6] = {0, 0, 0, 0, 0, 0};
    long long lph_t = clock64();
#define WL_PHASE(i) { const long long tn = clock64(); lph[i] += tn - lph_t; lph_t = tn; }
#else
#define WL_PHASE(i)
#endif
    uint32_t steps_left = (uint32_t)P.steps; // the host splits launches at 2^30 steps
    while (steps_left != 0u) {
        // -------- random words (generated 16 steps at a time, see mc_lean_kernel) --------
        const unsigned long long base = step & ~15ull;
        if (!REPLAY && base != batch_base) {
            batch_base = base;
            const unsigned long long st = base + (unsigned)(lane >> 2);
            const philox_out o = philox4x32_10((uint32_t)st, (uint32_t)(st >> 32), (uint32_t)(lane & 3), 0u, key0, key1);
            W0 = o.w[0];
            nsite = sbase + (int)__umulhi(o.w[1], nact);
            naddr = lean_swz(nsite, swa, swm, swb);
            if ((step & ~63ull) != batch64_base) {
                batch64_base = step & ~63ull;
                const unsigned long long s64 = batch64_base + (unsigned)lane;
                const philox_out a = philox4x32_10((uint32_t)s64, (uint32_t)(s64 >> 32), 0u, 0u, key0, key1);
                logu = log(philox_u53(a.w[2], a.w[3]));
            }
            if (STEP == SMOLMC_STEP_SWAP) {
                cand[0] = sbase + (int)__umulhi(o.w[0], nact);
                cand[1] = sbase + (int)__umulhi(o.w[1], nact);
                cand[2] = sbase + (int)__umulhi(o.w[2], nact);
                cand[3] = sbase + (int)__umulhi(o.w[3], nact);
#pragma unroll
                for (int j = 0; j < 4; ++j) canda[j] = lean_swz(cand[j], swa, swm, swb);
            }
        }
        uint32_t chunk = 16u - (uint32_t)(step & 15ull);
        chunk = min(chunk, steps_left);
        chunk = min(chunk, smp_countdown);
        steps_left -= chunk;
        smp_countdown -= chunk;
        const unsigned long long chunk_end = step + chunk;
        int l4 = (int)(step & 15ull) * 4;
        int l64 = (int)(step & 63ull);
        do {
            __builtin_amdgcn_s_setprio(1);
            WL_PHASE(0)
            int s1n, a1n;
            int rq1 = 0, rq2 = -1, rq3 = 0;
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
            } else {
                s1n = (int)rdlane((uint32_t)nsite, l4);
                a1n = (int)rdlane((uint32_t)naddr, l4);
            }
            uint32_t va1 = (uint32_t)a1;
            asm volatile("" : "+v"(va1));
            const int o1 = uni((int)occ[va1]);
            // entropies around the current bin (the lane-0 atomic of the previous post-step is older
            // in the LDS queue: these reads see it)
            // (lane l holds S[b - 8 + l], clamped into the guard records: one LDS read serves the
            // accept test through v_readlane for |new bin - bin| <= 8)
            const uint32_t widx1 = (uint32_t)min(max(b - 7 + lane, 0), Lm1 + 2); // record index + 1
            const double win = *WL_LDS_F64P(rec0 - 8u + 8u * widx1);
            int nfl, s2, a2, n1, n2 = 0, o2 = 0;
            if (STEP != SMOLMC_STEP_SWAP) { s2 = s1; a2 = a1; }
            if (REPLAY) { // the recorded proposal (a swap kernel takes proper swaps only)
                if (STEP == SMOLMC_STEP_FLIP) {
                    nfl = rp_empty ? 0 : 1;
                    n1 = rp_empty ? o1 : rq1;
                    rp_bad |= (rq2 >= 0) ? 1 : 0;
                } else if (!rp_empty && rq2 >= 0) {
                    nfl = 2;
                    s2 = rq2;
                    a2 = lean_swz(s2, swa, swm, swb);
                    o2 = uni((int)occ[a2]);
                    rp_bad |= (rq3 != o1 || rq1 != o2) ? 1 : 0;
                    n2 = o1; n1 = o2;
                } else {
                    nfl = 0; s2 = s1; a2 = a1; o2 = o1; n2 = o1; n1 = o1;
                    rp_bad |= (!rp_empty || rq2 >= 0) ? 1 : 0;
                }
            } else if (STEP == SMOLMC_STEP_FLIP) { // Flip.propose_step (mcusher.py:154-170)
                const uint32_t kk = __umulhi(rdlane(W0, l4 + 1), (uint32_t)(P.ncodes - 1));
                n1 = (int)kk + ((int)kk >= o1 ? 1 : 0);
                nfl = 1;
            } else { // Swap.propose_step (mcusher.py:176-200), see mc_lean_kernel
                nfl = 2;
                n2 = o1;
#define SMOLMC_CAND_MASK(J)                                                                        \
    const int v##J = (int)occ[canda[J]];                                                           \
    const unsigned long long m##J = __ballot(v##J != o1) & (0xEull << l4);
#define SMOLMC_CAND_TAKE(J)                                                                        \
    {                                                                                              \
        const int bb = __ffsll((long long)m##J) - 1;                                               \
        s2 = (int)rdlane((uint32_t)cand[J], bb);                                                   \
        a2 = (int)rdlane((uint32_t)canda[J], bb);                                                  \
        o2 = (int)rdlane((uint32_t)v##J, bb);                                                      \
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
                                const unsigned long long cur = chunk_end - chunk;
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
                                        const int bb = __ffsll((long long)m) - 1;
                                        s2 = (int)rdlane((uint32_t)selsite, bb);
                                        a2 = lean_swz(s2, swa, swm, swb);
                                        o2 = (int)rdlane((uint32_t)selv, bb);
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
                                if (!hit) { nfl = 0; s2 = s1; a2 = a1; o2 = o1; }
                            }
                        }
                    }
                }
#undef SMOLMC_CAND_MASK
#undef SMOLMC_CAND_TAKE
                n1 = o2;
            }
            (void)nfl;
            // chemical work and Ewald delta of the proposal (uniform; ensemble.py:368-374, ewald.pyx:38-58
            // through the potential field: flip 2 of a swap sees flip 1 through the cross term G[s2][s1])
            double dMu = 0.0, dEw = 0.0, dq1 = 0.0, dq2 = 0.0;
            if (HAS_MU && nfl >= 1) {
                dMu = s_mu[n1] - s_mu[o1];
                if (STEP == SMOLMC_STEP_SWAP && nfl == 2) dMu += s_mu[n2] - s_mu[o2];
            }
            if (EWF && nfl >= 1) {
                dq1 = s_q[n1] - s_q[o1];
                dEw = 2.0 * dq1 * phi[s1 - sbase] + (s_dg[n1] - s_dg[o1]);
                if (STEP == SMOLMC_STEP_SWAP && nfl == 2) {
                    dq2 = s_q[n2] - s_q[o2];
                    const double cross = s2 != s1 ? P.ew_G[(size_t)s2 * P.ew_nact + (s1 - sbase)] : 0.0;
                    dEw += 2.0 * dq2 * (phi[s2 - sbase] + dq1 * cross) + (s_dg[n2] - s_dg[o2]);
                }
                dEw = uni_d(dEw);
            }
            __builtin_amdgcn_s_setprio(2);
            WL_PHASE(1)
            RowWords<NW> row2 = row1;
            if (STEP == SMOLMC_STEP_SWAP) row2 = load_row<NW>(idx_rs, lane_voff, (uint32_t)s2 * SITE_BYTES);
            pend_commit(); // the run that ended with the previous step (its LDS reads have landed by now)

            // -------- enthalpy delta (swap: both flips read one (old, new) block, see mc_lean_kernel)
            constexpr bool DIFF = STEP == SMOLMC_STEP_SWAP;
            double e = 0.0, d1[NSLOT];
            uint32_t dp[NSLOT];
            {
                const uint32_t pair1 = (uint32_t)o1 * snt8 + (uint32_t)n1 * nt8;
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) {
                    uint32_t a = dp[it] = doff8[it] + pair1;
#pragma unroll
                    for (int m = 0; m < MM; ++m) a += __umul24(st8[it][m], (uint32_t)occ[bounded(row_entry<NW>(row1, it * MM + m), (uint32_t)P.Nlds)]);
                    d1[it] = *(const double *)((const unsigned char *)s_dt + a);
                    if (!DIFF) e = fma(wgt[it], d1[it], e);
                }
            }
            row1 = load_row<NW>(idx_rs, lane_voff, (uint32_t)s1n * SITE_BYTES);
            if (STEP == SMOLMC_STEP_SWAP) {
                occ[va1] = (uint8_t)n1; // tentative first flip (undone on rejection)
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) {
                    uint32_t a = dp[it];
#pragma unroll
                    for (int m = 0; m < MM; ++m) a += __umul24(st8[it][m], (uint32_t)occ[bounded(row_entry<NW>(row2, it * MM + m), (uint32_t)P.Nlds)]);
                    d1[it] -= *(const double *)((const unsigned char *)s_dt + a); // D[(o2,n2)] = -D[(o1,n1)]
                    e = fma(wgt[it], d1[it], e);
                }
            }
            __builtin_amdgcn_s_setprio(3);
            WL_PHASE(2)
            // -------- WangLandau._accept_step (wanglandau.py:186-202) --------
            const double lu = REPLAY ? lu_rp
                                     : __hiloint2double((int)rdlane((uint32_t)__double2hiint(logu), l64),
                                                        (int)rdlane((uint32_t)__double2loint(logu), l64));
            int nb = b;
            bool accepted = false, decided = false;
            double dHa = 0.0; // what hoff gains when the step is accepted
#ifdef WL_EXP_NODECIDE
            {
                const float S32 = wave_sum_f32_uniform((float)e);
                dHa = (double)S32;
                accepted = __ballot(dHa * 40.0 < -lu) != 0ull; // (some cheap test with a similar acceptance)
                decided = true;
                nb = b + (accepted ? (int)(l64 & 1) : 0) - (accepted ? (int)((l64 >> 1) & 1) : 0);
                nb = min(max(nb, 0), Lm1);
            }
#endif
            if (!EWF && !force_exact && !decided) {
                const float S32 = wave_sum_f32_uniform((float)((HAS_MU && lane == 0) ? e - dMu : e));
                dHa = (double)S32;
                // proposed enthalpy in bins from the window start; the bin is certain when the
                // fractional part is further from 0 and 1 than the tolerance (and the value inside
                // the window by the same margin)
                const float x = __builtin_fmaf(S32, invbin32, xb32), fl = __builtin_floorf(x), fr = x - fl;
                const bool inside = (fr > tol32) & (fr < omt32) & (x >= 0.0f) & (x < hi32);
                if (__ballot(inside) != 0ull) {
                    nb = uni((int)fl);
                    const int dl = nb - b + 8;
                    double Snb;
                    if ((uint32_t)dl <= 16u)
                        Snb = __hiloint2double((int)rdlane((uint32_t)__double2hiint(win), dl),
                                               (int)rdlane((uint32_t)__double2loint(win), dl));
                    else
                        Snb = *WL_LDS_F64P(rec0 + 8u * (uint32_t)nb);
                    const double ex = win - Snb + 0.0; // lane 8: S[bin] - S[new bin]
                    accepted = ((__ballot((ex >= 0.0) | (ex > lu)) >> 8) & 1ull) != 0ull;
                    decided = true;
                } else if (__ballot((x < -tol32) | (x > hiout32)) != 0ull) {
                    decided = true; // new_h outside [min, max): rejected
                }
            }
            if (!decided) { // exact: float64 delta, exact current enthalpy, exact floor division
                const double dH = wave_sum_all(e) + ew_coef * dEw - dMu;
                const double Hx = exact_enthalpy();
                const double new_h = Hx + dH;
                hoff = Hx - vmin;
                since_sync = 0;
                force_exact = never_fast;
                dHa = dH;
                if (__ballot(!(new_h < vmin || new_h >= P.wl.vmax)) != 0ull) {
                    nb = uni((int)floordiv_exact_inv(new_h - vmin, bin, inv_bin));
                    const double ex = win - *WL_LDS_F64P(rec0 + 8u * (uint32_t)nb) + 0.0;
                    accepted = ((__ballot((ex >= 0.0) | (ex > lu)) >> 8) & 1ull) != 0ull;
                }
            }
            WL_PHASE(3)
            // -------- update (kernel/base.py:327-343; wanglandau.py:204-220) --------
            nacc_before = nacc_add;
            uint32_t sel_hi = 0u;
            bool changed = false;
            if (accepted) {
                if (STEP == SMOLMC_STEP_FLIP) occ[va1] = (uint8_t)n1;
                if (STEP == SMOLMC_STEP_SWAP) occ[a2] = (uint8_t)n2;
                sel_hi = 0x3ff00000u;
                acc_mu += dMu;
                acc_ew += dEw;
                if (EWF) {
                    if (STEP == SMOLMC_STEP_SWAP) {
                        if (dq1 != 0.0 || dq2 != 0.0) field_apply2(P, phi, lane, s1, dq1, s2, dq2);
                    } else if (dq1 != 0.0) {
                        field_apply<1>(P, phi, lane, s1, dq1);
                    }
                }
                hoff += dHa;
                xb32 = (float)(hoff * inv_bin);
                if (++since_sync >= resync_after) force_exact = true;
#ifndef WL_EXP_NOFLUSH
                changed = nb != b;
#endif
            } else {
                if (STEP == SMOLMC_STEP_SWAP) occ[va1] = (uint8_t)o1; // undo the tentative first flip
            }
            // -------- WangLandau._do_post_step (wanglandau.py:222-266) --------
            if (changed) flush_run(b); // the run in bin b ended with the previous step
            {
                const uint32_t sh = (uint32_t)uni((int)sel_hi);
                const double sel = __hiloint2double((int)sh, 0);
#pragma unroll
                for (int it = 0; it < NSLOT; ++it) {
                    acc[it] = fma(sel, d1[it], acc[it]);
                    run[it] += acc[it];
                }
                if (HAS_MU) run_mu += acc_mu;
                if (EWF) run_ew += acc_ew;
                nacc_add += sh >> 29;
            }
            run_n++;
            b = accepted ? nb : b;
            if (++wl_rem_check == wl_check) wl_rem_check = 0;
#ifndef WL_EXP_NOATOM // (timing experiments: -DWL_EXP_* remove one part each; wrong results)
            if (lane == 0) { // LDS atomics without return value
                __hip_atomic_fetch_add(WL_LDS_F64P(rec0 + 8u * (uint32_t)b), wl_m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_add(WL_LDS_U32P(cnt0 + 4u * (uint32_t)b), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
#endif
            s1 = s1n;
            a1 = a1n;
            __builtin_amdgcn_s_setprio(0);
            WL_PHASE(4)
            if (REPLAY) { // accept flag and running enthalpy of every step (what smolmc_replay returns)
                const double Hnow = exact_enthalpy();
                if (lane == 0) {
                    const size_t k = (size_t)r * (uint32_t)P.steps + ridx;
                    P.rp_acc[k] = (uint8_t)(nacc_add != nacc_before);
                    P.rp_H[k] = Hnow;
                }
                ridx++;
            }
            if (wl_rem_check == 0) {
                const LeanParamsKernarg Q = rare_params();
                const size_t o = (size_t)r * Q->wl.L;
                wl_m = wl_multi_flatness_check(wl_S, wl_cnt, nullptr, Q->wl.hist + o, Q->wl.occur + o, Q->wl.L, Q->wl.flat,
                                               Q->wl.div, wl_m, lane);
            }
            l4 += 4;
            l64 += 1;
        } while (--chunk != 0u);
        step = chunk_end;

        if (smp_countdown == 0) { // record one thinned sample of this walker
            const LeanParamsKernarg Q = rare_params();
            const int qF = Q->F;
            double *const q_feat = Q->smp.feat;
            smp_countdown = (uint32_t)Q->smp.every;
            const size_t row = (size_t)smp_index * Q->R + r;
            smp_index++;
            // a pending flush owns pend_c (its cells were already zeroed): hand it over first
            pend_commit();
            const double fcur = current_features();
            if (lane < qF) q_feat[row * qF + lane] = fcur;
            const double Hnow = exact_enthalpy();
            if (lane == 0) {
                Q->smp.H[row] = Hnow;
                Q->smp.acc[row] = (uint8_t)(nacc_add != nacc_before);
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
        printf("wl phases (cycles per step): skeleton %.0f | proposal %.0f | gathers+tables %.0f | decision %.0f | update+post %.0f\n",
               (double)lph[0] / (double)P.steps, (double)lph[1] / (double)P.steps, (double)lph[2] / (double)P.steps,
               (double)lph[3] / (double)P.steps, (double)lph[4] / (double)P.steps);
#endif
    // ---- write back ---------------------------------------------------------------
    pend_commit(); // a flush of the last step
    if (run_n != 0u) { // the unfinished run of the current bin
        flush_run(b);
        pend_commit();
    }
    for (int slot = 0; slot < WL_ROWS; ++slot) { // write the cached rows back
        const int tag = (int)rdlane((uint32_t)vtag, slot);
        if (tag >= 0 && lane < P.F)
            unsafeAtomicAdd(P.wl.meanf + ((size_t)r * P.wl.L + tag) * P.F + lane, s_rows[(uint32_t)slot * rowF + lane]);
    }
    {
        uint32_t *dst = (uint32_t *)(P.occ + (size_t)r * P.Npad);
        for (int i = lane; i < P.Npad / 4; i += 64)
            dst[i] = *(const uint32_t *)(occ + lean_swz(4 * i, swa, swm, swb));
    }
    {
        const double fcur = current_features();
        if (lane < P.F) featp[lane] = fcur;
    }
    if (EWF)
        for (int j = lane; j < P.ew_nact; j += 64) P.ew_phi[(size_t)r * P.ew_nact + j] = phi[j];
    for (int i = lane; i < P.wl.L; i += 64) {
        const size_t o = (size_t)r * P.wl.L + i;
        P.wl.entropy[o] = wl_S[i];
        P.wl.hist[o] += (long long)wl_cnt[i];
        P.wl.occur[o] += (long long)wl_cnt[i];
    }
    if (REPLAY && lane == 0 && rp_bad) atomicOr(P.rp_err, 1);
    const double Hend = exact_enthalpy();
    if (lane == 0) {
        P.wl.m[r] = wl_m;
        P.wl.counter[r] = wl_counter0 + (long long)(uint32_t)P.steps;
        P.enthalpy[r] = Hend;
        P.nsteps[r] = step;
        P.nacc[r] += nacc_add;
        if (P.steps) P.last_acc[r] = (uint8_t)(nacc_add != nacc_before);
    }
}

#undef key0
#undef key1

static long wl_gcd(long a, long b) { while (b) { const long t = a % b; a = b; b = t; } return a; }

// Launch.  GROUP ROTATION (round 5): the kernel's residency is set by LDS -- config 4: three four-walker
// workgroups per CU, 3072 walkers on 256 CUs -- and a step of this kernel takes as long at three waves per SIMD as
// at one (latency, not issue).  4096 walkers in one launch are therefore a full round of 3072 followed by a round of
// 1024 that takes just as long: 34 ms for 20000 steps against 18.8 ms for 3072 walkers.  When the walkers exceed the
// residency C by less than 4x and split into groups of s = gcd(R, C) walkers, the launch becomes g = R / s
// sub-launches of c = C / s groups each -- sub-launch j runs groups j, j + 1, ..., j + c - 1 (mod g) for steps / c
// steps -- every group runs c times, every sub-launch fills the chip exactly: time R / C instead of ceil(R / C)
// rounds.  Walkers are independent and carry their whole state through HBM between launches, so the chains are
// the chains of one long launch (SMOLMC_NO_ROTATE: A/B switch; tests compare both against the oracle).
template <int NSLOT, int MM, int STEP, bool REPLAY, bool MU, bool EW>
static int launch_wl_kern(smolmc_handle *h, const LeanParams &lp) {
    auto kern = mc_wl_kernel<NSLOT, MM, STEP, REPLAY, MU, EW>;
    if (h->lean_lds > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lean_lds));
    long C = 0;
    if (!REPLAY && lp.smp.every == 0 && getenv("SMOLMC_NO_ROTATE") == nullptr) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kern, 256, h->lean_lds) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess)
            C = 4L * per_cu * cus;
    }
    const long R = h->R, s = C > 0 ? wl_gcd(R, C) : 0;
    const long c = s ? C / s : 0, g = s ? R / s : 0;
    const bool rotate = C > 0 && R > C && R < 4 * C && s % 4 == 0 && c <= 16 && g <= 64 && lp.steps >= 64 * c;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    LeanParams q = lp;
    long long left = lp.steps;
    if (rotate) {
        q.launch_slots = (int)(c * s); q.rot_s = (int)s; q.rot_g = (int)g;
        q.steps = lp.steps / c;
        for (int j = 0; j < (int)g; ++j) {
            q.rot_j = j;
            hipLaunchKernelGGL(kern, dim3((unsigned)(c * s / 4)), dim3(256), h->lean_lds, h->stream, q);
        }
        HIPCHK(hipGetLastError());
        left = lp.steps - q.steps * c; // (steps % c: all walkers together below)
        q.launch_slots = 0;
    }
    if (left > 0) {
        q.steps = left;
        hipLaunchKernelGGL(kern, dim3((unsigned)((h->R + 3) / 4)), dim3(256), h->lean_lds, h->stream, q);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return 0;
}
template <int NSLOT, int MM, int STEP, bool REPLAY = false>
static int launch_wl_inst(smolmc_handle *h, const LeanParams &lp) {
    const bool mu = lp.mu_row != nullptr, ew = lp.ew_G != nullptr; // (Ewald: only with the field in LDS, engine.hip)
    if (ew) return mu ? launch_wl_kern<NSLOT, MM, STEP, REPLAY, true, true>(h, lp) : launch_wl_kern<NSLOT, MM, STEP, REPLAY, false, true>(h, lp);
    return mu ? launch_wl_kern<NSLOT, MM, STEP, REPLAY, true, false>(h, lp) : launch_wl_kern<NSLOT, MM, STEP, REPLAY, false, false>(h, lp);
}
template <int NSLOT> static int launch_wl_nslot(smolmc_handle *h, const LeanParams &lp) {
    const bool swap = h->cfg.step_type == SMOLMC_STEP_SWAP;
    if (h->lean_mm == 2)
        return swap ? launch_wl_inst<NSLOT, 2, SMOLMC_STEP_SWAP>(h, lp) : launch_wl_inst<NSLOT, 2, SMOLMC_STEP_FLIP>(h, lp);
    return swap ? launch_wl_inst<NSLOT, 3, SMOLMC_STEP_SWAP>(h, lp) : launch_wl_inst<NSLOT, 3, SMOLMC_STEP_FLIP>(h, lp);
}
template <int NSLOT> static int launch_wl_replay_nslot(smolmc_handle *h, const LeanParams &lp) {
    const bool swap = h->cfg.step_type == SMOLMC_STEP_SWAP;
    if (h->lean_mm == 2)
        return swap ? launch_wl_inst<NSLOT, 2, SMOLMC_STEP_SWAP, true>(h, lp) : launch_wl_inst<NSLOT, 2, SMOLMC_STEP_FLIP, true>(h, lp);
    return swap ? launch_wl_inst<NSLOT, 3, SMOLMC_STEP_SWAP, true>(h, lp) : launch_wl_inst<NSLOT, 3, SMOLMC_STEP_FLIP, true>(h, lp);
}
