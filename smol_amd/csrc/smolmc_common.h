// smolmc_common.h -- shared declarations of the MI355X (gfx950) ensemble Monte-Carlo engine.
// Translation units: engine.hip (C-ABI, table preparation, evaluation kernels),
// general_n{2,4,8,16}.hip (mc_kernel instantiations), lean_n{2,4}.hip (mc_lean_kernel and
// mc_table_kernel instantiations).  mc_general.h / mc_lean.h hold the kernel templates.
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
#pragma once
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
extern thread_local std::string smolmc_g_err;
static int fail(const std::string &m) {
    smolmc_g_err = m;
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
    // potential field in HBM (general kernel): phi[r][j] = frozen[j] + sum_{k != j} q_k G[j][k]
    // over the changeable sites (contiguous, ew_act_base >= 0); see DESIGN 4.4
    int ew_field;
    double *ew_phi;               // [R][ew_nact]
    // translation-compressed site kernel (null: none), see field_sweep_gx
    const double *ew_gx;
    const uint32_t *ew_E8, *ew_S8; // [ew_nact] byte offsets: entry of (s, j) at E8[j] + S8[s]
    const double *mu;             // [N][mu_W]
    // MCBias (smol/moca/kernel/bias.py): table [N][bias_W], running bias / net charge [R]
    int bias_type, bias_W;
    const double *bias_tab;
    double bias_pen;
    double *bias, *charge;      // charge: [R x SMOLMC_MAX_BIAS_ROWS] running sum_sites table_r - b_r
    int bias_rows;              // SquareHyperplaneBias: number of hyperplanes (tables [rows][N][W])
    size_t bias_row_stride;     // N * W
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
    int wl_sum_mode;              // mc_kernel with update_period == 1: wl_meanf holds running SUMS (mean = sum /
                                  // occurrences, formed when read): one fire-and-forget atomic per feature and
                                  // step instead of a read-modify-write round trip on the step's path
    // Metropolis feature accumulators in LDS: acc_by_slot = 0 -> [Fce][64] cells (feature, lane);
    // 1 -> [nclasses][Cpad] cells (class, slot) + [Fce] scratch, used in interaction mode when
    // that is smaller (models with many orbits)
    int acc_by_slot;
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

// TableFlip proposal batches (mc_table_kernel, mc_table_multi_kernel): did this lane's scan examine a site that an
// accepted step has just changed?  The lane keeps the sites it examined as 32 half-words (q_c, 0xffff = none); the
// flipped sites are lane f < nfl of vsite.  Running minimum over the half-words of q_c ^ (site | site << 16)
// (v_pk_min_u16): a zero half-word <=> a kept site is a flipped site -- two VALU instructions per word and flip
// (the borrow trick it replaces took four: config 5, hot ladder, 10.6 -> 10.3 ms per sweep).
__device__ __forceinline__ bool batch_lane_examined(const uint32_t (&q_c)[16], int vsite, int nfl) {
    typedef unsigned short smolmc_us2 __attribute__((ext_vector_type(2)));
    smolmc_us2 mn = {0xffffu, 0xffffu};
    for (int f = 0; f < nfl; ++f) {
        const uint32_t sf = rdlane((uint32_t)vsite, f);
        const uint32_t pat = sf | (sf << 16);
#pragma unroll
        for (int i = 0; i < 16; ++i) mn = __builtin_elementwise_min(mn, __builtin_bit_cast(smolmc_us2, q_c[i] ^ pat));
    }
    return mn.x == 0 || mn.y == 0;
}
__device__ __forceinline__ float uni_f(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
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


// the same with a precomputed 1 / y: the estimate floor(x * inv_y) is within one of the exact
// quotient floor and the remainder test moves it there (no float64 division per step)
__device__ __forceinline__ double floordiv_exact_inv(double x, double y, double inv_y) {
    double q = floor(x * inv_y);
    double r = fma(-q, y, x);
    if (r < 0) q -= 1.0;
    else if (r >= y) q += 1.0;
    return q;
}

// ----------------------------------------------------------------------------
// Ewald potential-field sweep (shared by the lean, multi-sublattice and general kernels; phi may
// live in LDS or in HBM)
// ----------------------------------------------------------------------------
// One straight-line batch of the potential-field sweep: U groups of 64 entries starting at group
// g0, entry j gains dq1 * ga[j] (+ dq2 * gb[j]).  All loads of the batch are issued first (one
// exposed latency per batch), addresses are base + immediate offsets (no per-element address
// arithmetic: the clamped / masked form of this loop cost ~10 VALU and, worse, compiler-made
// branches per element).  Groups below gdone were already updated by an earlier batch -- the
// last batch of a sweep is shifted back so that it ends on the last full group -- and are
// rewritten unchanged (coefficient 0).
template <int U, bool TWO>
__device__ __forceinline__ void field_sweep_batch(double *phi, const double *ga, const double *gb, int lane,
                                                  int g0, int gdone, double dq1, double dq2) {
    const int j = g0 * 64 + lane;
    const double *pa = ga + j, *pb = gb + j;
    double *pp = phi + j;
    double va[U], vb[U], pv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        va[u] = pa[64 * u];
        if (TWO) vb[u] = pb[64 * u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) pv[u] = pp[64 * u];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool fresh = g0 + u >= gdone; // wave-uniform
        double v = fma(fresh ? dq1 : 0.0, va[u], pv[u]);
        if (TWO) v = fma(fresh ? dq2 : 0.0, vb[u], v);
        pp[64 * u] = v;
    }
}

template <bool TWO, int UB = 0>
__device__ __forceinline__ void field_sweep(double *phi, const double *ga, const double *gb, int lane, int na,
                                            double dq1, double dq2) {
    // groups per batch, measured on the 1728 cation sites (27 groups) of a 12^3 rocksalt cell:
    // one row 9 / 14 / 27 -> 1.52 / 1.63 / 1.56 ms (config 3), two rows 9 / 14 / 27 -> 2.80 / 2.70 /
    // 2.93 ms (swap + Ewald); the row reads run at several TB/s out of L2 / Infinity Cache, so
    // the differences are memory-system effects, not instruction counts
    // (UB: the general kernel is held to 128 VGPRs and asks for smaller batches)
    constexpr int U = UB ? UB : (TWO ? 14 : 9);
    const int ngf = na >> 6; // full groups of 64 entries
    int g = 0;
    if (ngf >= U) {
        do {
            const int g0 = min(g, ngf - U);
            field_sweep_batch<U, TWO>(phi, ga, gb, lane, g0, g, dq1, dq2);
            g = g0 + U;
        } while (g < ngf);
    }
    for (; g + 4 <= ngf; g += 4) field_sweep_batch<4, TWO>(phi, ga, gb, lane, g, g, dq1, dq2);
    for (; g < ngf; ++g) field_sweep_batch<1, TWO>(phi, ga, gb, lane, g, g, dq1, dq2);
    const int j = ngf * 64 + lane;
    if (j < na) { // ragged tail
        double v = fma(dq1, ga[j], phi[j]);
        if (TWO) v = fma(dq2, gb[j], v);
        phi[j] = v;
    }
}

// The same sweep from the TRANSLATION-COMPRESSED site kernel (engine.hip, compress_ewald_rows): on a
// supercell of a periodic lattice G[s][j] depends on the sublattices of s and j and on the
// translation between them only, so one table per sublattice pair, extended over the differences
// -(d-1) .. d-1 of each translation coordinate (no modular arithmetic), replaces the N rows:
// the entry of (s, j) sits at byte E8[j] + S8[s] of gx.  For BASELINE config 3 that is 97 KB shared
// by every walker (L2-resident) instead of a 13.8 KB row of a 48 MB matrix per accepted flip.
//
// NF flips of one accepted step (a flip: 1; a swap: 2; TableFlip: up to SMOLMC_MAX_STEP_FLIPS, four
// at a time) in ONE pass over phi: every entry gains sum_f dq[f] * G[s_f][j] -- NF gathers from the
// compressed tables, one read-modify-write of phi.  (G[s][s] == 0 by construction: the entry of a
// flipped site sees the other flips of the step and not its own, no patching needed.)
// The update of an accepted step is a chain of dependent memory round trips, ~1300 cycles each
// (measured on config 5: 14 of them per accepted 3-flip step, 19 000 cycles): the E8 entries of up
// to 36 groups are therefore fetched in one go, then the gathers run in batches of 9 groups, each
// one round trip (config 5, 27 groups: 1 + 3 round trips instead of 14).
// A single flip (NF = 1) on a lattice of >= 27 groups takes all its gathers in one batch of 27.
template <int NF, int U>
__device__ __forceinline__ void field_sweep_gx_groups(double *pp, const uint32_t (&e)[U], const unsigned char *gx,
                                                      const uint32_t (&s8)[NF], const double (&dq)[NF], int first_fresh) {
    double v[U][NF], pv[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int f = 0; f < NF; ++f) v[u][f] = *(const double *)(gx + (size_t)(e[u] + s8[f]));
#pragma unroll
    for (int u = 0; u < U; ++u) pv[u] = pp[64 * u];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        // (groups below first_fresh were updated by an earlier batch -- the last batch of a sweep is
        // shifted back so that it ends on the last full group -- and are rewritten unchanged)
        const bool fresh = u >= first_fresh; // wave-uniform
        double x = pv[u];
#pragma unroll
        for (int f = 0; f < NF; ++f) x = fma(fresh ? dq[f] : 0.0, v[u][f], x);
        pp[64 * u] = x;
    }
}
// A lattice of fewer than U full groups (na < 64 U: fewer than 576 changeable sites for the batches of nine) in masked
// batches of US slots, entry 64 (g0 + u) + lane where it exists (the other lanes re-read the last entry and store
// nothing): three round trips -- E8, table entries, phi -- per batch.  Until round 6 such lattices went group by group, two
// dependent round trips each (the reference's LiNiO2 model in a 6^3 cell, seven groups: 3.4e8 Wang-Landau steps/s against
// 5.1e8 in the 8^3 cell; the 8^3 ternary rocksalt of config 3, eight groups: 7 % behind 12^3 instead of ahead).  Batches of
// FOUR slots in a loop: one batch of nine, inlined beside the large lattices' batches, cost THEIR kernels registers (config 5
// at 12^3: 4 -> 30 vector spills, 10.27 -> 10.58 ms per sweep; out of line the call convention cost more still).
template <int NF, int US>
__device__ __forceinline__ void field_sweep_gx_small(double *phi, const uint32_t *E8, const unsigned char *gx, int lane, int na,
                                                     const uint32_t (&s8)[NF], const double (&dq)[NF]) {
    for (int j0 = lane; j0 - lane < na; j0 += 64 * US) { // (uniform trip count)
        uint32_t e[US];
        double v[US][NF], pv[US];
#pragma unroll
        for (int u = 0; u < US; ++u) e[u] = E8[min(j0 + 64 * u, na - 1)];
#pragma unroll
        for (int u = 0; u < US; ++u)
#pragma unroll
            for (int f = 0; f < NF; ++f) v[u][f] = *(const double *)(gx + (size_t)(e[u] + s8[f]));
#pragma unroll
        for (int u = 0; u < US; ++u) pv[u] = phi[min(j0 + 64 * u, na - 1)];
#pragma unroll
        for (int u = 0; u < US; ++u) {
            double x = pv[u];
#pragma unroll
            for (int f = 0; f < NF; ++f) x = fma(dq[f], v[u][f], x);
            if (j0 + 64 * u < na) phi[j0 + 64 * u] = x;
        }
    }
}
template <int NF, int U, int NB>
__device__ __forceinline__ void field_sweep_gx_sized(double *phi, const uint32_t *E8, const unsigned char *gx, int lane,
                                                     int na, const uint32_t (&s8)[NF], const double (&dq)[NF], int gstart = 0) {
    const int ngf = na >> 6; // full groups of 64 entries
    int g = gstart;          // (groups below gstart: done by the caller, see field_sweep_gx_pre27)
#ifndef SMOLMC_NO_SMALL_SWEEP // A/B switch (tools/build_variant.sh)
    // (flips and swaps only: the three- and four-flip sweeps of the TableFlip kernels keep the group-by-group form -- those
    // kernels sit at 256 VGPRs, and the extra batch moved their spills: config 5 at 12^3 lost 1 % for 23 % at 8^3)
    if (NF <= 2 && __builtin_expect(gstart == 0 && ngf < U, 0)) { // (uniform)
        field_sweep_gx_small<NF, 4>(phi, E8, gx, lane, na, s8, dq);
        return;
    }
#endif
    // Chunks of up to NB batches of U groups: the E8 entries of the whole chunk are fetched first, then the
    // batches run, each one round trip to the tables.  When fewer than U groups are left the last batch is
    // shifted back so that it ends on the last full group; the groups it shares with the batch before are
    // rewritten unchanged (coefficient 0).  (Flips and swaps only: with the three or four flips of a TableFlip
    // step the extra unrolled batch costs those kernels their registers -- 250 VGPRs, 128 SGPR spills.  Round 5
    // added the swaps and put the shifted batch into the chunk: the seven groups a 16-group lattice -- the
    // reference's LiNiO2 model in an 8^3 cell -- leaves after one batch of nine were seven dependent chains
    // E8 -> table entries -> phi, 8000 cycles per accepted swap.)
    constexpr bool SHIFT = NF <= 2;
    while (ngf >= U && g < ngf) {
        const int rem = ngf - g;
        const int nb = min(NB, rem / U);                       // full batches of this chunk (uniform)
        const bool tail = SHIFT && nb < NB && rem > nb * U;    // + one shifted batch
        const int nbt = nb + (tail ? 1 : 0);
        if (nbt == 0) break;
        const int g_tail = ngf - U, ff_tail = g + nb * U - g_tail; // the shifted batch: first group, first fresh group
        uint32_t e[NB][U];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b < nbt) { // (uniform)
                const uint32_t *pe = E8 + (b < nb ? g + b * U : g_tail) * 64 + lane;
#pragma unroll
                for (int u = 0; u < U; ++u) e[b][u] = pe[64 * u];
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (b < nbt)
                field_sweep_gx_groups<NF, U>(phi + (b < nb ? g + b * U : g_tail) * 64 + lane, e[b], gx, s8, dq, b < nb ? 0 : ff_tail);
        g = tail ? ngf : g + nb * U;
    }
    for (; g < ngf; ++g) { // (lattices of fewer than U groups) group by group
        const uint32_t e = E8[g * 64 + lane];
        double x = phi[g * 64 + lane];
#pragma unroll
        for (int f = 0; f < NF; ++f) x = fma(dq[f], *(const double *)(gx + (size_t)(e + s8[f])), x);
        phi[g * 64 + lane] = x;
    }
    const int j = ngf * 64 + lane;
    if (j < na) { // ragged tail
        const uint32_t e = E8[j];
        double x = phi[j];
#pragma unroll
        for (int f = 0; f < NF; ++f) x = fma(dq[f], *(const double *)(gx + (size_t)(e + s8[f])), x);
        phi[j] = x;
    }
}
// Single flip with the E8 entries of the first 27 groups already in registers: they are lane
// constants of the launch (E8[64 u + lane]), and fetching them is the first of the three dependent
// round trips of a sweep (E8 -> table entries -> phi).  The kernels that take the 27-group batch hold
// 27 such registers inside the sweep anyway; kept across the step loop they cost nothing at the peak.
__device__ __forceinline__ void field_sweep_gx_pre27(double *phi, const uint32_t *E8, const unsigned char *gx, int lane, int na,
                                                     const uint32_t (&s8)[1], const double (&dq)[1], const uint32_t (&e0)[27]) {
    field_sweep_gx_groups<1, 27>(phi + lane, e0, gx, s8, dq, 0);
    if ((na >> 6) > 27 || (na & 63)) field_sweep_gx_sized<1, 27, 1>(phi, E8, gx, lane, na, s8, dq, 27);
}
// FOOT: register footprint the caller can afford.  2: the 27-group batch for single flips (the
// one-sublattice flip kernels, whose occupancy is bound by the field in LDS anyway); 1: chunks of
// four batches with the E entries fetched first; 0: plain batches of nine -- the multi-sublattice
// kernels with the field in HBM live on their occupancy (LiNiO2 8^3 flips: 4.2e9 -> 2.7e9 steps/s
// with the big batch's 233 VGPRs); -1: batches of four groups, for the four-flip sweep of the
// pending-update list in those kernels.
template <int NF, int FOOT = 1>
__device__ __forceinline__ void field_sweep_gx_multi(double *phi, const uint32_t *E8, const unsigned char *gx, int lane,
                                                     int na, const uint32_t (&s8)[NF], const double (&dq)[NF]) {
    if (FOOT == 2 && NF == 1 && (na >> 6) >= 27) field_sweep_gx_sized<NF, 27, 1>(phi, E8, gx, lane, na, s8, dq);
    else if (FOOT >= 1) field_sweep_gx_sized<NF, 9, 4>(phi, E8, gx, lane, na, s8, dq);
    else if (FOOT == 0) field_sweep_gx_sized<NF, 9, 1>(phi, E8, gx, lane, na, s8, dq);
    else field_sweep_gx_sized<NF, 4, 1>(phi, E8, gx, lane, na, s8, dq); // (-1: four flips at a time inside a 128-register budget)
}

// ----------------------------------------------------------------------------
// lean Metropolis kernel (parameter blocks; the kernel itself is in mc_lean.h): one site class, one contiguous active sublattice with the
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
    double *meanf;       // [R][L][F]  running means -- or, while sum_mode is set, running SUMS
    int sum_mode;        // lean kernel with update_period == 1: mean = sum / occurrences (no
                         // read-modify-write round trip per step; converted when read)
    double *m;           // [R]
    long long *counter;  // [R]
};

struct LeanSlot {
    uint32_t doff8;      // byte offset of the slot's delta table
    uint32_t stride8[3]; // 8 * stride of the other members
    uint32_t feat;       // feature index (orbit id)
    uint32_t live;       // 0 for padded slots; else the number of feature tables of the slot (1, or K in
                         // the several-correlation-functions mode)
    double w;            // natural parameter * size / (ratio * J)
    double fs;           // size / (ratio * J)
};

struct LeanParams {
    // walker of launch slot q (= workgroup * waves + wave), or null: slot q runs walker q.  Set when
    // the walkers' temperatures differ (an exchange ladder): see walker_order_kernel, engine.hip
    const int *order;
    const uint16_t *idx;   // [N][64][NSLOT][MM]
    const uint32_t *idx32; // the same entries as 32-bit words (one-wave-per-workgroup layout)
    const double *dt;      // delta tables, all padded to a common [S*S][NTP] shape
    const LeanSlot *slots; // [NSLOT][64]
    const double *mu_row;  // [ncodes] chemical potentials of the active sublattice (or null)
    // MCBias on the single active sublattice: bias_pair[old * 8 + new] = log(f_new / f_old)
    // (FugacityBias) or q_new - q_old (SquareChargeBias); running bias / net charge per walker.
    // mc_lean_multi_kernel: one such table per sublattice, bias_pair[sub * 64 + old * 8 + new].
    int bias_type;
    // (in the padding behind bias_type: no other offset of this block moves) device ring of a BIASED handle: the bias
    // column of the block starts this many doubles behind smp.H (both live in one arena); 0: the kernel records no
    // bias column (the snapshot path does).  Round 6: biased Metropolis handles record their rows in-kernel.
    uint32_t smp_bias_off;
    const double *bias_pair;
    double bias_pen;
    int bias_rows, bias_row_stride; // SquareHyperplaneBias: bias_rows pair tables, bias_row_stride doubles apart (one row otherwise)
    double *bias, *charge;
    uint8_t *occ;
    double *enthalpy, *features;
    const double *beta;
    const uint64_t *seeds;
    uint64_t *nsteps, *nacc;
    uint8_t *last_acc;
    int dt_len, R, N, Npad, F, Fce, sbase, nact, ncodes;
    uint32_t nt8, snt8;    // 8*NTP and 8*NTP*S: (old, new) -> byte offset old*snt8 + new*nt8
    uint32_t ktab8;        // KF mode: byte size of one table
    // KF mode: the correlation-function tables, in HBM / L2 (they are read on accepted steps only;
    // in LDS they cost config 3 its second workgroup per CU).  The group of the slot whose decision
    // table starts at byte D of dt holds SMOLMC_LEAN_MAX_KF tables from byte D * SMOLMC_LEAN_MAX_KF.
    const double *dtk;
    // LDS address of site s = s ^ (((s >> swz_a) & swz_m) << swz_b): a bank swizzle chosen on
    // the host (bank-conflict model over the cluster tables); idx rows hold swizzled addresses
    int swz_a, swz_m, swz_b, Nlds;
    long long steps;
    // bound on |float32 tree sum - exact sum| of the per-lane enthalpy partials of one step
    // (see the fast accept decision in mc_lean_kernel); 0 disables the float32 pre-test
    double fast_eps;
    SampleBufs smp;
    // compact Ewald term (see build_compact_ewald); feature index Fce, coefficient ew_coef
    int ew_W, ew_nact, ew_act_base;
    const int *ew_act;
    const double *ew_G, *ew_qs, *ew_dg, *ew_frozen;
    double ew_coef;
    // potential-field mode: phi[r][j] = sum over changeable sites k != j of q(k, occ_k) G[j][k]
    // lives in LDS for the launch (HBM copy between launches): a proposal costs O(1), an
    // accepted flip one row update
    int ew_field;
    double *ew_phi;          // [R][ew_nact]  (includes the frozen-site sums)
    const double *ew_qrow, *ew_dgrow; // [8] charge / diagonal term per species code (field mode:
                                      // identical for every active site, checked at create)
    WlParams wl;
    // mc_lean_multi_kernel: several site classes / active sublattices (classes == sublattices)
    int m_ncls, m_nsub, m_ndims;         // m_ndims = sum of species over the sublattices (TableFlip)
    int m_sbase[4], m_nact[4], m_ncodes[4], m_cls[4];
    double m_cum[4];                     // cumulative sublattice probabilities
    const double *m_mu, *m_q, *m_dg;     // [4][8] per-sublattice mu / charge / diagonal rows (or null)
    // TableFlip (mcusher.py:397-711) for the single active sublattice
    int tf_n;               // number of flip vectors
    const int *tf_table;    // [tf_n][ncodes]
    const double *tf_w;     // [2 tf_n]
    double tf_sw;           // swap_weight
    const double *tf_ln;    // [tf_ln_len] ln(k), host libm (LDS copy; 0 = compute on the device)
    int tf_ln_len;
    // translation-compressed site kernel for the potential-field updates (null: rows of ew_G);
    // read through rare_params() where an accepted flip needs them, never live across the step loop
    const double *ew_gx;
    const uint32_t *ew_E8, *ew_S8;
    // group rotation (mc_wl_kernel, see launch_wl_kern): this launch runs launch_slots walkers, slot q = walker
    // ((rot_j + q / rot_s) % rot_g) * rot_s + q % rot_s; launch_slots 0 = all R walkers, slot q = walker q
    int launch_slots, rot_s, rot_g, rot_j;
    // replay (REPLAY instantiations only): host-provided proposals / uniforms, per-step outputs
    const int *rp_steps;   // [R][steps][4]
    const double *rp_u;    // [R][steps]
    uint8_t *rp_acc;       // [R][steps]
    double *rp_H;          // [R][steps]
    int *rp_err;           // set when a record does not fit the kernel's step type
    const double *rp_lp;   // TableFlip replay: [R][steps] a-priori factors (NaN = derive) or null
    double *rp_lp_out;     // [R][steps] the factor that entered the exponent (null: not recorded)
};

__device__ __forceinline__ int lean_swz(int s, int a, int m, int b) { return s ^ (((s >> a) & m) << b); }


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


// one local record (site, orbit) of the reference's LocalEvalData (processor/expansion.py:24-36) packed
// for the universal kernel: everything a wave needs of it in one uniform 96-byte load instead of a
// dozen dependent loads from the flattened smolmc_tables arrays
struct URec {
    int32_t I, K, Nt, J;  // sites per cluster, functions (1 in interaction mode), tensor length, rows
    int32_t st[6];        // tensor strides of the members
    int32_t feat, pad;    // first feature index (bit_id / orbit id)
    int64_t idx_off;      // offset of the rows in loc_idx
    int64_t t_off;        // offset of the tensor(s) in corr_tensors / interaction_tensors
    double scale;         // size / ratio / J
    double ratio;         // cluster_ratio (the feature pass divides like the reference: p / ratio / J)
};

// ... the part the enthalpy pass reads, one gather of 56 bytes per cluster row.  Records that agree in every
// field are stored once (a translation-invariant model has one per orbit and site class, not one per site): the
// table is then a few cache lines, or sits in LDS (UParams::dict_lds).
struct URecE {
    int32_t st[6];   // tensor strides of the members (0 beyond the cluster)
    int32_t K, Nt;   // functions (1 in interaction mode), tensor length
    int32_t feat;    // first feature index
    uint32_t t_off;  // offset of the tensor(s) in corr_tensors / interaction_tensors
    double scale;    // size / ratio / J
    double nat0;     // natural parameter of feature `feat` (all a record of one function needs)
};
#define SMOLMC_UNIV_SCRATCH 128         // per-wave step scratch of the universal kernel: flips of a replayed step
#define SMOLMC_UNIV_SCRATCH_TABLE 2464  // ... + TableFlip: counts, picks, masked weights (x2), a-priori factors, running sums
#define SMOLMC_UNIV_DICT_RECS 64    // dictionaries in LDS: at most this many distinct records,
#define SMOLMC_UNIV_DICT_TENS 1024  // ... tensor entries (doubles)
#define SMOLMC_UNIV_DICT_NAT 128    // ... and features
#define SMOLMC_UNIV_DICT_BYTES (SMOLMC_UNIV_DICT_RECS * 56 + SMOLMC_UNIV_DICT_TENS * 8 + SMOLMC_UNIV_DICT_NAT * 8)
// one cluster row of a site, packed (32 bytes, two 16-byte loads): the member sites (members beyond the
// cluster repeat the first one: their stride is 0) and the local record
struct URow {
    int32_t x[6];
    int32_t rec, pad;
};
// ... and in 16 bytes (round 6) for cells of at most 65535 sites: six u16 member sites + the record.  The rows are what
// the kernel streams -- 230 of them per swap step of config 2, a 15 MB table against 4 MB of L2 per XCD: 3.0 KB of
// Infinity-Cache / HBM fetches per step (round 5's PMC passes).  Measured: 25.0 -> 8.7 GB per launch (the 7.5 MB table
// hits L2 more often), and 4-11 % SLOWER: the kernel is VALU-bound (0.65-0.68 issue) and the unpacking adds ~23 vector
// instructions per step.  Behind -DSMOLMC_UNIV_ROWS16 (make EXTRA=..., then SMOLMC_UNIV_ROWS16=1): the default build has the
// 32-byte rows only -- a never-taken uniform branch in the row loads cost the default path 3-4 %.
struct URow16 {
    uint16_t x[6];
    uint16_t rec_lo, rec_hi;
};

// parameter block of the universal kernel (mc_univ.h)
struct UParams {
    KParams K;   // walker state, sublattices, Ewald / mu / bias tables, Wang-Landau state, replay, samples
    RefTables T; // reference-layout tables (device copies of smolmc_tables)
    const double *natural; // [F] natural parameters
    const URec *recs;      // [n_loc] packed local records, indexed like loc_orbit
    // the cluster rows of every site flattened over its records (lane <-> row in the enthalpy pass):
    const uint32_t *row_ptr;  // [N+1] rows of site s: row_ptr[s] .. row_ptr[s+1]
    const uint4 *rows;        // [nrows] URow
    const URecE *recs_e;      // [n_recs_e] the enthalpy pass's part of recs, distinct ones
    int n_recs_e;
    int tens_len;             // entries of the feature mode's tensor array the records reach
    int dict_lds;             // records, tensors and natural parameters are copied to LDS (workgroup-shared, SMOLMC_UNIV_DICT_BYTES)
    int rows_uniform;         // > 0: every site has this many rows (row_ptr[s] = s * rows_uniform, not read)
    int max_I;                // largest cluster of the model (sites)
    int all_k1;               // every record has one function
    // TableFlip
    int tf_n, tf_d;          // flip vectors, dims (species over the active sublattices)
    const int *tf_table;     // [tf_n][tf_d]
    const double *tf_w;      // [2 tf_n]
    double tf_sw;            // swap_weight
    const double *tf_ln;     // ln(k), k = 0 .. largest sublattice (host libm, as the oracle's log)
    const int *tf_dim_sub;   // [tf_d] sublattice of a dim
    int occ_lds;             // occupancy staged in LDS (else read / written in HBM)
    int dfeat_cells;         // > 0: per-wave LDS cells [dfeat_cells] take the step's feature deltas in the enthalpy
                             // pass (committed on acceptance); 0: features come from a second pass over the flips
    int dfeat_shift;         // ... each cell exists 1 << dfeat_shift times (LDS atomics serialise per address)
    int acc_cells;           // dfeat_cells > 0: F rounded up to even -- LDS cells of the feature changes accepted in this launch
    int lds_per_wave;
    int lds_shared;          // bytes of workgroup-shared LDS in front of the per-wave blocks (the dictionaries)
    int wl;                  // Wang-Landau kernel
    int rows16;              // rows are packed 16-byte records (u16 member sites: cells of <= 65535 sites) instead of URow
    // replay extras
    const double *rp_lp;     // [R][nsteps] a-priori factors (NaN = derive) or null
    double *rp_lp_out;       // [R][nsteps] or null
    int *rp_err;             // bit 0: step not in the flip table; bit 1: inactive site / impossible code
};

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

// One slot of the sample ring: `n` samples x R walkers recorded by one smolmc_run_sampled call.  Every array is a
// range of ONE device arena and of ONE pinned host arena with the same offsets (a single asynchronous copy moves it).
struct SampleSlot {
    unsigned char *d = nullptr, *hst = nullptr; // device arena, pinned host mirror
    size_t cap = 0;                             // bytes allocated (grow-only)
    size_t used = 0;                            // bytes of the recorded block
    size_t o_H = 0, o_feat = 0, o_acc = 0, o_occ = 0, o_bias = 0, o_wlm = 0, o_wlS = 0, o_wlh = 0, o_wlo = 0, o_wlf = 0;
    long long n = 0;                            // samples in the block
    int flags = 0;                              // SMOLMC_SAMPLE_* the block was recorded with
    int state = 0;                              // 0 empty, 1 recorded and not yet delivered, 2 delivered
    unsigned long long seq = 0;                 // order of recording
    hipEvent_t kernel_done = nullptr, copy_done = nullptr;
};

struct smolmc_handle {
    smolmc_config cfg;
    int device = 0;
    int *d_order = nullptr;    // launch-slot -> walker permutation (TableFlip kernels), see walker_order_kernel
    bool order_dirty = true;   // temperatures changed since it was computed
    int order_mode = -1;
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
    int waves_per_block_lean = 4; // mc_lean_multi_kernel (LDS-sized)
    // lean kernel (single class / single contiguous sublattice / interactions / no ewald)
    bool lean_tables = false, lean = false;
    int lean_nslot = 0, lean_mm = 0, lean_ncls = 0;
    bool lean_multi = false;            // dispatch to mc_lean_multi_kernel
    bool lean_multi_wl = false;         // ... its Wang-Landau variant (WLK)
    bool lean_solo = false;             // mc_lean_kernel in its one-wave-per-workgroup layout
    int lean_occ = 0;                   // > 0: the solo instantiation held to this many waves per SIMD
    int lean_wpb = 4;          // TableFlip kernel: walkers (waves) per workgroup -- 8 when one such workgroup fills a CU's LDS (see launch_table_ewm)
    size_t lean_lds_wpb8 = 0;  // its dynamic LDS with 8 waves
    int lean_kf = 0;                    // > 0: correlation features with up to lean_kf functions per orbit
    std::vector<uint16_t> lean_idx_host; // lane-packed index rows (kept for the 32-bit copy)
    std::vector<int> site_class_host;   // site -> class (255 = no clusters)
    size_t lean_lds = 0;
    LeanParams lp;
    // device-side samples (smolmc_run_sampled): two ring slots, each a device arena + a pinned host mirror; the
    // download of a slot runs on its own stream while the next block's kernel fills the other slot (see engine.hip)
    SampleSlot slots[2];
    int next_slot = 0;
    unsigned long long slot_seq = 0;
    hipStream_t copy_stream = nullptr;
    // scratch
    uint8_t *d_eval_occ = nullptr;
    size_t eval_occ_cap = 0;
    double *d_natural = nullptr;
    double *d_beta = nullptr;
    bool wl_sums = false; // kp.wl_meanf currently holds sums (lean Wang-Landau) instead of means
    std::vector<double> bias_host; // host copy of the MCBias table (initial bias in set_state)
    std::vector<double> bias_icpt; // SquareHyperplaneBias intercepts (zeros otherwise)
    std::vector<double> ew_qs_host, ew_dg_host; // compact-Ewald per-(site, code) charge / diagonal
    int ew_gx_dims[3] = {0, 0, 0}, ew_gx_blocks = 0; // translation-compressed site kernel (0: none)
    std::vector<uint8_t> site_ncodes; // species codes allowed on each site (occupancy validation)
    std::vector<uint8_t> site_active; // 1 on the sites of the active sublattices (replay validation)
    // universal kernel (mc_univ.h): always available; `univ` = every launch of this handle takes it
    UParams up;
    bool univ = false;
    bool general_ok = true;      // mc_kernel can run this model (else: why not)
    std::string general_reason;
    // lazy cluster features (engine.hip, build_mc_tables): the lean kernels of this handle carry the scalar features
    // only (d_lazy_scal [R][2]: Ewald energy, chemical work); the cluster part of kp.features is evaluated from the
    // occupancies when it is read
    bool lazy_tables = false, lazy = false, ce_dirty = false;
    double *d_lazy_scal = nullptr;
    std::string lean_reason;           // why the model runs neither lean family (first failing condition; empty: it does)
    int univ_wpb = 4;
    int max_step_flips = 2;      // most flips a native step of this handle makes (TableFlip: from the table)
    // site relabelling behind the boundary (engine.hip, plan_relabelling): the tables this handle was built from are
    // the caller's with the sites renumbered; occupancies and step records are translated at every entry point
    bool relabelled = false;
    std::vector<int32_t> new_of, old_of; // caller's site -> engine's site, and back
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


// launchers defined in the per-NSLOT translation units

#ifdef __HIPCC__
// Parameters that only rare paths of a step loop need (sample rows, flatness checks) are re-read
// from the kernel-argument segment where they are used: held in SGPRs for the whole loop they
// push the hot path into SGPR spills (v_readlane reloads on every step).  The empty asm makes the
// pointer opaque, so the scalar loads cannot be hoisted back out of the loop.  Valid in kernels
// whose first argument is the LeanParams block.
typedef const LeanParams __attribute__((address_space(4))) *LeanParamsKernarg;
__device__ __forceinline__ LeanParamsKernarg rare_params() {
    LeanParamsKernarg p = (LeanParamsKernarg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
#endif
int smolmc_launch_univ(smolmc_handle *h, const UParams &up, int replay);
int smolmc_launch_general_2(smolmc_handle *h, const KParams &kp, int replay);
int smolmc_launch_general_4(smolmc_handle *h, const KParams &kp, int replay);
int smolmc_launch_general_8(smolmc_handle *h, const KParams &kp, int replay);
int smolmc_launch_general_16(smolmc_handle *h, const KParams &kp, int replay);
int smolmc_launch_lean_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_table_bias_2(smolmc_handle *h, const LeanParams &lp); // TableFlip + MCBias (table_bias_n*.hip)
int smolmc_launch_table_bias_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_table_wl_2(smolmc_handle *h, const LeanParams &lp); // Wang-Landau + TableFlip (table_wl_n*.hip)
int smolmc_launch_table_wl_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_table_wl_2(smolmc_handle *h, const LeanParams &lp); // ... on the multi-class layout (multi_table_wl_n*.hip)
int smolmc_launch_multi_table_wl_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_table_wl_8(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_table_bias_2(smolmc_handle *h, const LeanParams &lp); // TableFlip + MCBias on the multi-class layout (multi_table_bias_n*.hip)
int smolmc_launch_multi_table_bias_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_table_bias_8(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_lean_4(smolmc_handle *h, const LeanParams &lp);
// replay instantiations of the TableFlip / biased lean kernels (0: those replays take the universal / general kernel)
#ifndef SMOLMC_HAVE_TABLE_REPLAY
#define SMOLMC_HAVE_TABLE_REPLAY 1
#endif
#ifndef SMOLMC_HAVE_BIAS_REPLAY
#define SMOLMC_HAVE_BIAS_REPLAY 1
#endif
int smolmc_launch_table_replay_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_table_replay_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_table_replay_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_table_replay_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_table_replay_8(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_lean_bias_replay_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_lean_bias_replay_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_bias_replay_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_bias_replay_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_bias_replay_8(smolmc_handle *h, const LeanParams &lp);
#define SMOLMC_WL_ROWS 32  // mc_wl_kernel: cached rows of per-bin feature sums per walker (LDS)
// per-walker LDS bytes of the Wang-Landau state of mc_lean_multi_kernel<..., WLK> (mc_lean_multi.h):
// S f64 [L] | counted steps u32 [L] | sums: a log of SMOLMC_WLM_LOG finished runs [F] -- running means: occurrences at
// launch start f64 [L] and SMOLMC_WL_ROWS cached rows [F]
#define SMOLMC_WLM_LOG 16
// per-walker LDS bytes of the per-bin state of mc_wl_kernel (mc_wl.h): entropies f64 [L + 2 guards] | step counts u32 [L]
__host__ __device__ inline size_t wl_lean_bins_bytes(int L) { return ((size_t)L + 2) * 8 + (((size_t)L * 4 + 7) & ~(size_t)7); }
__host__ __device__ inline size_t wl_multi_wave_bytes(int L, int F, int sum_mode) {
    return (size_t)L * 8 + (((size_t)L * 4 + 7) & ~(size_t)7) +
           (sum_mode ? (size_t)SMOLMC_WLM_LOG * F * 8 : (size_t)L * 8 + (size_t)SMOLMC_WL_ROWS * F * 8);
}
#define SMOLMC_LEAN_MAX_KF 6 // correlation functions per orbit served by the lean kernels (ternary triplets)
int smolmc_launch_lean_corr_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_lean_corr_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_wl_replay_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_wl_replay_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_replay_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_replay_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_replay_8(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_lean_replay_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_lean_replay_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_wl_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_wl_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_lean_bias_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_lean_bias_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_8(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_wl_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_wl_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_wl_8(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_wl_kf_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_wl_kf_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_wl_kf_8(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_wl_replay_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_wl_replay_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_wl_replay_8(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_bias_2(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_bias_4(smolmc_handle *h, const LeanParams &lp);
int smolmc_launch_multi_bias_8(smolmc_handle *h, const LeanParams &lp);
