/*
 * smolmc_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference algorithm for the one hot path this
 * repository accelerates.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (smol_amd/ + libsmolmc_hip.so) never does.
 *
 * Parity status: PINNED.  Every evaluator-level function below is checked
 * against outputs of the reference's own compiled Cython core
 * (smol/utils/cluster/{evaluator,ewald,correlations}.pyx built out-of-tree by
 * tests/golden/make_golden.py) on the committed fixtures tests/golden/ (npz files);
 * the kernel-level control flow (Metropolis / Wang-Landau / ushers), whose
 * Python home cannot be imported here (pymatgen/monty absent), is pinned by
 * replayed trajectories whose arithmetic came from that compiled core.
 * Ewald matrix VALUES are third-party (pymatgen) and unpinned; the arithmetic
 * applied to the matrix is pinned.
 *
 * Each function cites the reference lines it follows (paths relative to the
 * reference checkout).  Loop nests and accumulation order follow the
 * reference so float64 results agree to ~1 ulp with it.
 */
#include "../include/smolmc.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
typedef struct {
    int id, bit_id, I, K, N;
    const int32_t *strides;
    const double *ct; /* [K x N] */
    const double *it; /* [N] */
} orbit_rec;

static orbit_rec get_orbit(const smolmc_tables *t, int o) {
    orbit_rec r;
    r.id = t->orb_id[o];
    r.bit_id = t->orb_bit_id[o];
    r.I = t->orb_nsites[o];
    r.K = t->orb_nfunc[o];
    r.N = t->orb_tensor_len[o];
    r.strides = t->tensor_indices + t->orb_stride_off[o];
    r.ct = t->corr_tensors + t->orb_ctensor_off[o];
    r.it = t->interaction_tensors ? t->interaction_tensors + t->orb_itensor_off[o] : NULL;
    return r;
}

/* ---- H3: smol/utils/cluster/evaluator.pyx:121-168 ------------------------ */
void orc_correlations_from_occupancy(const smolmc_tables *t, const int32_t *occu, double *out) {
    memset(out, 0, sizeof(double) * (size_t)t->num_corr);
    out[0] = 1.0; /* empty cluster, :145 */
    for (int n = 0; n < t->n_orb; ++n) { /* :148 */
        orbit_rec orb = get_orbit(t, n);
        const int32_t *ind = t->full_idx + t->full_off[n];
        int I = orb.I;
        long J = (long)((t->full_off[n + 1] - t->full_off[n]) / I);
        int bit_id = orb.bit_id;
        for (int k = 0; k < orb.K; ++k) { /* :156 */
            double p = 0;
            for (long j = 0; j < J; ++j) { /* :158 */
                int index = 0;
                for (int i = 0; i < I; ++i) /* :160-161 */
                    index += orb.strides[i] * occu[ind[j * I + i]];
                p += orb.ct[(long)k * orb.N + index]; /* :163 */
            }
            out[bit_id] = p / (double)J; /* :164 */
            bit_id++;
        }
    }
}

/* ---- H3: smol/utils/cluster/evaluator.pyx:170-209 ------------------------ */
void orc_interactions_from_occupancy(const smolmc_tables *t, const int32_t *occu, double *out) {
    memset(out, 0, sizeof(double) * (size_t)t->num_orbits);
    out[0] = t->offset; /* :193 */
    for (int n = 0; n < t->n_orb; ++n) { /* :195 */
        orbit_rec orb = get_orbit(t, n);
        const int32_t *ind = t->full_idx + t->full_off[n];
        int I = orb.I;
        long J = (long)((t->full_off[n + 1] - t->full_off[n]) / I);
        double p = 0;
        for (long j = 0; j < J; ++j) { /* :202 */
            int index = 0;
            for (int i = 0; i < I; ++i) index += orb.strides[i] * occu[ind[j * I + i]];
            p += orb.it[index]; /* :206 */
        }
        out[orb.id] = p / (double)J; /* :207 */
    }
}

/* ---- H1: smol/utils/cluster/evaluator.pyx:211-265 ------------------------ */
/* local records of `site` play the role of (self=local evaluator, cluster_ratio,
 * cluster_indices) built at processor/expansion.py:120-156.  Adds into out. */
void orc_delta_correlations(const smolmc_tables *t, const int32_t *occu_f, const int32_t *occu_i,
                            int site, double *out_add) {
    for (int64_t r = t->site_ptr[site]; r < t->site_ptr[site + 1]; ++r) { /* :244 */
        orbit_rec orb = get_orbit(t, t->loc_orbit[r]);
        const int32_t *ind = t->loc_idx + t->loc_off[r];
        int I = orb.I, J = t->loc_nrows[r];
        double ratio = t->loc_ratio[r];
        int bit_id = orb.bit_id;
        for (int k = 0; k < orb.K; ++k) { /* :253 */
            double p = 0;
            for (int j = 0; j < J; ++j) { /* :255 */
                int ind_i = 0, ind_f = 0;
                for (int i = 0; i < I; ++i) { /* :257-259 */
                    ind_i += orb.strides[i] * occu_i[ind[j * I + i]];
                    ind_f += orb.strides[i] * occu_f[ind[j * I + i]];
                }
                p += (orb.ct[(long)k * orb.N + ind_f] - orb.ct[(long)k * orb.N + ind_i]); /* :261 */
            }
            out_add[bit_id] += p / ratio / (double)J; /* :262 */
            bit_id++;
        }
    }
}

/* ---- H2: smol/utils/cluster/evaluator.pyx:267-317 ------------------------ */
void orc_delta_interactions(const smolmc_tables *t, const int32_t *occu_f, const int32_t *occu_i,
                            int site, double *out_add) {
    for (int64_t r = t->site_ptr[site]; r < t->site_ptr[site + 1]; ++r) { /* :302 */
        orbit_rec orb = get_orbit(t, t->loc_orbit[r]);
        const int32_t *ind = t->loc_idx + t->loc_off[r];
        int I = orb.I, J = t->loc_nrows[r];
        double p = 0;
        for (int j = 0; j < J; ++j) { /* :309 */
            int ind_i = 0, ind_f = 0;
            for (int i = 0; i < I; ++i) {
                ind_i += orb.strides[i] * occu_i[ind[j * I + i]];
                ind_f += orb.strides[i] * occu_f[ind[j * I + i]];
            }
            p += (orb.it[ind_f] - orb.it[ind_i]); /* :314 */
        }
        out_add[orb.id] += p / t->loc_ratio[r] / (double)J; /* :315 */
    }
}

/* ---- H4: smol/utils/cluster/ewald.pyx:9-59 ------------------------------- */
double orc_delta_ewald_single_flip(const smolmc_tables *t, const int32_t *occu_f,
                                   const int32_t *occu_i, int site_ind) {
    const int32_t *inds = t->ewald_inds;
    const double *m = t->ewald_matrix;
    int W = t->ewald_width;
    long M = t->ewald_dim;
    double out = 0;
    int add = inds[(long)site_ind * W + occu_f[site_ind]]; /* :38 */
    int sub = inds[(long)site_ind * W + occu_i[site_ind]]; /* :39 */
    for (int k = 0; k < t->num_sites; ++k) { /* :43 */
        int i = inds[(long)k * W + occu_f[k]];
        double out_k = 0;
        if (i != -1 && add != -1) { /* :46-50 */
            if (i != add)
                out_k = out_k + 2 * m[i * M + add];
            else
                out_k = out_k + m[i * M + add];
        }
        int j = inds[(long)k * W + occu_i[k]]; /* :52 */
        if (j != -1 && sub != -1) {
            if (j != sub)
                out_k = out_k - 2 * m[j * M + sub];
            else
                out_k = out_k - m[j * M + sub];
        }
        out += out_k; /* :58 */
    }
    return out;
}

/* ---- processor / ensemble level ------------------------------------------ */
int orc_num_ce_features(const smolmc_tables *t) {
    return t->feature_mode == SMOLMC_FEATURES_CORRELATIONS ? t->num_corr : t->num_orbits;
}
int orc_num_features(const smolmc_tables *t) {
    return orc_num_ce_features(t) + (t->has_ewald ? 1 : 0) + (t->has_mu ? 1 : 0);
}
/* Ensemble.natural_parameters: processor coefs (composite.py:86-87 concatenation)
 * + the chemical-work parameter -1 (ensemble.py:25,61-65) */
void orc_natural_parameters(const smolmc_tables *t, double *out) {
    int n = orc_num_ce_features(t);
    memcpy(out, t->ce_coefs, sizeof(double) * (size_t)n);
    if (t->has_ewald) out[n++] = t->ewald_coef;
    if (t->has_mu) out[n++] = -1.0;
}

/* Ensemble.compute_feature_vector (ensemble.py:323-351) over
 * Cluster{Expansion,Decomposition}Processor.compute_feature_vector
 * (expansion.py:165-189, :391-414), EwaldProcessor.compute_feature_vector
 * (processor/ewald.py:128-145), CompositeProcessor (composite.py:116-133). */
void orc_feature_vector(const smolmc_tables *t, const int32_t *occ, double *out) {
    int n = orc_num_ce_features(t);
    if (t->feature_mode == SMOLMC_FEATURES_CORRELATIONS)
        orc_correlations_from_occupancy(t, occ, out);
    else
        orc_interactions_from_occupancy(t, occ, out);
    for (int i = 0; i < n; ++i) out[i] *= (double)t->size;
    if (t->has_ewald) {
        /* np.sum(M[mask][:, mask]) with mask from get_ewald_occu (extern/ewald.py:102-130) */
        long M = t->ewald_dim;
        int W = t->ewald_width;
        double s = 0;
        for (int a = 0; a < t->num_sites; ++a) {
            int ia = t->ewald_inds[(long)a * W + occ[a]];
            if (ia == -1) continue;
            double row = 0;
            for (int b = 0; b < t->num_sites; ++b) {
                int ib = t->ewald_inds[(long)b * W + occ[b]];
                if (ib == -1) continue;
                row += t->ewald_matrix[ia * M + ib];
            }
            s += row;
        }
        out[n++] = s;
    }
    if (t->has_mu) { /* ensemble.py:343-349 */
        double w = 0;
        for (int s = 0; s < t->num_sites; ++s) w += t->mu_table[(long)s * t->mu_width + occ[s]];
        out[n++] = w;
    }
}

/* Ensemble.compute_feature_vector_change (ensemble.py:353-376): sequential flips
 * (expansion.py:217-229, processor/ewald.py:166-181), x size at the end
 * (expansion.py:231,464), mu delta against the ORIGINAL occupancy (ensemble.py:368-374).
 * work_f / work_i: scratch int32[N] buffers holding copies of occ (restored on exit). */
void orc_feature_vector_change(const smolmc_tables *t, const int32_t *occ, const int32_t *flips,
                               int nflips, int32_t *work_f, int32_t *work_i, double *out) {
    int n = orc_num_ce_features(t);
    int F = orc_num_features(t);
    memset(out, 0, sizeof(double) * (size_t)F);
    double dew = 0;
    for (int f = 0; f < nflips; ++f) {
        int site = flips[2 * f], code = flips[2 * f + 1];
        work_f[site] = code; /* occu_f = occu_i.copy(); occu_f[site] = code */
        if (t->feature_mode == SMOLMC_FEATURES_CORRELATIONS)
            orc_delta_correlations(t, work_f, work_i, site, out);
        else
            orc_delta_interactions(t, work_f, work_i, site, out);
        if (t->has_ewald) dew += orc_delta_ewald_single_flip(t, work_f, work_i, site);
        work_i[site] = code; /* occu_i = occu_f */
    }
    for (int i = 0; i < n; ++i) out[i] *= (double)t->size;
    if (t->has_ewald) out[n++] = dew;
    if (t->has_mu) {
        double dw = 0;
        for (int f = 0; f < nflips; ++f) {
            int site = flips[2 * f], code = flips[2 * f + 1];
            dw += t->mu_table[(long)site * t->mu_width + code] -
                  t->mu_table[(long)site * t->mu_width + occ[site]];
        }
        out[n++] = dw;
    }
    for (int f = 0; f < nflips; ++f) { /* restore scratch */
        int site = flips[2 * f];
        work_f[site] = occ[site];
        work_i[site] = occ[site];
    }
}

/* ---- engine RNG: Philox4x32-10 (Salmon et al., SC'11) -------------------- */
/* The engine's own counter-based stream (NOT the reference's PCG64; SURVEY
 * App. B "native mode").  key = walker seed (lo, hi); counter =
 * (step_lo, step_hi, block, 0).  Shared bit-for-bit with the HIP kernels. */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline uint32_t mulhi32(uint32_t w, uint32_t n) { return (uint32_t)(((uint64_t)w * n) >> 32); }
static inline double u53(uint32_t a, uint32_t b) {
    return (double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) * (1.0 / 9007199254740992.0);
}

/* exact floor(x / y) for y > 0: Python's float // (wanglandau.py:180) */
static inline double floordiv_exact(double x, double y) {
    double q = floor(x / y);
    double r = fma(-q, y, x);
    if (r < 0) q -= 1.0;
    else if (r >= y) q += 1.0;
    return q;
}

/* ---- Monte-Carlo handle --------------------------------------------------- */
typedef struct orc_mc {
    const smolmc_tables *t;
    smolmc_config cfg;
    int R, N, F, L;
    int32_t *occ;      /* R x N */
    int32_t *work_f, *work_i; /* R x N scratch copies */
    double *features;  /* R x F */
    double *enthalpy;  /* R */
    double *temperature;
    uint64_t *seeds, *nsteps, *naccepted;
    uint8_t *last_accepted;
    double *natural;   /* F */
    /* Wang-Landau aux state (wanglandau.py:107-122) */
    double *wl_entropy, *wl_meanf, *wl_m, *wl_cur_h, *wl_cur_f;
    int64_t *wl_hist, *wl_occur, *wl_counter;
    double *bias;      /* R: trace.bias (kernel/base.py:362-363) */
} orc_mc;

/* MCBias.compute_bias (smol/moca/kernel/bias.py:174-186 Fugacity, :264-277 SquareCharge) */
double orc_compute_bias(const smolmc_tables *t, const int32_t *occ) {
    int N = t->num_sites, W = t->bias_width;
    if (t->bias_type == SMOLMC_BIAS_FUGACITY) {
        double b = 0;
        for (int s = 0; s < N; ++s) b += log(t->bias_table[(size_t)s * W + occ[s]]);
        return b;
    }
    if (t->bias_type == SMOLMC_BIAS_SQUARE_CHARGE) {
        double c = 0;
        for (int s = 0; s < N; ++s) c += t->bias_table[(size_t)s * W + occ[s]];
        return -t->bias_penalty * (c * c);
    }
    if (t->bias_type == SMOLMC_BIAS_SQUARE_HYPERPLANE) {
        /* SquareHyperplaneBias.compute_bias (bias.py:352-366): n = species counts,
         * -penalty * sum((A n - b)^2); table r holds A[r][dim_id(site, code)], so
         * A_r . n = sum over the sites of table_r[site][occ] */
        double sq = 0;
        for (int k = 0; k < t->bias_rows; ++k) {
            const double *tab = t->bias_table + (size_t)k * N * W;
            double c = 0;
            for (int s = 0; s < N; ++s) c += tab[(size_t)s * W + occ[s]];
            c -= t->bias_intercepts ? t->bias_intercepts[k] : 0.0;
            sq += c * c;
        }
        return -t->bias_penalty * sq;
    }
    return 0.0;
}

/* compute_bias_change: FugacityBias overrides it with the log-ratio of the last flip per
 * site (bias.py:188-206); SquareChargeBias inherits the recompute-and-subtract default
 * (bias.py:75-93), restated on the running total charge */
double orc_compute_bias_change(const smolmc_tables *t, const int32_t *occ, const int32_t *flips,
                               int nflips) {
    int W = t->bias_width;
    if (t->bias_type == SMOLMC_BIAS_FUGACITY) {
        double d = 0;
        for (int f = 0; f < nflips; ++f) {
            int site = flips[2 * f], last = 1;
            for (int g = f + 1; g < nflips; ++g) last &= flips[2 * g] != site; /* dict keeps the last */
            if (!last) continue;
            d += log(t->bias_table[(size_t)site * W + flips[2 * f + 1]] /
                     t->bias_table[(size_t)site * W + occ[site]]);
        }
        return d;
    }
    if (t->bias_type == SMOLMC_BIAS_SQUARE_CHARGE) {
        double c = 0;
        for (int s = 0; s < t->num_sites; ++s) c += t->bias_table[(size_t)s * W + occ[s]];
        double cn = c;
        for (int f = 0; f < nflips; ++f) {
            int site = flips[2 * f], last = 1;
            for (int g = f + 1; g < nflips; ++g) last &= flips[2 * g] != site;
            if (!last) continue;
            cn += t->bias_table[(size_t)site * W + flips[2 * f + 1]] -
                  t->bias_table[(size_t)site * W + occ[site]];
        }
        return -t->bias_penalty * (cn * cn) - (-t->bias_penalty * (c * c));
    }
    if (t->bias_type == SMOLMC_BIAS_SQUARE_HYPERPLANE) {
        /* inherited MCBias.compute_bias_change (bias.py:75-93): bias of the occupancy with the
         * flips applied (the last flip of a site wins) minus bias of the occupancy */
        int N = t->num_sites;
        double sq_new = 0, sq_old = 0;
        for (int k = 0; k < t->bias_rows; ++k) {
            const double *tab = t->bias_table + (size_t)k * N * W;
            double c = 0;
            for (int s = 0; s < N; ++s) c += tab[(size_t)s * W + occ[s]];
            c -= t->bias_intercepts ? t->bias_intercepts[k] : 0.0;
            double cn = c;
            for (int f = 0; f < nflips; ++f) {
                int site = flips[2 * f], last = 1;
                for (int g = f + 1; g < nflips; ++g) last &= flips[2 * g] != site;
                if (!last) continue;
                cn += tab[(size_t)site * W + flips[2 * f + 1]] - tab[(size_t)site * W + occ[site]];
            }
            sq_old += c * c;
            sq_new += cn * cn;
        }
        return -t->bias_penalty * sq_new - (-t->bias_penalty * sq_old);
    }
    return 0.0;
}

static const double ORC_KB = 8.617333262145e-5; /* smol/constants.py:4 */

int orc_mc_create(const smolmc_tables *t, const smolmc_config *cfg, orc_mc **out) {
    if (t->bias_type && cfg->kernel_type == SMOLMC_KERNEL_WANGLANDAU)
        return 2; /* "Cannot apply bias to Wang-Landau simulation!" (wanglandau.py:127-128) */
    if (cfg->step_type == SMOLMC_STEP_TABLE_FLIP &&
        (t->n_flip_vectors > SMOLMC_MAX_FLIP_VECTORS || t->sub_code_ptr[t->n_sublattices] > SMOLMC_MAX_FLIP_DIMS))
        return 4;
    orc_mc *h = (orc_mc *)calloc(1, sizeof(orc_mc));
    if (!h) return 1;
    h->t = t;
    h->cfg = *cfg;
    h->R = cfg->n_replicas;
    h->N = t->num_sites;
    h->F = orc_num_features(t);
    size_t RN = (size_t)h->R * h->N, RF = (size_t)h->R * h->F;
    h->occ = (int32_t *)calloc(RN, 4);
    h->work_f = (int32_t *)calloc(RN, 4);
    h->work_i = (int32_t *)calloc(RN, 4);
    h->features = (double *)calloc(RF, 8);
    h->enthalpy = (double *)calloc(h->R, 8);
    h->temperature = (double *)calloc(h->R, 8);
    h->seeds = (uint64_t *)calloc(h->R, 8);
    h->nsteps = (uint64_t *)calloc(h->R, 8);
    h->naccepted = (uint64_t *)calloc(h->R, 8);
    h->last_accepted = (uint8_t *)calloc(h->R, 1);
    h->natural = (double *)calloc(h->F, 8);
    h->bias = (double *)calloc(h->R, 8);
    orc_natural_parameters(t, h->natural);
    if (cfg->kernel_type == SMOLMC_KERNEL_WANGLANDAU) {
        /* len(np.arange(min, max, bin)) (wanglandau.py:107) */
        h->L = (int)ceil((cfg->wl_max_enthalpy - cfg->wl_min_enthalpy) / cfg->wl_bin_size);
        size_t RL = (size_t)h->R * h->L;
        h->wl_entropy = (double *)calloc(RL, 8);
        h->wl_hist = (int64_t *)calloc(RL, 8);
        h->wl_occur = (int64_t *)calloc(RL, 8);
        h->wl_meanf = (double *)calloc(RL * h->F, 8);
        h->wl_m = (double *)calloc(h->R, 8);
        h->wl_cur_h = (double *)calloc(h->R, 8);
        h->wl_cur_f = (double *)calloc(RF, 8);
        h->wl_counter = (int64_t *)calloc(h->R, 8);
        for (int r = 0; r < h->R; ++r) h->wl_m[r] = cfg->wl_mod_factor;
    }
    *out = h;
    return 0;
}

void orc_mc_destroy(orc_mc *h) {
    if (!h) return;
    free(h->occ); free(h->work_f); free(h->work_i); free(h->features); free(h->enthalpy);
    free(h->temperature); free(h->seeds); free(h->nsteps); free(h->naccepted);
    free(h->last_accepted); free(h->natural); free(h->wl_entropy); free(h->wl_hist);
    free(h->wl_occur); free(h->wl_meanf); free(h->wl_m); free(h->wl_cur_h); free(h->wl_cur_f);
    free(h->wl_counter); free(h->bias);
    free(h);
}

int orc_mc_num_features(const orc_mc *h) { return h->F; }
int orc_mc_wl_num_levels(const orc_mc *h) { return h->L; }

static double dotp(const double *a, const double *b, int n) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* Sampler.setup_sample (sampler/sampler.py:386-434): kernel.set_aux_state +
 * compute_initial_trace (kernel/base.py:345-365; wanglandau.py:290-300). */
int orc_mc_set_state(orc_mc *h, const int32_t *occ, const uint64_t *seeds,
                     const double *temperature, int reset_aux) {
    size_t RN = (size_t)h->R * h->N;
    memcpy(h->occ, occ, RN * 4);
    memcpy(h->work_f, occ, RN * 4);
    memcpy(h->work_i, occ, RN * 4);
    for (int r = 0; r < h->R; ++r) {
        h->temperature[r] = temperature ? temperature[r] : 0.0;
        if (reset_aux) {
            h->seeds[r] = seeds ? seeds[r] : (uint64_t)r;
            h->nsteps[r] = 0;
            h->naccepted[r] = 0;
        }
        h->last_accepted[r] = 1; /* trace.accepted = True initially (base.py:364) */
        double *f = h->features + (size_t)r * h->F;
        orc_feature_vector(h->t, h->occ + (size_t)r * h->N, f);
        h->enthalpy[r] = dotp(h->natural, f, h->F);
        h->bias[r] = orc_compute_bias(h->t, h->occ + (size_t)r * h->N);
        if (h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU) {
            memcpy(h->wl_cur_f + (size_t)r * h->F, f, sizeof(double) * h->F);
            h->wl_cur_h[r] = h->enthalpy[r];
            if (reset_aux) {
                size_t L = h->L;
                memset(h->wl_entropy + r * L, 0, L * 8);
                memset(h->wl_hist + r * L, 0, L * 8);
                memset(h->wl_occur + r * L, 0, L * 8);
                memset(h->wl_meanf + r * L * h->F, 0, L * h->F * 8);
                h->wl_m[r] = h->cfg.wl_mod_factor;
                h->wl_counter[r] = 0;
            }
        }
    }
    return 0;
}

int orc_mc_set_temperature(orc_mc *h, const double *temperature) {
    for (int r = 0; r < h->R; ++r) h->temperature[r] = temperature[r];
    return 0;
}

int orc_mc_get_state(orc_mc *h, int32_t *occ, double *features, double *enthalpy,
                     uint64_t *n_accepted, uint64_t *n_steps, uint8_t *last_accepted) {
    if (occ) memcpy(occ, h->occ, (size_t)h->R * h->N * 4);
    if (features) memcpy(features, h->features, (size_t)h->R * h->F * 8);
    if (enthalpy) memcpy(enthalpy, h->enthalpy, (size_t)h->R * 8);
    if (n_accepted) memcpy(n_accepted, h->naccepted, (size_t)h->R * 8);
    if (n_steps) memcpy(n_steps, h->nsteps, (size_t)h->R * 8);
    if (last_accepted) memcpy(last_accepted, h->last_accepted, (size_t)h->R);
    return 0;
}

/* smolmc_set_counters: the position of every walker in its random stream / its accept counter */
int orc_mc_set_counters(orc_mc *h, const uint64_t *n_steps, const uint64_t *n_accepted) {
    if (n_steps) {
        memcpy(h->nsteps, n_steps, (size_t)h->R * 8);
        if (h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU) /* the check period counts the same steps */
            for (int r = 0; r < h->R; ++r) h->wl_counter[r] = (int64_t)n_steps[r];
    }
    if (n_accepted) memcpy(h->naccepted, n_accepted, (size_t)h->R * 8);
    return 0;
}

int orc_mc_get_bias(orc_mc *h, double *bias) {
    if (!h->t->bias_type) return 1;
    memcpy(bias, h->bias, (size_t)h->R * 8);
    return 0;
}

int orc_mc_get_wl(orc_mc *h, double *entropy, int64_t *hist, int64_t *occur, double *meanf,
                  double *mod_factor) {
    if (h->cfg.kernel_type != SMOLMC_KERNEL_WANGLANDAU) return 1;
    size_t RL = (size_t)h->R * h->L;
    if (entropy) memcpy(entropy, h->wl_entropy, RL * 8);
    if (hist) memcpy(hist, h->wl_hist, RL * 8);
    if (occur) memcpy(occur, h->wl_occur, RL * 8);
    if (meanf) memcpy(meanf, h->wl_meanf, RL * h->F * 8);
    if (mod_factor) memcpy(mod_factor, h->wl_m, (size_t)h->R * 8);
    return 0;
}

/* ---- ushers on the engine stream (distribution of mcusher.py:146-200) ----- */
/* Random words of one step: W(step, block, j).  block 0: w0 sublattice,
 * (w2,w3) acceptance uniform; the SITE of step k is drawn from W(k-1, 0, 1), i.e.
 * from the previous step's block (so a wavefront can prefetch the next step's index
 * row while it evaluates the current one).  block 1+: flip species word (block 1,
 * word 0) or the swap candidate sequence (see propose_step). */
typedef struct {
    uint32_t key[2];
    uint64_t step;
} rng_ctx;

static void rng_block(const rng_ctx *g, uint32_t block, uint32_t w[4]) {
    uint32_t ctr[4] = {(uint32_t)g->step, (uint32_t)(g->step >> 32), block, 0u};
    orc_philox4x32(ctr, g->key, w);
}

static int pick_sublattice(const smolmc_tables *t, uint32_t w0) {
    /* MCUsher.get_random_sublattice (mcusher.py:146-148): choice(active, p=probs) */
    if (t->n_sublattices == 1) return 0;
    double x = (double)w0 * (1.0 / 4294967296.0), c = 0;
    for (int s = 0; s < t->n_sublattices; ++s) {
        c += t->sub_probs[s];
        if (x < c) return s;
    }
    return t->n_sublattices - 1;
}

static int propose_swap_in(const orc_mc *h, const int32_t *occ, const rng_ctx *g, int sl,
                           uint32_t w_site, int32_t *flips);

/* returns number of flips (0, 1 or 2) written to flips[4] */
static int propose_step(const orc_mc *h, const int32_t *occ, const rng_ctx *g, uint32_t w0[4],
                        uint32_t w_site, int32_t flips[4]) {
    const smolmc_tables *t = h->t;
    int sl = pick_sublattice(t, w0[0]);
    const int32_t *sites = t->sub_active_sites + t->sub_site_ptr[sl];
    uint32_t nact = (uint32_t)(t->sub_site_ptr[sl + 1] - t->sub_site_ptr[sl]);
    int site1 = sites[mulhi32(w_site, nact)];
    if (h->cfg.step_type == SMOLMC_STEP_FLIP) {
        /* Flip.propose_step (mcusher.py:154-170): uniform among the other codes */
        const int32_t *codes = t->sub_codes + t->sub_code_ptr[sl];
        uint32_t nc = (uint32_t)(t->sub_code_ptr[sl + 1] - t->sub_code_ptr[sl]);
        uint32_t w1[4];
        rng_block(g, 1, w1);
        uint32_t k = mulhi32(w1[0], nc - 1);
        int cur = occ[site1], code = -1;
        for (uint32_t c = 0, seen = 0; c < nc; ++c) {
            if (codes[c] == cur) continue;
            if (seen == k) { code = codes[c]; break; }
            seen++;
        }
        flips[0] = site1; flips[1] = code;
        return 1;
    }
    return propose_swap_in(h, occ, g, sl, w_site, flips);
}

/* Swap.propose_step (mcusher.py:176-200): site2 uniform over active sites of the sublattice
 * whose species differs; realised as rejection sampling over the candidate sequence
 * (identical distribution). */
static int propose_swap_in(const orc_mc *h, const int32_t *occ, const rng_ctx *g, int sl,
                           uint32_t w_site, int32_t *flips) {
    const smolmc_tables *t = h->t;
    const int32_t *sites = t->sub_active_sites + t->sub_site_ptr[sl];
    uint32_t nact = (uint32_t)(t->sub_site_ptr[sl + 1] - t->sub_site_ptr[sl]);
    int site1 = sites[mulhi32(w_site, nact)];
    int sp1 = occ[site1];
    /* first 12 candidates: c_t = W(step, 1 + t % 3, t / 3)  (blocks 1..3, word-major:
     * the order the wavefront tests them, three lanes per LDS read) */
    {
        uint32_t w[3][4];
        for (int b = 0; b < 3; ++b) rng_block(g, 1 + b, w[b]);
        for (int t = 0; t < 12; ++t) {
            int site2 = sites[mulhi32(w[t % 3][t / 3], nact)];
            if (occ[site2] != sp1) {
                flips[0] = site1; flips[1] = occ[site2];
                flips[2] = site2; flips[3] = sp1;
                return 2;
            }
        }
    }
    /* then c_t = W(step, 4 + (t - 12) / 4, (t - 12) % 4), t >= 12 */
    for (uint32_t blk = 4;; ++blk) {
        uint32_t w[4];
        rng_block(g, blk, w);
        for (int j = 0; j < 4; ++j) {
            int site2 = sites[mulhi32(w[j], nact)];
            if (occ[site2] != sp1) {
                flips[0] = site1; flips[1] = occ[site2];
                flips[2] = site2; flips[3] = sp1;
                return 2;
            }
        }
        if (blk == 4 + 63 || ((blk - 4) & 4095u) == 4095u) { /* swap_options.size == 0 -> empty step (:197-199) */
            int any = 0;
            for (uint32_t a = 0; a < nact; ++a)
                if (occ[sites[a]] != sp1) { any = 1; break; }
            if (!any) return 0;
        }
    }
}

/* ---- TableFlip (smol/moca/kernel/mcusher.py:397-711) on the engine stream ---------- */
/* species counts over the ACTIVE sites, "counts" format over the active sublattices
 * (occu_to_counts, smol/moca/occu_utils.py:103-135) */
static void table_counts(const smolmc_tables *t, const int32_t *occ, int *n) {
    int d = (int)t->sub_code_ptr[t->n_sublattices];
    for (int i = 0; i < d; ++i) n[i] = 0;
    for (int sl = 0; sl < t->n_sublattices; ++sl) {
        const int32_t *codes = t->sub_codes + t->sub_code_ptr[sl];
        int nc = (int)(t->sub_code_ptr[sl + 1] - t->sub_code_ptr[sl]);
        for (int64_t a = t->sub_site_ptr[sl]; a < t->sub_site_ptr[sl + 1]; ++a) {
            int v = occ[t->sub_active_sites[a]];
            for (int c = 0; c < nc; ++c)
                if (codes[c] == v) n[t->sub_code_ptr[sl] + c]++;
        }
    }
}

/* flip_weights_mask (smol/utils/math.py:832-867): sum of the weights of feasible directions */
static double table_masked_weights(const smolmc_tables *t, const int *n, double *mw) {
    int d = (int)t->sub_code_ptr[t->n_sublattices];
    double sum = 0;
    for (int idx = 0; idx < 2 * t->n_flip_vectors; ++idx) {
        const int32_t *row = t->flip_table + (size_t)(idx / 2) * d;
        int sgn = (idx & 1) ? -1 : 1, ok = 1;
        for (int i = 0; i < d && ok; ++i) {
            int sl = 0;
            while (t->sub_code_ptr[sl + 1] <= i) sl++;
            int max_n = (int)(t->sub_site_ptr[sl + 1] - t->sub_site_ptr[sl]); /* mcusher.py:497-501 */
            int v = n[i] + sgn * row[i];
            if (v < 0 || v > max_n) ok = 0;
        }
        mw[idx] = ok ? t->flip_weights[idx] : 0.0;
        sum += mw[idx];
    }
    return sum;
}

/* compute_log_priori_factor (mcusher.py:656-711) for direction idx at counts n */
static double table_log_priori(const smolmc_tables *t, const int *n, int idx, double sum_now,
                               const double *mw_now) {
    int d = (int)t->sub_code_ptr[t->n_sublattices];
    int n_next[SMOLMC_MAX_FLIP_DIMS] = {0};
    double mw_next[2 * SMOLMC_MAX_FLIP_VECTORS];
    const int32_t *row = t->flip_table + (size_t)(idx / 2) * d;
    int sgn = (idx & 1) ? -1 : 1;
    for (int i = 0; i < d; ++i) n_next[i] = n[i] + sgn * row[i];
    double sum_next = table_masked_weights(t, n_next, mw_next);
    double sw = t->swap_weight;
    double p_now = (1 - sw) * mw_now[idx] / sum_now;
    double p_next = (1 - sw) * mw_next[idx ^ 1] / sum_next;
    double lf = log(p_next / p_now);
    for (int i = 0; i < d; ++i) { /* ln(n_now!) - ln(n_next!) as sums of logs */
        int u = sgn * row[i];
        for (int k = 1; k <= u; ++k) lf -= log((double)(n[i] + k));
        for (int k = 0; k < -u; ++k) lf += log((double)(n[i] - k));
    }
    return lf;
}

/* TableFlip.propose_step (mcusher.py:553-639).  Stream: W(step,0,0) swap-or-table,
 * W(step,1,0) direction, W(step,1,1) sublattice of the swap branch, blocks 2-3 the
 * assignment draws, blocks 4+ the site candidate sequence.  Returns the number of flips;
 * *log_priori receives compute_log_priori_factor of the proposed step. */
static int propose_table_flip(const orc_mc *h, const int32_t *occ, const rng_ctx *g, uint32_t w0[4],
                              uint32_t w_site, int32_t *flips, double *log_priori) {
    const smolmc_tables *t = h->t;
    int d = (int)t->sub_code_ptr[t->n_sublattices];
    uint32_t w1[4];
    rng_block(g, 1, w1);
    *log_priori = 0.0;
    int n[SMOLMC_MAX_FLIP_DIMS];
    double mw[2 * SMOLMC_MAX_FLIP_VECTORS];
    double sumw = 0;
    int do_swap = (double)w0[0] * (1.0 / 4294967296.0) < t->swap_weight; /* mcusher.py:577-578 */
    if (!do_swap) {
        table_counts(t, occ, n);
        sumw = table_masked_weights(t, n, mw);
        if (!(sumw > 0)) do_swap = 1; /* no feasible flip: canonical swap only (:604-611) */
    }
    if (do_swap) return propose_swap_in(h, occ, g, pick_sublattice(t, w1[1]), w_site, flips);
    /* choose_section_from_partition (math.py:870-893) */
    double target = (double)w1[0] * (1.0 / 4294967296.0) * sumw, cum = 0;
    int idx = -1, last = -1;
    for (int i = 0; i < 2 * t->n_flip_vectors; ++i) {
        if (mw[i] <= 0) continue;
        last = i;
        cum += mw[i];
        if (target < cum) { idx = i; break; }
    }
    if (idx < 0) idx = last;
    const int32_t *row = t->flip_table + (size_t)(idx / 2) * d;
    int sgn = (idx & 1) ? -1 : 1;
    *log_priori = table_log_priori(t, n, idx, sumw, mw);
    uint32_t tcand = 0, qdraw = 0, wc[4], wd[4];
    uint32_t wc_blk = 0xffffffffu, wd_blk = 0xffffffffu;
    int nfl = 0;
    for (int sl = 0; sl < t->n_sublattices; ++sl) {
        const int32_t *sites = t->sub_active_sites + t->sub_site_ptr[sl];
        uint32_t nact = (uint32_t)(t->sub_site_ptr[sl + 1] - t->sub_site_ptr[sl]);
        const int32_t *codes = t->sub_codes + t->sub_code_ptr[sl];
        int nc = (int)(t->sub_code_ptr[sl + 1] - t->sub_code_ptr[sl]);
        int base = (int)t->sub_code_ptr[sl];
        int collected[SMOLMC_MAX_STEP_FLIPS], ncol = 0;
        for (int c = 0; c < nc; ++c) { /* depleted species: pick -u sites without replacement */
            int u = sgn * row[base + c];
            for (int k = 0; k < -u; ++k) {
                for (;;) {
                    uint32_t blk = 4u + tcand / 4u;
                    if (blk != wc_blk) { rng_block(g, blk, wc); wc_blk = blk; }
                    int site = sites[mulhi32(wc[tcand % 4u], nact)];
                    tcand++;
                    if (occ[site] != codes[c]) continue;
                    int dup = 0;
                    for (int z = 0; z < ncol; ++z) dup |= collected[z] == site;
                    if (dup) continue;
                    collected[ncol++] = site;
                    break;
                }
            }
        }
        for (int c = 0; c < nc; ++c) { /* enriched species: random assignment (:627-631) */
            int u = sgn * row[base + c];
            for (int k = 0; k < u; ++k) {
                uint32_t blk = 2u + qdraw / 4u;
                if (blk != wd_blk) { rng_block(g, blk, wd); wd_blk = blk; }
                uint32_t rr = mulhi32(wd[qdraw % 4u], (uint32_t)ncol);
                qdraw++;
                flips[2 * nfl] = collected[rr];
                flips[2 * nfl + 1] = codes[c];
                nfl++;
                for (int z = (int)rr; z + 1 < ncol; ++z) collected[z] = collected[z + 1];
                ncol--;
            }
        }
    }
    return nfl;
}

/* one MC step of walker r given its proposal; follows
 * StandardSingleStepMixin.single_step (kernel/base.py:145-166),
 * MCKernel._compute_step_trace (:291-311), Metropolis _accept_step
 * (metropolis.py:31-49) / WangLandau (wanglandau.py:186-266), _do_accept_step
 * (base.py:327-343) and the trace accumulation of Sampler.sample
 * (sampler/sampler.py:199-207). */
static int do_step(orc_mc *h, int r, const int32_t *flips, int nflips, double u, double log_priori,
                   double *dfeat) {
    const smolmc_tables *t = h->t;
    int32_t *occ = h->occ + (size_t)r * h->N;
    int F = h->F;
    orc_feature_vector_change(t, occ, flips, nflips, h->work_f + (size_t)r * h->N,
                              h->work_i + (size_t)r * h->N, dfeat);
    double dH = dotp(h->natural, dfeat, F); /* base.py:303-306 */
    double dB = t->bias_type ? orc_compute_bias_change(t, occ, flips, nflips) : 0.0; /* :307-311 */
    int accepted;
    if (h->cfg.kernel_type == SMOLMC_KERNEL_METROPOLIS) {
        double beta = 1.0 / (ORC_KB * h->temperature[r]); /* base.py:398 */
        double exponent = -beta * dH + log_priori;         /* metropolis.py:41-42 */
        if (t->bias_type) exponent += dB;                  /* :43-44 */
        accepted = exponent >= 0 ? 1 : (exponent > log(u)); /* :46-48 */
    } else {
        double emin = h->cfg.wl_min_enthalpy, emax = h->cfg.wl_max_enthalpy, bsz = h->cfg.wl_bin_size;
        double cur = h->wl_cur_h[r];
        double new_h = cur + dH; /* wanglandau.py:188 */
        if (new_h < emin || new_h >= emax) { /* :191 */
            accepted = 0;
        } else {
            long b = (long)floordiv_exact(cur - emin, bsz); /* :187,:180 */
            long nb = (long)floordiv_exact(new_h - emin, bsz);
            const double *S = h->wl_entropy + (size_t)r * h->L;
            double exponent = S[b] - S[nb] + log_priori; /* :197-198 */
            accepted = exponent >= 0 ? 1 : (exponent > log(u));
        }
    }
    if (accepted) {
        for (int f = 0; f < nflips; ++f) { /* base.py:338-339 */
            int site = flips[2 * f], code = flips[2 * f + 1];
            occ[site] = code;
            h->work_f[(size_t)r * h->N + site] = code;
            h->work_i[(size_t)r * h->N + site] = code;
        }
        double *feat = h->features + (size_t)r * F;
        for (int i = 0; i < F; ++i) feat[i] += dfeat[i]; /* sampler.py:204-207 */
        h->enthalpy[r] += dH;
        h->bias[r] += dB;
        h->naccepted[r]++;
        if (h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU) { /* wanglandau.py:216-218 */
            double *cf = h->wl_cur_f + (size_t)r * F;
            for (int i = 0; i < F; ++i) cf[i] += dfeat[i];
            h->wl_cur_h[r] += dH;
        }
    }
    if (h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU) { /* _do_post_step :222-266 */
        double emin = h->cfg.wl_min_enthalpy, bsz = h->cfg.wl_bin_size;
        double bq = floordiv_exact(h->wl_cur_h[r] - emin, bsz);
        size_t L = h->L;
        if (bq >= 0 && bq < (double)L) { /* :231 */
            long b = (long)bq;
            h->wl_counter[r]++;
            int64_t total = h->wl_occur[r * L + b];
            double *mf = h->wl_meanf + ((size_t)r * L + b) * F;
            const double *cf = h->wl_cur_f + (size_t)r * F;
            double inv = 1.0 / (double)(total + 1);
            for (int i = 0; i < F; ++i) mf[i] = inv * (cf[i] + (double)total * mf[i]); /* :235-239 */
            if (h->wl_counter[r] % h->cfg.wl_update_period == 0) { /* :241-245 */
                h->wl_entropy[r * L + b] += h->wl_m[r];
                h->wl_hist[r * L + b] += 1;
                h->wl_occur[r * L + b] += 1;
            }
        }
        if (h->cfg.wl_check_period != 0 && h->wl_counter[r] % h->cfg.wl_check_period == 0) { /* :253-264; 0 = no check (smolmc.h) */
            const double *S = h->wl_entropy + r * L;
            int64_t *H = h->wl_hist + r * L;
            long cnt = 0;
            double sum = 0;
            for (size_t i = 0; i < L; ++i)
                if (S[i] > 0) { cnt++; sum += (double)H[i]; }
            if (cnt >= 2) {
                double thr = h->cfg.wl_flatness * (sum / (double)cnt);
                int flat = 1;
                for (size_t i = 0; i < L; ++i)
                    if (S[i] > 0 && !((double)H[i] > thr)) { flat = 0; break; }
                if (flat) {
                    memset(H, 0, L * 8);
                    h->wl_m[r] = h->wl_m[r] / h->cfg.wl_mod_divisor;
                }
            }
        }
    }
    h->last_accepted[r] = (uint8_t)accepted;
    h->nsteps[r]++;
    return accepted;
}

/* native mode: engine RNG.  OpenMP over walkers (they are independent,
 * sampler/sampler.py:436-440). */
int orc_mc_run(orc_mc *h, int64_t nsteps) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < h->R; ++r) {
        double *dfeat = (double *)malloc(sizeof(double) * h->F);
        rng_ctx g;
        g.key[0] = (uint32_t)h->seeds[r];
        g.key[1] = (uint32_t)(h->seeds[r] >> 32);
        uint32_t wprev[4];
        g.step = h->nsteps[r] - 1; /* wraps to 2^64-1 for the very first step */
        rng_block(&g, 0, wprev);
        for (int64_t k = 0; k < nsteps; ++k) {
            g.step = h->nsteps[r];
            uint32_t w0[4];
            int32_t flips[2 * SMOLMC_MAX_STEP_FLIPS];
            for (int z = 0; z < 2 * SMOLMC_MAX_STEP_FLIPS; ++z) flips[z] = -1;
            double lp = 0.0;
            rng_block(&g, 0, w0);
            int nf;
            if (h->cfg.step_type == SMOLMC_STEP_TABLE_FLIP)
                nf = propose_table_flip(h, h->occ + (size_t)r * h->N, &g, w0, wprev[1], flips, &lp);
            else
                nf = propose_step(h, h->occ + (size_t)r * h->N, &g, w0, wprev[1], flips);
            do_step(h, r, flips, nf, u53(w0[2], w0[3]), lp, dfeat);
            memcpy(wprev, w0, sizeof(w0));
        }
        free(dfeat);
    }
    return 0;
}

/* TableFlip.compute_log_priori_factor (mcusher.py:656-711) of an arbitrary step: _get_flip_id
 * (:641-654) over delta_counts_from_step (occu_utils.py:131-168, a site may appear several times:
 * the running occupancy is followed), then the factor at the counts of `occ`.  *status: 0 ok,
 * 1 = step is not in the flip table (the reference raises ValueError, :673-674), 2 = a flip on an
 * inactive site / impossible code (occu_utils.py:160-163). */
double orc_table_step_log_priori(const smolmc_tables *t, const int32_t *occ, const int32_t *flips,
                                 int nflips, int *status) {
    int d = (int)t->sub_code_ptr[t->n_sublattices];
    int dn[SMOLMC_MAX_FLIP_DIMS], n[SMOLMC_MAX_FLIP_DIMS];
    double mw[2 * SMOLMC_MAX_FLIP_VECTORS];
    *status = 0;
    for (int i = 0; i < d; ++i) dn[i] = 0;
    for (int f = 0; f < nflips; ++f) {
        int site = flips[2 * f], code = flips[2 * f + 1];
        int cur = occ[site];
        for (int g = 0; g < f; ++g)
            if (flips[2 * g] == site) cur = flips[2 * g + 1]; /* occu_now[site] = code */
        int dim_ori = -1, dim_nex = -1;
        for (int sl = 0; sl < t->n_sublattices; ++sl) {
            int in_sl = 0;
            for (int64_t a = t->sub_site_ptr[sl]; a < t->sub_site_ptr[sl + 1] && !in_sl; ++a)
                in_sl = t->sub_active_sites[a] == site;
            if (!in_sl) continue;
            for (int64_t c = t->sub_code_ptr[sl]; c < t->sub_code_ptr[sl + 1]; ++c) {
                if (t->sub_codes[c] == cur) dim_ori = (int)c;
                if (t->sub_codes[c] == code) dim_nex = (int)c;
            }
        }
        if (dim_ori < 0 || dim_nex < 0) { *status = 2; return 0.0; }
        dn[dim_ori] -= 1;
        dn[dim_nex] += 1;
    }
    int zero = 1;
    for (int i = 0; i < d; ++i) zero &= dn[i] == 0;
    if (zero) return 0.0; /* canonical swap: fid = -1 (:668-671) */
    int idx = -1;
    for (int v = 0; v < t->n_flip_vectors && idx < 0; ++v) {
        const int32_t *row = t->flip_table + (size_t)v * d;
        int eqp = 1, eqm = 1;
        for (int i = 0; i < d; ++i) { eqp &= row[i] == dn[i]; eqm &= -row[i] == dn[i]; }
        if (eqp) idx = 2 * v;
        else if (eqm) idx = 2 * v + 1;
    }
    if (idx < 0) { *status = 1; return 0.0; }
    table_counts(t, occ, n);
    double sum = table_masked_weights(t, n, mw);
    return table_log_priori(t, n, idx, sum, mw);
}

/* replay mode: proposals + uniforms from the host (reference Generator order); records of
 * SMOLMC_STEP_ROW ints, log_priori as smolmc_replay documents it.  Returns 0, or 3 when a step of a
 * TableFlip handle is not in the flip table / touches an inactive site. */
int orc_mc_replay(orc_mc *h, int64_t nsteps, const int32_t *steps, const double *uniforms,
                  const double *log_priori, uint8_t *accepted_out, double *enthalpy_out,
                  double *log_priori_out) {
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int r = 0; r < h->R; ++r) {
        double *dfeat = (double *)malloc(sizeof(double) * h->F);
        for (int64_t k = 0; k < nsteps && !bad; ++k) {
            const int32_t *st = steps + ((size_t)r * nsteps + k) * SMOLMC_STEP_ROW;
            int nf = 0;
            while (nf < SMOLMC_MAX_STEP_FLIPS && st[2 * nf] >= 0) nf++;
            double u = uniforms[(size_t)r * nsteps + k];
            if (isnan(u)) u = 0.0; /* not drawn by the reference => it accepted without a draw */
            double lp = log_priori ? log_priori[(size_t)r * nsteps + k] : NAN;
            if (isnan(lp)) {
                lp = 0.0;
                if (h->cfg.step_type == SMOLMC_STEP_TABLE_FLIP) {
                    int status = 0;
                    lp = orc_table_step_log_priori(h->t, h->occ + (size_t)r * h->N, st, nf, &status);
                    if (status) { bad |= 1; break; }
                }
            }
            int a = do_step(h, r, st, nf, u, lp, dfeat);
            if (accepted_out) accepted_out[(size_t)r * nsteps + k] = (uint8_t)a;
            if (enthalpy_out)
                enthalpy_out[(size_t)r * nsteps + k] =
                    h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU ? h->wl_cur_h[r] : h->enthalpy[r];
            if (log_priori_out) log_priori_out[(size_t)r * nsteps + k] = lp;
        }
        free(dfeat);
    }
    return bad ? 3 : 0;
}

/* proposal only (for usher statistics tests, tests/test_moca/test_mcushers.py:124-196) */
int orc_mc_propose(orc_mc *h, int r, uint64_t step, int32_t *flips /* 2*SMOLMC_MAX_STEP_FLIPS */,
                   double *log_priori) {
    rng_ctx g;
    g.key[0] = (uint32_t)h->seeds[r];
    g.key[1] = (uint32_t)(h->seeds[r] >> 32);
    uint32_t w0[4], wprev[4];
    g.step = step - 1;
    rng_block(&g, 0, wprev);
    g.step = step;
    rng_block(&g, 0, w0);
    for (int z = 0; z < 2 * SMOLMC_MAX_STEP_FLIPS; ++z) flips[z] = -1;
    double lp = 0.0;
    int nf;
    if (h->cfg.step_type == SMOLMC_STEP_TABLE_FLIP)
        nf = propose_table_flip(h, h->occ + (size_t)r * h->N, &g, w0, wprev[1], flips, &lp);
    else
        nf = propose_step(h, h->occ + (size_t)r * h->N, &g, w0, wprev[1], flips);
    if (log_priori) *log_priori = lp;
    return nf;
}
