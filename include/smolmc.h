/*
 * smolmc.h -- C-ABI of the MI355X-native ensemble Monte-Carlo engine.
 *
 * Drop-in boundary for ONE hot path of CederGroupHub/smol: the
 * per-flip local correlation / cluster-interaction delta, the Ewald single-flip
 * delta and the Metropolis / Wang-Landau accept step, batched over independent
 * replica walkers.  The reference has no C ABI of its own for this path: its
 * native entry points are the Cython cpdef methods listed below, called from
 * Python objects.  Each entry point here names the reference interface it
 * replaces (paths relative to the reference checkout).
 *
 * Conventions (same as the reference):
 *   - occupancies are int32, C-contiguous, values = species codes per site
 *     (smol/moca/processor/base.py:228-237);
 *   - index tables are int32 C-contiguous, tensors / outputs are float64;
 *   - feature vectors handed back are EXTENSIVE (x size) like
 *     Processor.compute_feature_vector[_change] (processor/expansion.py:184,231);
 *   - the engine copies every table at smolmc_create(); the caller keeps
 *     ownership of all buffers it passes; outputs are caller-allocated;
 *   - every function returns 0 on success, non-zero on error; the message is
 *     available from smolmc_last_error() (the Python shim raises
 *     ValueError / RuntimeError like the reference, expansion.py:186-188).
 *   - a handle is not thread-safe; one host thread drives one device.
 *
 * All pointers in this header are HOST pointers unless the name says _dev.
 */
#ifndef SMOLMC_H
#define SMOLMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMOLMC_ABI_VERSION 8

/* Status codes: 0 = ok; SMOLMC_ERR (1) = any failure, message in smolmc_last_error(); the codes below name failures a
 * caller may want to handle without parsing the message (the message is set for them too). */
#define SMOLMC_ERR 1
#define SMOLMC_ERR_RING_FULL 2 /* smolmc_run_sampled: both ring slots hold blocks that were not fetched */

#define SMOLMC_BIAS_NONE 0
#define SMOLMC_BIAS_FUGACITY 1
#define SMOLMC_BIAS_SQUARE_CHARGE 2
#define SMOLMC_BIAS_SQUARE_HYPERPLANE 3
#define SMOLMC_MAX_BIAS_ROWS 4 /* hyperplanes of a SquareHyperplaneBias */
#define SMOLMC_MAX_CLUSTER_SITES 6 /* largest cluster (sites per cluster) supported */

/* feature_mode */
#define SMOLMC_FEATURES_CORRELATIONS 0 /* ClusterExpansionProcessor  (expansion.py:39)  */
#define SMOLMC_FEATURES_INTERACTIONS 1 /* ClusterDecompositionProcessor (expansion.py:243) */
/* kernel_type (smol/moca/kernel/__init__.py:35-56 mckernel_factory names) */
#define SMOLMC_KERNEL_METROPOLIS 0 /* kernel/metropolis.py:52 */
#define SMOLMC_KERNEL_WANGLANDAU 1 /* kernel/wanglandau.py:17 */
/* step_type (smol/moca/kernel/mcusher.py) */
#define SMOLMC_STEP_FLIP 0 /* Flip  mcusher.py:151 */
#define SMOLMC_STEP_SWAP 1 /* Swap  mcusher.py:173 */
#define SMOLMC_STEP_TABLE_FLIP 2 /* TableFlip mcusher.py:397 (charge-neutral semigrand) */
#define SMOLMC_MAX_STEP_FLIPS 8  /* most single-site flips in one TableFlip step */
/* One MC step as the ushers return it (a list of (site, code) tuples, mcusher.py:104-116), as a fixed
 * record of SMOLMC_STEP_ROW int32: (site_0, code_0, ..., site_7, code_7), flips in the order the
 * reference applies them (sequential-flip semantics, expansion.py:217-229); the record ends at the
 * first site < 0 (-1 = no flip; an all -1 record is the empty step of mcusher.py:197-199). */
#define SMOLMC_STEP_ROW (2 * SMOLMC_MAX_STEP_FLIPS)
#define SMOLMC_MAX_FLIP_VECTORS 32 /* rows of flip_table (2 x that many directions) */
#define SMOLMC_MAX_FLIP_DIMS 64    /* columns of flip_table: species over the active sublattices */

/*
 * Read-only model tables.  Flattened, caller-owned equivalents of the
 * containers in smol/utils/cluster/container.pyx (OrbitContainer :21,
 * IntArray2DContainer :287, FloatArray1DContainer :173) and struct.pxd:12-38.
 */
typedef struct smolmc_tables {
    int32_t num_sites;   /* N, length of an occupancy (Processor.num_sites) */
    int32_t size;        /* P, number of prim cells in the supercell (Processor.size) */
    int32_t num_orbits;  /* incl. empty cluster (ClusterSubspace.num_orbits) */
    int32_t num_corr;    /* incl. empty cluster (num_corr_functions) */
    int32_t max_species; /* largest number of species codes on any site */
    int32_t n_orb;       /* number of OrbitC records = num_orbits - 1 */

    /* OrbitC records, order of ClusterSubspace.orbits
     * (struct.pxd:33-38, get_orbit_data smol/utils/cluster/__init__.py:4-15) */
    const int32_t *orb_id;          /* [n_orb] orbit.id        (>= 1) */
    const int32_t *orb_bit_id;      /* [n_orb] orbit.bit_id    (>= 1) */
    const int32_t *orb_nsites;      /* [n_orb] I = tensor_indices.size */
    const int32_t *orb_nfunc;       /* [n_orb] K = correlation_tensors.size_r */
    const int32_t *orb_tensor_len;  /* [n_orb] correlation_tensors.size_c */
    const int32_t *orb_stride_off;  /* [n_orb] offset into tensor_indices[] */
    const int32_t *tensor_indices;  /* concatenated flat_tensor_indices (orbit.py:268-275) */
    const int64_t *orb_ctensor_off; /* [n_orb] offset into corr_tensors[] */
    const double *corr_tensors;     /* concatenated [K x len] row-major (orbit.py:251-266) */
    const int64_t *orb_itensor_off; /* [n_orb] offset into interaction_tensors[] */
    const double *interaction_tensors; /* concatenated [len] (expansion.py:186-201) */
    double offset;                  /* interaction of the empty cluster (evaluator.pxd:20) */

    /* full cluster-site tables: OrbitIndices (clusterspace.py:1329-1366) */
    const int64_t *full_off; /* [n_orb+1] offsets (int32 entries) into full_idx */
    const int32_t *full_idx; /* per orbit rows [J_full x I] */

    /* per-site reduced tables: LocalEvalData (processor/expansion.py:24-36,120-156) */
    const int64_t *site_ptr;  /* [N+1] local-record range of each site (empty if inactive) */
    const int32_t *loc_orbit; /* [n_loc] index into orb_* records */
    const double *loc_ratio;  /* [n_loc] cluster_ratio */
    const int32_t *loc_nrows; /* [n_loc] J_local */
    const int64_t *loc_off;   /* [n_loc] offset (int32 entries) into loc_idx */
    const int32_t *loc_idx;   /* rows [J_local x I] */

    /* natural parameters of the cluster-expansion part (Processor.coefs) */
    int32_t feature_mode;   /* SMOLMC_FEATURES_* */
    const double *ce_coefs; /* [num_corr] (mode 0) or [num_orbits] (mode 1) */

    /* EwaldProcessor tables (smol/moca/processor/ewald.py:76-101), optional */
    int32_t has_ewald;
    int32_t ewald_dim;         /* M */
    int32_t ewald_width;       /* columns of ewald_inds */
    const int32_t *ewald_inds; /* [N x ewald_width], -1 = vacancy / absent */
    const double *ewald_matrix; /* [M x M] */
    double ewald_coef;          /* coefficient of the Ewald feature (ensemble.py:191-199) */

    /* chemical-potential table (smol/moca/ensemble.py:90-99), optional */
    int32_t has_mu;
    int32_t mu_width;
    const double *mu_table; /* [N x mu_width]; natural parameter -1 appended last */

    /* active sublattices (smol/moca/sublattice.py:23; mcusher.py:55-57) */
    int32_t n_sublattices;
    const int64_t *sub_site_ptr;   /* [n_sub+1] ranges into sub_active_sites */
    const int32_t *sub_active_sites;
    const int64_t *sub_code_ptr;   /* [n_sub+1] ranges into sub_codes */
    const int32_t *sub_codes;      /* Sublattice.encoding */
    const double *sub_probs;       /* [n_sub] sublattice_probabilities (sum to 1) */

    /* Optional: charge of every Ewald index, [ewald_dim] (the oxidation states of
     * EwaldProcessor._ewald_structure, processor/ewald.py:76-78).  When given, the engine
     * checks that ewald_matrix[a][b] == q_a q_b G[site_a][site_b] off the diagonal (what
     * pymatgen's EwaldSummation builds) and, if it holds to 1e-12, evaluates single-flip
     * deltas from the N x N site kernel G (4x less memory traffic than two matrix rows);
     * otherwise it silently uses the dense rows.  NULL = dense. */
    const double *ewald_charges;

    /* TableFlip (smol/moca/kernel/mcusher.py:397-711), used with SMOLMC_STEP_TABLE_FLIP.
     * flip_table rows are flip vectors in "counts" format over the ACTIVE sublattices'
     * species, concatenated in sub_* order (the reference's dims of inactive sublattices
     * are always zero and are dropped); directions 2i / 2i+1 are +row i / -row i. */
    int32_t n_flip_vectors;      /* <= SMOLMC_MAX_FLIP_VECTORS */
    const int32_t *flip_table;   /* [n_flip_vectors x sum(codes of active sublattices)], <= SMOLMC_MAX_FLIP_DIMS columns */
    const double *flip_weights;  /* [2 x n_flip_vectors] (mcusher.py:519-538) */
    double swap_weight;          /* probability of a canonical Swap instead (mcusher.py:540) */

    /* MCBias (smol/moca/kernel/bias.py): an extra term delta_bias added to the Metropolis
     * exponent (metropolis.py:43-44) and traced as `bias` (kernel/base.py:307-311,362-363).
     * bias_table is [num_sites x bias_width], indexed [site][species code]:
     *   SMOLMC_BIAS_FUGACITY       FugacityBias._fu_table (bias.py:208-226): fugacity fractions,
     *                              1 where unused; bias = sum_sites log(table[site][occ])
     *   SMOLMC_BIAS_SQUARE_CHARGE  SquareChargeBias._c_table (bias.py:256-262): oxidation states,
     *                              0 where unused; bias = -bias_penalty * (sum_sites table)^2
     *   SMOLMC_BIAS_SQUARE_HYPERPLANE  SquareHyperplaneBias (bias.py:290-366): bias_rows tables
     *                              [bias_rows x num_sites x bias_width], table r holding
     *                              A[r][dim_id(site, code)] (the hyperplane normal's entry of the
     *                              species, get_dim_ids_table occu_utils.py:8-23; 0 where unused)
     *                              and bias_intercepts[r] = b[r];
     *                              bias = -bias_penalty * sum_r (sum_sites table_r - b_r)^2
     * Not allowed with Wang-Landau (wanglandau.py:127-128). */
    int32_t bias_type;           /* SMOLMC_BIAS_* */
    int32_t bias_width;
    const double *bias_table;
    double bias_penalty;         /* SquareChargeBias / SquareHyperplaneBias .penalty (> 0) */
    int32_t bias_rows;           /* hyperplanes (<= SMOLMC_MAX_BIAS_ROWS); 0 or 1 for the other types */
    const double *bias_intercepts; /* [bias_rows] (NULL = zeros) */
} smolmc_tables;

typedef struct smolmc_config {
    int32_t n_replicas;  /* walkers owned by this handle (Sampler nwalkers) */
    int32_t kernel_type; /* SMOLMC_KERNEL_* */
    int32_t step_type;   /* SMOLMC_STEP_* */
    int32_t device;      /* HIP device ordinal (ignored by the CPU oracle) */
    /* Wang-Landau parameters (kernel/wanglandau.py:26-39) */
    double wl_min_enthalpy, wl_max_enthalpy, wl_bin_size;
    double wl_flatness, wl_mod_factor, wl_mod_divisor; /* mod_update = m / divisor */
    int64_t wl_check_period, wl_update_period; /* check period 0: no device-side flatness check (the
                                                * caller checks and applies its own mod_update callable) */
} smolmc_config;

typedef struct smolmc_handle smolmc_handle;

/* ---- lifetime ----------------------------------------------------------- */
/* Replaces construction of Processor + Ensemble + one MCKernel per walker
 * (processor/expansion.py:61-163, ensemble.py:102-217, kernel/base.py:192-239). */
int smolmc_create(const smolmc_tables *tables, const smolmc_config *config,
                  smolmc_handle **out);
int smolmc_destroy(smolmc_handle *h);
const char *smolmc_last_error(void);
int smolmc_abi_version(void);

/* Which kernel family this handle dispatches to, as a short text ("lean nslot=2 mm=2 field=1" /
 * "general nslot=8 mm=2 field=1"): lets callers and tests assert that the intended path runs
 * (no reference counterpart). Writes at most n bytes including the terminator (give it 512).
 * A handle on mc_kernel / the universal kernel appends " | not lean: <the first condition of
 * smolmc_create that kept the model off the specialised kernels>".  "lazy-features" marks a handle
 * whose kernels carry the scalar features only (Ewald energy, chemical work) and whose cluster
 * features -- several correlation functions per orbit, evaluator.pyx:211-265, or more than 64 of
 * them -- are evaluated from the occupancies wherever this interface returns them
 * (smolmc_get_state, smolmc_get_samples*): same values, to rounding, as the step-by-step trace
 * of sampler/sampler.py:204-207. */
int smolmc_kernel_info(const smolmc_handle *h, char *buf, int n);

/* ---- sizes -------------------------------------------------------------- */
int smolmc_num_features(const smolmc_handle *h); /* len(ensemble.natural_parameters) */
int smolmc_wl_num_levels(const smolmc_handle *h); /* len(np.arange(min,max,bin)) wanglandau.py:107 */
/* natural parameters vector (Ensemble.natural_parameters, ensemble.py:25,61-65) */
int smolmc_natural_parameters(const smolmc_handle *h, double *out /*F*/);

/* ---- state: Sampler.setup_sample + MCKernel.compute_initial_trace ------- */
/* (sampler/sampler.py:386-434, kernel/base.py:345-365, wanglandau.py:290-300).
 * occ [R x N] int32, seeds [R], temperature [R] (Kelvin; ignored by WL).
 * Computes the initial features / enthalpy of every walker on the device.
 * reset_aux != 0: fresh kernels -- per-walker RNG counters (= step counters), accept
 * counters and Wang-Landau aux arrays are reset.  reset_aux == 0: continuation --
 * counters and WL arrays are kept (a kernel's Generator and aux state persist
 * across Sampler.run calls in the reference, sampler.py:254-262), seeds are ignored. */
int smolmc_set_state(smolmc_handle *h, const int32_t *occ, const uint64_t *seeds,
                     const double *temperature, int reset_aux);
/* ThermalKernelMixin.temperature setter (kernel/base.py:418-422), per walker */
int smolmc_set_temperature(smolmc_handle *h, const double *temperature /*R*/);
/* any output pointer may be NULL */
int smolmc_get_state(smolmc_handle *h, int32_t *occ /*RxN*/, double *features /*RxF*/,
                     double *enthalpy /*R*/, uint64_t *n_accepted /*R*/,
                     uint64_t *n_steps /*R*/, uint8_t *last_accepted /*R*/);
/* WangLandau trace extras (wanglandau.py:247-251,268-288); NULLs allowed */
/* trace.bias of every walker (kernel/base.py:362-363 + accumulated delta_trace.bias);
 * error when the model has no bias term */
int smolmc_get_bias(smolmc_handle *h, double *bias /*R*/);
int smolmc_get_wl(smolmc_handle *h, double *entropy /*RxL*/, int64_t *histogram /*RxL*/,
                  int64_t *occurrences /*RxL*/, double *mean_features /*RxLxF*/,
                  double *mod_factor /*R*/);

/* Restore of a kernel's auxiliary state in a fresh handle / process: the inverse of smolmc_get_wl
 * (WangLandau._entropy, _histogram, _occurrences, _mean_features, _m, wanglandau.py:107-122, which
 * set_aux_state resets, :290-300; the reference's SampleContainer carries an `aux_checkpoint`
 * placeholder it never fills, sampler/container.py:89,539-541).  Arrays as smolmc_get_wl returns
 * them; NULL = leave as it is.  Also what a host-side `mod_update` callable needs
 * (wanglandau.py:100-105): download, apply, upload. */
int smolmc_set_wl(smolmc_handle *h, const double *entropy /*RxL*/, const int64_t *histogram /*RxL*/,
                  const int64_t *occurrences /*RxL*/, const double *mean_features /*RxLxF*/,
                  const double *mod_factor /*R*/);
/* Step / accept counters of every walker.  n_steps is the walker's position in its random stream
 * (Philox counter) and, for Wang-Landau, the step counter the check period refers to
 * (WangLandau._steps_counter).  Resume in a fresh handle / process, bit for bit:
 * smolmc_set_state(occ, seeds, T, reset_aux = 1) -- the seeds are only taken with reset_aux != 0 --
 * then smolmc_set_counters and, for Wang-Landau, smolmc_set_wl.  NULL = leave as it is. */
int smolmc_set_counters(smolmc_handle *h, const uint64_t *n_steps /*R*/, const uint64_t *n_accepted /*R*/);

/* ---- the hot path -------------------------------------------------------- */
/* Advance every walker nsteps MC steps: the body of Sampler.sample
 * (sampler/sampler.py:195-208) = MCUsher.propose_step + compute_feature_vector_change
 * + accept + in-place update + trace accumulation, engine RNG (Philox4x32-10
 * keyed by the walker seed, counter = step index).
 * Asynchronous on the handle's stream; smolmc_sync() or any get_* waits. */
int smolmc_run(smolmc_handle *h, int64_t nsteps);
int smolmc_sync(smolmc_handle *h);
/* Device-side thinning: advance nsamples x thin_by steps and record, after every thin_by
 * steps, one row per walker -- what Sampler.sample yields and
 * SampleContainer.save_sampled_trace stores (sampler/sampler.py:195-210,
 * container.py:384-397): enthalpy, features, the accept flag of the last step of the
 * block and, as `flags` asks (SMOLMC_SAMPLE_*): the occupancy; trace.bias (kernel/base.py:307-311,
 * 362-363); the Wang-Landau trace (wanglandau.py:247-251: entropy, histogram, occurrences,
 * cumulative mean features, mod_factor of every walker AT that sample).
 *
 * A sample ring of two slots (ABI 7): the call returns as soon as the launches are queued; when they
 * finish the block is copied to a pinned host mirror on a stream of its own, so that the download of
 * block k overlaps the kernel of block k + 1 (the reference streams samples while it runs,
 * sampler/sampler.py:286-291, container.py:420-437).  The intended sequence is
 *     run_sampled(0); run_sampled(1); get_samples -> block 0; run_sampled(2); get_samples -> block 1; ...
 * smolmc_get_samples* deliver the OLDEST block not yet delivered (waiting for its copy only, not for
 * younger launches); with none pending they deliver the newest block again.  A third
 * smolmc_run_sampled while two blocks are pending is REFUSED with SMOLMC_ERR_RING_FULL and changes nothing
 * (ABI 8; ABI 7 silently dropped the older block) -- fetch or smolmc_discard_samples first.  A call that fails
 * for any other reason (arguments, an allocation, a launch) leaves the ring as it was: the slot is taken only
 * when every launch of the block has been queued.  Nothing is allocated per call once the slots have grown
 * to the block size. */
#define SMOLMC_SAMPLE_OCCUPANCY 1 /* flags bit 0 */
#define SMOLMC_SAMPLE_BIAS 2      /* flags bit 1: trace.bias (error without an MCBias term) */
#define SMOLMC_SAMPLE_WL 4        /* flags bit 2: the Wang-Landau trace (error on a Metropolis handle) */
int smolmc_run_sampled(smolmc_handle *h, int64_t nsamples, int64_t thin_by, int flags);
/* The ring as the next smolmc_get_samples* call sees it (ABI 8): *n_pending = blocks queued and not fetched
 * (0..2), *nsamples / *flags = sample count and SMOLMC_SAMPLE_* flags of the block that call would deliver
 * (0 / 0 when the ring is empty) -- get_samples takes no array lengths, a caller that did not queue the
 * block itself sizes its arrays from here (container.py:409-413 allocates from nsamples the same way).
 * Any pointer may be NULL. */
int smolmc_pending_samples(smolmc_handle *h, int *n_pending, int64_t *nsamples, int *flags);
/* Forget every block of the ring, fetched or not: what a sampling loop that is abandoned half way
 * (an exception between two blocks, sampler/sampler.py:195-210 left early) calls so that the next loop
 * does not receive its blocks.  The walkers keep the state the queued launches leave them in. */
int smolmc_discard_samples(smolmc_handle *h);
/* [nsamples x R (x F | x N)], NULLs allowed */
int smolmc_get_samples(smolmc_handle *h, double *enthalpy, double *features,
                       uint8_t *accepted, int32_t *occupancy);
/* smolmc_get_samples with the occupancies as bytes, [nsamples x R x N] uint8 -- the device
 * ring's own format: a quarter of the PCIe traffic and host memory of the int32 form the
 * reference's trace uses (trace.occupancy is int32, sampler/sampler.py:411-415; codes are
 * < 256 by construction, smolmc_tables.n_codes). */
int smolmc_get_samples_u8(smolmc_handle *h, double *enthalpy, double *features,
                          uint8_t *accepted, uint8_t *occupancy);
/* ... and the columns ABI 7 added (each may be NULL; asking for one that was not recorded is an error):
 * bias [nsamples x R]; wl_entropy / wl_histogram / wl_occurrences [nsamples x R x L], wl_mean_features
 * [nsamples x R x L x F], wl_mod_factor [nsamples x R] (as smolmc_get_wl returns them, at every sample). */
int smolmc_get_samples_ex(smolmc_handle *h, double *enthalpy, double *features, uint8_t *accepted,
                          uint8_t *occupancy_u8, double *bias, double *wl_entropy, int64_t *wl_histogram,
                          int64_t *wl_occurrences, double *wl_mean_features, double *wl_mod_factor);
/* Same loop driven by host-provided proposals ("replay mode"): what
 * StandardSingleStepMixin.single_step (kernel/base.py:145-166) does after mcusher.propose_step
 * returned, with the numbers the reference's Generator produced.
 *   steps       [R x nsteps x SMOLMC_STEP_ROW] step records (see SMOLMC_STEP_ROW); every site must be
 *               a changeable site of an active sublattice and every code one of its species.
 *   uniforms    [R x nsteps] the number rng.random() returned in _accept_step (NaN if not drawn:
 *               the reference accepted without drawing, metropolis.py:46-48).
 *   log_priori  [R x nsteps] or NULL: mcusher.compute_log_priori_factor(occupancy, step) added to the
 *               exponent (metropolis.py:41-42, wanglandau.py:197-198).  NULL or a NaN entry: the
 *               engine's own value -- 0 for Flip / Swap handles (mcusher.py:118-134); for TableFlip
 *               handles TableFlip.compute_log_priori_factor (mcusher.py:656-711) evaluated on the
 *               device from the step and the walker's species counts; a step that is neither a
 *               canonical swap nor +-(a row of flip_table) fails like the reference's
 *               ValueError("Step ... is not in flip table.", :673-674).
 * Outputs (each may be NULL): accepted_out [R x nsteps]; enthalpy_out [R x nsteps] the walker's
 * enthalpy AFTER the step; log_priori_out [R x nsteps] the a-priori factor that entered the
 * exponent.  Bias terms (MCBias) and Wang-Landau state take part exactly as in smolmc_run. */
int smolmc_replay(smolmc_handle *h, int64_t nsteps, const int32_t *steps, const double *uniforms,
                  const double *log_priori, uint8_t *accepted_out, double *enthalpy_out,
                  double *log_priori_out);
/* elapsed device time of the last smolmc_run / smolmc_replay launch in ms
 * (HIP events recorded on the launch stream) */
int smolmc_last_kernel_ms(smolmc_handle *h, float *ms);

/* ---- evaluator-level entry points (parity + Processor shim) ------------- */
/* Ensemble.compute_feature_vector (ensemble.py:323-351) for nocc occupancies:
 * correlations_from_occupancy / interactions_from_occupancy x size
 * (evaluator.pyx:121-209; expansion.py:165-189,391-414), Ewald feature
 * (processor/ewald.py:128-145), chemical work.  features [nocc x F]. */
int smolmc_eval_full(smolmc_handle *h, const int32_t *occ /*nocc x N*/, int nocc,
                     double *features);
/* Ensemble.compute_feature_vector_change (ensemble.py:353-376) for nstep steps of
 * up to SMOLMC_MAX_STEP_FLIPS sequential flips each on ONE occupancy:
 * delta_correlations_from_occupancies / delta_interactions_from_occupancies
 * (evaluator.pyx:211-317) x size with sequential-flip semantics
 * (expansion.py:217-229), delta_ewald_single_flip (ewald.pyx:9-59), mu table.
 * flips [nstep x SMOLMC_STEP_ROW] step records like smolmc_replay; dfeatures [nstep x F]. */
int smolmc_eval_delta(smolmc_handle *h, const int32_t *occ /*N*/, const int32_t *flips,
                      int nstep, double *dfeatures);

/* ---- device plumbing (multi-GPU / RCCL glue, optional) ------------------- */
/* Use an externally created hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) */
int smolmc_set_stream(smolmc_handle *h, void *hip_stream);
/* Copy per-walker enthalpies (float64[R]) into a DEVICE buffer (for all_gather) */
int smolmc_export_enthalpy_dev(smolmc_handle *h, double *dst_dev);
/* Permute per-walker temperatures from a DEVICE or host array after an exchange */
int smolmc_import_temperature_dev(smolmc_handle *h, const double *src_dev);
/* One exchange attempt of a replica-exchange temperature ladder decided on the device (ABI 8; no reference
 * counterpart: SURVEY 8e, BASELINE configs[4]).  n_total walkers over all ranks, this handle holds walkers
 * first .. first + R - 1 of them.  All arrays are DEVICE arrays of this handle's GPU:
 *   enthalpy_all_dev [n_total]   the all-gathered enthalpies, global walker order
 *   ladder_dev       [n_total]   temperature of every rung
 *   log_u_dev        [>= n_total / 2]  log of the acceptance uniform of the p-th pair of this attempt (the caller's
 *                                counter-based stream: identical on every rank)
 *   rung_of_dev      [n_total]   in / out: walker -> rung
 *   stats_dev        [2 (n_total - 1)] in / out, may be NULL: attempts | acceptances per neighbouring pair of rungs
 * parity 0 / 1: pairs (0,1),(2,3),... / (1,2),(3,4),...  Accepts with min(1, exp((beta_k - beta_k+1)(H_a - H_b))),
 * moves the rung assignment and sets THIS handle's temperatures to ladder[rung_of[first + r]]; asynchronous on the
 * handle's stream (the inputs must be complete when it is called).  Every rank that calls it with the same inputs
 * takes the same decisions; only temperatures move, occupancies never leave their GPU. */
int smolmc_exchange_dev(smolmc_handle *h, int n_total, int first, int parity, const double *enthalpy_all_dev,
                        const double *ladder_dev, const double *log_u_dev, int32_t *rung_of_dev, int64_t *stats_dev);

#ifdef __cplusplus
}
#endif
#endif /* SMOLMC_H */
