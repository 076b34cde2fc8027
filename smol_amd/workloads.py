"""The five BASELINE.json configurations (and three shapes outside it) as builders.

One place defines the synthetic workloads so that ``bench.py`` (headline + ``other_configs``),
``tools/bench_configs.py`` and the full-size tests measure / check the same thing.  Host setup
only: tables come from smol_amd.synth / smol_amd.ewald, nothing here touches the GPU.
"""

from __future__ import annotations

import numpy as np

from . import capi, ewald, synth


class Workload:
    """tables + engine config + initial state of one configuration (this rank's walkers)."""

    def __init__(self, key, name, sc, tables, config_kwargs, occupancy, seeds, temperature,
                 flips_per_step, mc_per_launch, extras=None):
        self.key, self.name, self.sc, self.tables = key, name, sc, tables
        self.config_kwargs = config_kwargs
        self.occupancy, self.seeds, self.temperature = occupancy, seeds, temperature
        self.flips_per_step, self.mc_per_launch = flips_per_step, mc_per_launch
        self.extras = extras or {}

    @property
    def n_walkers(self):
        return len(self.occupancy)

    def make_config(self, device=0):
        kw = dict(self.config_kwargs)
        return capi.make_config(self.n_walkers, kw.pop("kernel"), kw.pop("step"), device, **kw)


def balanced_binary(sc, first, count, seed=1000):
    """50/50 occupancies; walker g (global index) always gets the same start, so the job is
    independent of how the walkers are sharded over ranks."""
    occ = np.zeros((count, sc.num_sites), dtype=np.int32)
    for i in range(count):
        perm = np.random.default_rng(seed + first + i).permutation(sc.num_sites)
        occ[i, perm[: sc.num_sites // 2]] = 1
    return occ


def random_codes(sc, first, count, seed):
    nsp = np.array([sc.model.prim.nspecies[b] for b in sc.site_b])
    out = np.zeros((count, sc.num_sites), np.int32)
    for i in range(count):
        out[i] = (np.random.default_rng(seed + first + i).random(sc.num_sites) * nsp).astype(np.int32)
    return out


def _seeds(first, count, base):
    return np.arange(first, first + count, dtype=np.uint64) + np.uint64(base)


def neutral_rocksalt_occupancy(sc, first, count, seed=5):
    """Charge-neutral Li+/Mn3+/Ti4+ start on the cation sublattice of a rocksalt supercell
    (2 n_Mn + 3 n_Ti = P with codes Li=0, Mn=1, Ti=2)."""
    P = sc.size
    n_ti = 2 * (P // 12)
    n_mn = (P - 3 * n_ti) // 2
    occ = np.zeros((count, sc.num_sites), np.int32)
    for i in range(count):
        perm = np.random.default_rng(seed * 100003 + first + i).permutation(P)
        occ[i, perm[:n_mn]] = 1
        occ[i, perm[n_mn:n_mn + n_ti]] = 2
    return occ


HEADLINE_T = 2500.0  # acceptance ~0.38 on the config-2 Hamiltonian (tuned once and frozen)
# configs 3 and 5: temperature / chemical-potential scale (see tools/equil_sweep.py).
# Config 3 (round 6): tuned once and frozen like the headline's T.  SURVEY 8d fixes the lattice, epsilon = 10 and the seeded
# mu draw, not the temperature.  Unconstrained semigrand flips on an Ewald energy without the charged-cell term run to one
# pure composition below ~2e4 K (acceptance 1e-6 at rounds 2-5's 3000 K: their sweeps stopped at 12000 K and concluded that
# no mixed steady state exists); at 40000 K the chain is stationary -- acceptance 0.380 after 4e5 AND after 2e6 steps per
# walker, cations Li / Mn / Ti = 0.045 / 0.29 / 0.665 (profiles/r06_equil_sweep.jsonl).  CONFIG3_T_REJECT keeps the old
# point as `config3_reject_path`.
CONFIG3_T, CONFIG3_MU = 40000.0, 0.5
CONFIG3_T_REJECT = 3000.0
# (round 3: config 5's ladder moved from SURVEY's 400-2000 K, where the equilibrated walkers accept
# 0.1 % of their steps, to 2500-12500 K: steady-state acceptance 0.17, profiles/r03_equil_sweep.jsonl)
CONFIG5_T, CONFIG5_MU = (2500.0, 12500.0), 0.5
CONFIG9_T, CONFIG9_PENALTY = 5000.0, 0.05  # steady-state acceptance 0.11 at 4000 K, 0.24 at 6000 K (profiles/r03_equil_sweep.jsonl)


def config2(first=0, count=4096, dim=16, feature_mode=capi.FEATURES_INTERACTIONS, mc=10000):
    """BASELINE configs[1] (headline): binary FCC dim^3 primitive supercell, point + 4 pair + 2
    triplet orbits (115 clusters per site), canonical swap Metropolis, ECI U(-0.02, 0.02) eV."""
    model = synth.build_cluster_model(synth.fcc_prim(a=4.09), {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [dim] * 3)
    coefs = synth.random_coefs(model, seed=20260928, scale=0.02)
    tab = capi.TableSet.from_synth(sc, coefs, feature_mode=feature_mode)
    trace = "cluster-interaction" if feature_mode == capi.FEATURES_INTERACTIONS else "correlation"
    return Workload(
        2, f"binary FCC {dim}x{dim}x{dim} ({sc.num_sites} sites), point+4 pair+2 triplet CE, canonical "
           f"swap Metropolis, {trace} trace, T=2500K",
        sc, tab, dict(kernel=capi.KERNEL_METROPOLIS, step=capi.STEP_SWAP),
        balanced_binary(sc, first, count), _seeds(first, count, 12345), HEADLINE_T, 2, mc,
        extras=dict(model=model, coefs=coefs))


def config1(first=0, count=4096, dim=4, mc=5000):
    """BASELINE configs[0]: binary FCC conventional 4x4x4 (256 sites), pair-only CE."""
    model = synth.build_cluster_model(synth.fcc_conventional_prim(), {2: 6.0})
    sc = synth.build_supercell(model, [dim] * 3)
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model))
    return Workload(
        1, f"config1: binary FCC conventional {dim}^3 ({sc.num_sites} sites), pairs, canonical swap",
        sc, tab, dict(kernel=capi.KERNEL_METROPOLIS, step=capi.STEP_SWAP),
        balanced_binary(sc, first, count, seed=1), _seeds(first, count, 777), 2500.0, 2, mc)


_ROCKSALT_LAST = {}  # dim -> (model, supercell, Ewald tables) of the default lattice: the last one built


def _rocksalt(dim, cutoffs=None, prim=None):
    """Cluster model, supercell and Ewald tables of the ternary rocksalt workloads.  The default lattice at one
    size is shared by configs 3 / 5 / 6 / 9 / 13 (read-only: TableSet keeps references and replaces, never
    writes, what it relabels): one entry is kept, so a bench run builds the 382 MB Ewald matrix once instead of
    once per configuration."""
    if cutoffs is None and prim is None and dim in _ROCKSALT_LAST:
        return _ROCKSALT_LAST[dim]
    model = synth.build_cluster_model(prim or synth.rocksalt_prim(), cutoffs or {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [dim] * 3)
    out = (model, sc, ewald.supercell_ewald(sc))
    if cutoffs is None and prim is None:
        _ROCKSALT_LAST.clear()
        _ROCKSALT_LAST[dim] = out
    return out


def _mu_rows(sc, scale, values=None):
    """Chemical potentials of the three cation species on every cation site: SURVEY 8d's U(-scale, scale) draw
    with seed 7, or the three ``values`` given (tools/equil_sweep.py's mu axis)."""
    mu = np.zeros((sc.num_sites, 3))
    mu[: sc.size] = (np.random.default_rng(7).uniform(-scale, scale, 3) if values is None
                     else np.asarray(values, dtype=float))[None, :]
    return mu


def config3(first=0, count=2048, dim=12, mc=2000, temperature=None, mu_scale=None, feature_mode=capi.FEATURES_INTERACTIONS,
            ewald_coef=0.1, mu_values=None):
    """BASELINE configs[2]: ternary rocksalt dim^3, triplet CE + Ewald, semigrand flip.  ``ewald_coef`` (1 / epsilon,
    SURVEY 8d: epsilon = 10) and ``mu_values`` are the axes of tools/equil_sweep.py."""
    model, sc, ew = _rocksalt(dim)
    temperature = CONFIG3_T if temperature is None else temperature
    mu = _mu_rows(sc, CONFIG3_MU if mu_scale is None else mu_scale, mu_values)
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model), ewald=ew, ewald_coef=ewald_coef, mu_table=mu,
                                   feature_mode=feature_mode)
    return Workload(
        3, f"config3: ternary rocksalt {dim}^3 ({sc.num_sites} sites), triplet CE + Ewald, semigrand flip, T={temperature:g}K",
        sc, tab, dict(kernel=capi.KERNEL_METROPOLIS, step=capi.STEP_FLIP),
        random_codes(sc, first, count, 3), _seeds(first, count, 777), temperature, 1, mc)


def config4(first=0, count=1024, dim=16, mc=5000, h0=None):
    """BASELINE configs[3]: config-2 Hamiltonian, Wang-Landau, 512 bins of 0.5 eV.  ``h0`` (the
    enthalpy of a 50/50 random start, which centres the window) is evaluated by the caller on the
    engine; pass it to get the final configuration."""
    model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [dim] * 3)
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=20260928))
    occ = balanced_binary(sc, first, count, seed=4)
    kw = dict(kernel=capi.KERNEL_WANGLANDAU, step=capi.STEP_SWAP)
    if h0 is not None:
        # edges incommensurate with the starting enthalpy (NOTES.md: bin-edge ties)
        kw.update(min_enthalpy=h0 - 160.37, max_enthalpy=h0 + 95.63, bin_size=0.5, flatness=0.8,
                  check_period=1000)
    return Workload(
        4, f"config4: binary FCC {dim}^3 pair+triplet, Wang-Landau swap, 512 bins",
        sc, tab, kw, occ, _seeds(first, count, 777), 0.0, 2, mc)


def config5(first=0, count=2048, dim=12, mc=None, total=None, t_lo=None, t_hi=None, mu_scale=None):
    """BASELINE configs[4]: config-3 lattice, charge-neutral TableFlip (3 Mn3+ <-> Li+ + 2 Ti4+)
    with a geometric replica-exchange ladder 400-2000 K over ``total`` walkers (this rank holds
    walkers first .. first+count)."""
    from . import parallel

    model, sc, ew = _rocksalt(dim)
    mu = _mu_rows(sc, CONFIG5_MU if mu_scale is None else mu_scale)
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model), ewald=ew, ewald_coef=0.1,
                                   mu_table=mu, flip_table=[[1, -3, 2]], swap_weight=0.1)
    total = total or count
    t_lo, t_hi = (CONFIG5_T[0] if t_lo is None else t_lo), (CONFIG5_T[1] if t_hi is None else t_hi)
    ladder = parallel.geometric_ladder(t_lo, t_hi, total)
    return Workload(
        5, f"config5: ternary rocksalt {dim}^3 + Ewald, charge-neutral TableFlip, replica-exchange "
           f"ladder {t_lo:g}-{t_hi:g} K over {total} walkers",
        sc, tab, dict(kernel=capi.KERNEL_METROPOLIS, step=capi.STEP_TABLE_FLIP),
        neutral_rocksalt_occupancy(sc, first, count), _seeds(first, count, 777),
        ladder[first:first + count].copy(), 1, mc or sc.num_sites, extras=dict(ladder=ladder))


def config6(first=0, count=2048, dim=12, mc=2000):
    """(not in BASELINE.json) config-3 lattice, canonical swap with the Ewald term."""
    model, sc, ew = _rocksalt(dim)
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model), ewald=ew, ewald_coef=0.1)
    return Workload(
        6, f"config6: ternary rocksalt {dim}^3 ({sc.num_sites} sites), triplet CE + Ewald, canonical swap",
        sc, tab, dict(kernel=capi.KERNEL_METROPOLIS, step=capi.STEP_SWAP),
        random_codes(sc, first, count, 3), _seeds(first, count, 777), 3000.0, 2, mc)


def config7(first=0, count=2048, dim=12, mc=1000, with_ewald=True):
    """(not in BASELINE.json) two ACTIVE sublattices: Li+/Mn3+/Ti4+ cations and O2-/F- anions."""
    prim = synth.rocksalt_prim(anion_charges=(-2.0, -1.0))
    model = synth.build_cluster_model(prim, {2: 6.0, 3: 4.5})
    sc = synth.build_supercell(model, [dim] * 3)
    ew = ewald.supercell_ewald(sc) if with_ewald else None
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model), ewald=ew, ewald_coef=0.1)
    return Workload(
        7, f"config7: rocksalt {dim}^3 ({sc.num_sites} sites), ternary cations + binary anions, CE"
           f"{' + Ewald' if with_ewald else ''}, canonical swap",
        sc, tab, dict(kernel=capi.KERNEL_METROPOLIS, step=capi.STEP_SWAP),
        random_codes(sc, first, count, 3), _seeds(first, count, 777), 3000.0, 2, mc)


def config8(first=0, count=4096, dim=16, mc=2000):
    """(not in BASELINE.json) 451 clusters per site on the config-2 lattice."""
    model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.5, 3: 5.2})
    sc = synth.build_supercell(model, [dim] * 3)
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=20260928))
    return Workload(
        8, f"config8: binary FCC {dim}^3 ({sc.num_sites} sites), pairs <= 6.5 A + triplets <= 5.2 A "
           "(451 clusters/site), canonical swap",
        sc, tab, dict(kernel=capi.KERNEL_METROPOLIS, step=capi.STEP_SWAP),
        balanced_binary(sc, first, count, seed=1), _seeds(first, count, 777), 2500.0, 2, mc)


def config9(first=0, count=2048, dim=12, mc=2000, temperature=None, penalty=None, mu_scale=None):
    """(not in BASELINE.json) config 3 made well-posed: the same lattice, CE, Ewald term and
    semigrand single flips, plus a SquareChargeBias (smol/moca/kernel/bias.py:229-287; the
    reference's recipe for charge-neutral semigrand sampling with Flip steps,
    docs/src/notebooks/running-charge-balanced-gcmc.ipynb) and a charge-neutral start.  Config 3
    as specified has no mixed steady state: its Ewald energy without the charged-cell term is
    concave in the net charge Q (-1.3e-3 eV Q^2 at coefficient 0.1), unconstrained flips run away
    to a pure composition and the steady-state acceptance is ~1e-6 (profiles/r03_equil_sweep.jsonl)."""
    model, sc, ew = _rocksalt(dim)
    temperature = CONFIG9_T if temperature is None else temperature
    mu = _mu_rows(sc, CONFIG3_MU if mu_scale is None else mu_scale)
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model), ewald=ew, ewald_coef=0.1, mu_table=mu)
    q = np.zeros((sc.num_sites, 3))
    for s in range(sc.num_sites):
        ch = model.prim.charges[sc.site_b[s]]
        q[s, :len(ch)] = [c or 0.0 for c in ch]
    tab.set_bias(capi.BIAS_SQUARE_CHARGE, q, CONFIG9_PENALTY if penalty is None else penalty)
    return Workload(
        9, f"config9: config 3 + SquareChargeBias (charge-neutral semigrand flips), ternary rocksalt {dim}^3 "
           f"({sc.num_sites} sites), triplet CE + Ewald",
        sc, tab, dict(kernel=capi.KERNEL_METROPOLIS, step=capi.STEP_FLIP),
        neutral_rocksalt_occupancy(sc, first, count), _seeds(first, count, 777), temperature, 1, mc)


def config10(first=0, count=1024, dim=12, mc=2000, h0=None):
    """(not in BASELINE.json) semigrand Wang-Landau with the Ewald term: the config-3 model (ternary
    rocksalt dim^3, triplet CE + Ewald + mu, single flips) under the Wang-Landau kernel, window
    centred on the starting enthalpy ``h0`` (evaluated by the caller on the engine, as for config 4).
    The model class mc_wl_kernel took over from mc_kernel in round 4."""
    model, sc, ew = _rocksalt(dim)
    mu = _mu_rows(sc, CONFIG3_MU)
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model), ewald=ew, ewald_coef=0.1, mu_table=mu)
    kw = dict(kernel=capi.KERNEL_WANGLANDAU, step=capi.STEP_FLIP)
    if h0 is not None:
        kw.update(min_enthalpy=h0 - 400.37, max_enthalpy=h0 + 239.63, bin_size=1.25, flatness=0.8, check_period=1000)
    return Workload(
        10, f"config10: config-3 model (ternary rocksalt {dim}^3, triplet CE + Ewald + mu), Wang-Landau flips, 512 bins",
        sc, tab, kw, neutral_rocksalt_occupancy(sc, first, count), _seeds(first, count, 777), 0.0, 1, mc)


_LNO_TABLES = {}


def config11(first=0, count=1024, dim=8, mc=2000, h0=None, step=capi.STEP_SWAP, update_period=1):
    """(not in BASELINE.json) the model the reference ships -- LiNiO2 with Li+/vacancy and Ni3+/Ni4+ disorder and an
    Ewald term (docs/src/notebooks/data/basic_ce_ewald.mson, slimmed copy under tests/golden) -- in a dim^3 cell
    under Wang-Landau: two active sublattices, the class mc_lean_multi_kernel<..., WLK> took over from mc_kernel in
    round 5.  Window of 512 bins of 0.5 eV around the starting enthalpy ``h0`` (evaluated by the caller, as for config 4)."""
    import os

    from . import mson

    if dim not in _LNO_TABLES:  # (10 s of table generation: bench.py builds the workload at three walker counts)
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lno_ce_ewald.mson.json.gz")
        _LNO_TABLES[dim] = mson.load_mson(path).tables(np.diag([dim] * 3))
    tab = _LNO_TABLES[dim]
    cell = tab.supercell
    P = cell.size
    occ = np.ones((count, cell.num_sites), dtype=np.int32)
    occ[:, 2 * P:] = 0
    for i in range(count):  # half the Li sites vacant, as many Ni4+: charge neutral
        rng = np.random.default_rng(3 * 100003 + first + i)
        occ[i, rng.permutation(P)[:P // 2]] = 0
        occ[i, P + rng.permutation(P)[:P // 2]] = 0
    kw = dict(kernel=capi.KERNEL_WANGLANDAU, step=step)
    if h0 is not None:
        kw.update(min_enthalpy=h0 - 160.37, max_enthalpy=h0 + 95.63, bin_size=0.5, flatness=0.8, check_period=1000,
                  update_period=update_period)
    return Workload(
        11, f"config11: LiNiO2 {dim}^3 ({cell.num_sites} sites, the reference's basic_ce_ewald.mson), CE + Ewald, "
            "Wang-Landau swaps on two active sublattices, 512 bins",
        cell, tab, kw, occ, _seeds(first, count, 777), 0.0, 2 if step == capi.STEP_SWAP else 1, mc)


def config12(**kw):
    """config 2 with the correlation-function trace (ClusterExpansionProcessor, evaluator.pyx:211-265; K = 1 per orbit)."""
    w = config2(feature_mode=capi.FEATURES_CORRELATIONS, **kw)
    w.key = 12
    return w


def config13(**kw):
    """config 3 with the correlation-function trace (K = 3 / 4 / 6 functions per orbit: the KF kernels)."""
    w = config3(feature_mode=capi.FEATURES_CORRELATIONS, **kw)
    w.key, w.name = 13, w.name.replace("config3:", "config13 (config 3, correlation trace):")
    return w


BUILDERS = {1: config1, 2: config2, 3: config3, 4: config4, 5: config5, 6: config6, 7: config7,
            8: config8, 9: config9, 10: config10, 11: config11, 12: config12, 13: config13}
