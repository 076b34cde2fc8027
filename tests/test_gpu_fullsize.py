"""BASELINE.json's full problem sizes, checked through size-independent properties (the oracle
only follows a handful of walkers here):

  config 2  binary FCC 16^3 (4096 sites), pair+triplet CE, 4096 walkers, canonical swap
  config 3  ternary rocksalt 12^3 (3456 sites), CE + Ewald, 2048 walkers, semigrand flip
  config 4  config-2 Hamiltonian, Wang-Landau, 1024 walkers

Properties: composition conservation, running trace == from-scratch evaluation (drift),
launch-chunking invariance and run-to-run determinism (checksums over all walkers),
delta == difference / reversibility at full size, Wang-Landau bookkeeping identities, and an
oracle spot check on a few walkers of the same launch."""

import zlib

import numpy as np
import pytest

from smol_amd import capi, ewald, synth
from smol_amd.engine import Engine

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-10, 1e-8


def checksum(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def assert_enthalpy_rel(got, want, what, record_property):
    """north_star's bar itself: |got - want| / |want| < 1e-10, PURELY relative (no absolute slack), over every walker
    whose enthalpy is not itself rounding noise (|want| > 1e-6 eV; the enthalpies here are 1e1 ... 1e3 eV).  The worst
    figure goes into the junit record of the test (as tests/test_gpu_parity.py::_max_rel does for the deltas)."""
    got, want = np.asarray(got, float), np.asarray(want, float)
    m = np.abs(want) > 1e-6
    assert m.sum() >= max(1, got.size // 2), what
    worst = float(np.max(np.abs(got[m] - want[m]) / np.abs(want[m])))
    record_property("max_rel_enthalpy_" + what, worst)
    print(f"max relative enthalpy error [{what}]: {worst:.2e} over {int(m.sum())} walkers")
    assert worst < 1e-10, (what, worst)


@pytest.fixture(scope="module")
def config2():
    model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [16, 16, 16])
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=20260928))
    R = 4096
    rng = np.random.default_rng(7)
    occ = np.zeros((R, sc.num_sites), dtype=np.int32)
    for r in range(R):
        occ[r, rng.permutation(sc.num_sites)[: sc.num_sites // 2]] = 1
    return sc, tab, occ


def test_config2_full_size_properties(config2, record_property):
    from oracle import oracle as orc

    sc, tab, occ0 = config2
    R = len(occ0)
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
    seeds = np.arange(R, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(99)
    a, b = Engine(tab, cfg), Engine(tab, cfg)
    a.set_state(occ0, seeds, 2500.0)
    b.set_state(occ0, seeds, 2500.0)
    a.run(3000)
    for chunk in (1, 999, 1500, 500):  # same 3000 steps in uneven launches
        b.run(chunk)
    sa, sb = a.get_state(), b.get_state()
    # chunking invariance + determinism: bit-identical occupancies and counters
    assert checksum(sa["occupancy"]) == checksum(sb["occupancy"])
    assert checksum(sa["n_accepted"]) == checksum(sb["n_accepted"])
    assert_enthalpy_rel(sa["enthalpy"], sb["enthalpy"], "config2_chunked_vs_one_launch", record_property)
    # canonical swaps conserve every walker's composition
    assert np.all(sa["occupancy"].sum(axis=1) == sc.num_sites // 2)
    assert np.all(sa["n_steps"] == 3000) and 0.2 < sa["n_accepted"].mean() / 3000 < 0.6
    # running trace == from-scratch evaluation for ALL walkers (drift audit)
    full = a.eval_full(sa["occupancy"])
    np.testing.assert_allclose(sa["features"], full, rtol=RTOL, atol=ATOL)
    assert_enthalpy_rel(sa["enthalpy"], full @ a.natural_parameters, "config2_running_vs_from_scratch", record_property)
    # oracle spot check: the first 6 walkers of the same run
    k = 6
    ora = orc.OracleMC(tab, capi.make_config(k, capi.KERNEL_METROPOLIS, capi.STEP_SWAP))
    ora.set_state(occ0[:k], seeds[:k], 2500.0)
    ora.run(3000)
    so = ora.get_state()
    assert np.array_equal(sa["occupancy"][:k], so["occupancy"])
    assert np.array_equal(sa["n_accepted"][:k], so["n_accepted"])
    assert_enthalpy_rel(sa["enthalpy"][:k], so["enthalpy"], "config2_vs_oracle", record_property)
    # delta == difference and reversibility at full size (tests/test_moca/test_processor.py:175-231)
    rng = np.random.default_rng(3)
    occ = sa["occupancy"][17].copy()
    for _ in range(10):
        s1, s2 = rng.choice(sc.num_sites, 2, replace=False)
        flips = [(int(s1), int(1 - occ[s1])), (int(s2), int(1 - occ[s2]))]
        new = occ.copy()
        for s, c in flips:
            new[s] = c
        d = np.ravel(a.eval_delta(occ, flips))
        f0, f1 = a.eval_full(occ[None])[0], a.eval_full(new[None])[0]
        np.testing.assert_allclose(d, f1 - f0, rtol=1e-8, atol=1e-8)
        back = np.ravel(a.eval_delta(new, [(s, int(occ[s])) for s, _ in flips][::-1]))
        np.testing.assert_allclose(d, -back, rtol=1e-12, atol=1e-9)
        occ = new


def test_config3_full_size_properties(record_property):
    from oracle import oracle as orc

    model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [12, 12, 12])
    ew = ewald.supercell_ewald(sc)
    mu = np.zeros((sc.num_sites, 3))
    mu[: sc.size] = np.random.default_rng(7).uniform(-0.5, 0.5, 3)[None, :]
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model), ewald=ew, ewald_coef=0.1, mu_table=mu)
    R = 2048
    rng = np.random.default_rng(11)
    occ0 = np.zeros((R, sc.num_sites), dtype=np.int32)
    occ0[:, : sc.size] = rng.integers(0, 3, size=(R, sc.size))
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(4242)
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_FLIP)
    a, b = Engine(tab, cfg), Engine(tab, cfg)
    for e in (a, b):
        e.set_state(occ0, seeds, 3000.0)
    a.run(1200)
    for chunk in (7, 593, 600):
        b.run(chunk)
    sa, sb = a.get_state(), b.get_state()
    assert checksum(sa["occupancy"]) == checksum(sb["occupancy"])
    assert checksum(sa["n_accepted"]) == checksum(sb["n_accepted"])
    assert np.all(sa["occupancy"][:, sc.size:] == 0)  # the anion sublattice is never touched
    # running trace (CE + Ewald + chemical work) == from-scratch evaluation, all walkers:
    # in particular the Ewald potential field has not drifted from the occupancies
    full = a.eval_full(sa["occupancy"])
    np.testing.assert_allclose(sa["features"], full, rtol=RTOL, atol=1e-7)
    assert_enthalpy_rel(sa["enthalpy"], full @ a.natural_parameters, "config3_running_vs_from_scratch", record_property)
    k = 4
    ora = orc.OracleMC(tab, capi.make_config(k, capi.KERNEL_METROPOLIS, capi.STEP_FLIP))
    ora.set_state(occ0[:k], seeds[:k], 3000.0)
    ora.run(1200)
    so = ora.get_state()
    assert np.array_equal(sa["occupancy"][:k], so["occupancy"])
    assert_enthalpy_rel(sa["enthalpy"][:k], so["enthalpy"], "config3_vs_oracle", record_property)


def test_config4_full_size_wang_landau_identities(config2, record_property):
    sc, tab, occ0 = config2
    R = 1024
    probe = Engine(tab, capi.make_config(1))
    h0 = float(probe.natural_parameters @ probe.eval_full(occ0[:1])[0])
    probe.close()
    cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, capi.STEP_SWAP, min_enthalpy=h0 - 160.37,
                           max_enthalpy=h0 + 96.11, bin_size=0.5, flatness=0.8, check_period=1000)
    eng = Engine(tab, cfg)
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(31337)
    eng.set_state(occ0[:R], seeds, 0.0)
    nsteps = 2500
    eng.run(1000)
    eng.run(1500)
    st, wl = eng.get_state(), eng.get_wl()
    assert np.all(st["occupancy"].sum(axis=1) == sc.num_sites // 2)
    lo, hi = cfg.wl_min_enthalpy, cfg.wl_max_enthalpy
    assert np.all((st["enthalpy"] >= lo) & (st["enthalpy"] < hi))  # walkers never leave the window
    # every in-window step adds one occurrence and mod_factor of entropy (update_period 1);
    # histograms were reset at most at the flatness checks
    assert np.all(wl["occurrences"].sum(axis=1) == nsteps)
    assert np.all(wl["histogram"].sum(axis=1) <= nsteps)
    # ... and the entropy added over all bins is the sum of the modification factors the steps were taken at:
    # between n m_final (every step after the last reduction) and n m_0
    assert np.all(wl["entropy"].sum(axis=1) >= nsteps * wl["mod_factor"] * (1 - 1e-12))
    assert np.all(wl["entropy"].sum(axis=1) <= nsteps * cfg.wl_mod_factor * (1 + 1e-12))
    assert np.all((wl["entropy"] > 0) == (wl["occurrences"] > 0))
    # the per-bin mean features average to the global mean weighted by occurrences; each mean
    # row reproduces an enthalpy inside its bin
    nat = eng.natural_parameters
    r = 5
    vis = np.flatnonzero(wl["occurrences"][r] > 0)
    h_bin = wl["mean_features"][r][vis] @ nat
    edges = lo + 0.5 * vis
    assert np.all(h_bin >= edges - 1e-9) and np.all(h_bin < edges + 0.5 + 1e-9)
    # running trace == from-scratch evaluation
    full4 = eng.eval_full(st["occupancy"])
    np.testing.assert_allclose(st["features"], full4, rtol=RTOL, atol=ATOL)
    assert_enthalpy_rel(st["enthalpy"], full4 @ nat, "config4_running_vs_from_scratch", record_property)
    # oracle spot check: four walkers of the SAME launches (walkers 0, 1, 511, 1023 of the 1024; a walker's chain
    # depends on its seed and start only) followed step for step on the CPU -- occupancies, counters, histograms,
    # occurrences, entropies bit-equal, per-bin mean features 1e-10
    from oracle import oracle as orc

    pick = np.array([0, 1, 511, 1023])
    cfg4 = capi.make_config(len(pick), capi.KERNEL_WANGLANDAU, capi.STEP_SWAP, min_enthalpy=h0 - 160.37,
                            max_enthalpy=h0 + 96.11, bin_size=0.5, flatness=0.8, check_period=1000)
    ora = orc.OracleMC(tab, cfg4)
    ora.set_state(occ0[pick], seeds[pick], 0.0)
    ora.run(nsteps)
    so, wo = ora.get_state(), ora.get_wl()
    assert np.array_equal(st["occupancy"][pick], so["occupancy"])
    assert np.array_equal(st["n_accepted"][pick], so["n_accepted"])
    assert_enthalpy_rel(st["enthalpy"][pick], so["enthalpy"], "config4_vs_oracle", record_property)
    assert np.array_equal(wl["histogram"][pick], wo["histogram"])
    assert np.array_equal(wl["occurrences"][pick], wo["occurrences"])
    np.testing.assert_allclose(wl["entropy"][pick], wo["entropy"], rtol=0, atol=0)
    np.testing.assert_allclose(wl["mean_features"][pick], wo["mean_features"], rtol=1e-10, atol=1e-8)
    np.testing.assert_allclose(wl["mod_factor"][pick], wo["mod_factor"])


def test_config5_full_size_properties(record_property):
    """BASELINE configs[4] as a whole: 12^3 ternary rocksalt (3456 sites) + Ewald, charge-neutral
    TableFlip (3 Mn3+ <-> Li+ + 2 Ti4+, swap_weight 0.1) on 2048 walkers with a geometric
    replica-exchange ladder 400-2000 K, one exchange attempt per sweep (3456 steps).  Checked
    through size-independent properties + an oracle spot check between exchanges
    (smol/moca/kernel/mcusher.py:553-711 for the step; the ladder is new functionality)."""
    from oracle import oracle as orc
    from smol_amd import parallel, workloads

    wl = workloads.config5()
    sc, tab, R, N = wl.sc, wl.tables, wl.n_walkers, wl.sc.num_sites
    assert (R, N, wl.mc_per_launch) == (2048, 3456, 3456)
    P = sc.size
    charge = np.array([1.0, 3.0, 4.0])

    def net_cation_charge(occ):
        return charge[occ[:, :P]].sum(axis=1)

    q0 = net_cation_charge(wl.occupancy)
    assert np.all(q0 == 2.0 * P)  # neutral against P O2- anions
    ladder = wl.extras["ladder"]
    cfg = wl.make_config()
    a, b = Engine(tab, cfg), Engine(tab, cfg)
    assert a.kernel_info().startswith("lean")
    rex_a = parallel.ReplicaExchange(ladder, R, seed=11)
    rex_b = parallel.ReplicaExchange(ladder, R, seed=11)
    for e in (a, b):
        e.set_state(wl.occupancy, wl.seeds, ladder)
    # --- between two exchanges the walkers are plain TableFlip chains: oracle spot check --------
    k = 4
    pick = np.array([0, 700, 1400, R - 1])  # cold, two middle rungs, hot
    ora = orc.OracleMC(tab, capi.make_config(k, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP))
    ora.set_state(wl.occupancy[pick], wl.seeds[pick], ladder[pick])
    a.run(600)
    ora.run(600)
    sa, so = a.get_state(), ora.get_state()
    assert np.array_equal(sa["occupancy"][pick], so["occupancy"])
    assert np.array_equal(sa["n_accepted"][pick], so["n_accepted"])
    assert_enthalpy_rel(sa["enthalpy"][pick], so["enthalpy"], "config5_vs_oracle", record_property)
    # --- the ladder: 3 sweeps + exchanges in one go (a) vs uneven launch chunks (b) -------------
    a.run(wl.mc_per_launch - 600)
    rex_a.decide(a.get_enthalpy())
    a.set_temperature(rex_a.local_temperatures())
    parallel.run_replica_exchange(a, rex_a, 2, wl.mc_per_launch)
    for sweep in range(3):
        for chunk in ((600, 2856) if sweep == 0 else (1, 1455, 2000)):
            b.run(chunk)
        rex_b.decide(b.get_enthalpy())
        b.set_temperature(rex_b.local_temperatures())
    sa, sb = a.get_state(), b.get_state()
    assert checksum(sa["occupancy"]) == checksum(sb["occupancy"])  # chunking invariance, determinism
    assert checksum(sa["n_accepted"]) == checksum(sb["n_accepted"])
    assert np.array_equal(rex_a.rung_of, rex_b.rung_of)
    # rungs stay a permutation of the walkers and exchanges do happen
    assert sorted(rex_a.rung_of) == list(range(R))
    assert rex_a.attempted.sum() == 3 * (R // 2) - 1 and rex_a.accepted.sum() > 0
    np.testing.assert_allclose(np.sort(rex_a.temperatures), ladder)
    # charge neutrality of EVERY walker, anion sublattice untouched
    assert np.all(net_cation_charge(sa["occupancy"]) == 2.0 * P)
    assert np.all(sa["occupancy"][:, P:] == 0)
    assert np.all(sa["n_steps"] == 3 * wl.mc_per_launch)
    # compositions move along the flip direction only: (dLi, dMn, dTi) = k (1, -3, 2)
    n1 = np.stack([(sa["occupancy"][:, :P] == c).sum(axis=1) for c in range(3)], axis=1)
    n0 = np.stack([(wl.occupancy[:, :P] == c).sum(axis=1) for c in range(3)], axis=1)
    kdir = (n1 - n0)[:, 0]
    assert np.array_equal(n1 - n0, kdir[:, None] * np.array([[1, -3, 2]]))
    assert len(np.unique(kdir)) > 3  # the table steps are taken
    # running trace (CE + Ewald field + chemical work) == from-scratch evaluation, all walkers
    full = a.eval_full(sa["occupancy"])
    np.testing.assert_allclose(sa["features"], full, rtol=RTOL, atol=1e-6)
    assert_enthalpy_rel(sa["enthalpy"], full @ a.natural_parameters, "config5_running_vs_from_scratch", record_property)
    # neighbouring rungs of a 2048-step geometric ladder overlap almost completely
    assert 0.8 < rex_a.acceptance.mean() <= 1.0


def test_config5_model_under_wang_landau_table_flip(record_property):
    """The model of BASELINE configs[4] (12^3 ternary rocksalt, 3456 sites, Ewald, charge-neutral TableFlip) under the
    Wang-Landau kernel -- the usher x kernel pair the reference composes (kernel/base.py:192-239, wanglandau.py:186-266)
    and round 6 moved onto mc_table_kernel<..., WLT>: 256 walkers at full size.  An oracle spot check of four walkers
    (occupancies, accept flags, entropies, histograms bit-exact; enthalpies to 1e-10 relative), then the size-independent
    identities of every walker: histogram == occurrences == steps counted while no flatness check fired, entropy == m x
    occurrences, charge neutrality, the composition on the flip direction, the running trace against a from-scratch
    evaluation, chunking invariance."""
    from oracle import oracle as orc
    from smol_amd import workloads

    wl = workloads.config5()
    sc, tab, N = wl.sc, wl.tables, wl.sc.num_sites
    R, P = 256, sc.size
    occ, seeds = wl.occupancy[:R], wl.seeds[:R]
    probe = Engine(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP))
    probe.set_state(occ, seeds, 2000.0)
    h0 = probe.get_state()["enthalpy"]
    probe.close()
    lo, hi = h0.min() - 40.0, h0.max() + 40.0
    cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, capi.STEP_TABLE_FLIP, min_enthalpy=lo, max_enthalpy=hi,
                           bin_size=(hi - lo) / 200.0, check_period=10 ** 9)
    a, b = Engine(tab, cfg), Engine(tab, cfg)
    assert a.kernel_info().startswith("lean ") and "wl=table" in a.kernel_info(), a.kernel_info()
    pick = np.array([0, 85, 170, R - 1])
    ora = orc.OracleMC(tab, capi.make_config(len(pick), capi.KERNEL_WANGLANDAU, capi.STEP_TABLE_FLIP, min_enthalpy=lo,
                                             max_enthalpy=hi, bin_size=(hi - lo) / 200.0, check_period=10 ** 9))
    for e in (a, b):
        e.set_state(occ, seeds, 0.0)
    ora.set_state(occ[pick], seeds[pick], 0.0)
    a.run(500)
    ora.run(500)
    sa, so = a.get_state(), ora.get_state()
    wa, wo = a.get_wl(), ora.get_wl()
    assert np.array_equal(sa["occupancy"][pick], so["occupancy"])
    assert np.array_equal(sa["n_accepted"][pick], so["n_accepted"])
    assert_enthalpy_rel(sa["enthalpy"][pick], so["enthalpy"], "config5_wl_table_vs_oracle", record_property)
    assert np.array_equal(wa["histogram"][pick], wo["histogram"])
    np.testing.assert_allclose(wa["entropy"][pick], wo["entropy"], rtol=0, atol=0)
    np.testing.assert_allclose(wa["mean_features"][pick], wo["mean_features"], rtol=1e-10, atol=1e-6)
    a.run(2956)  # one sweep in all
    for chunk in (1, 1455, 2000):
        b.run(chunk)
    sa, sb = a.get_state(), b.get_state()
    wa, wb = a.get_wl(), b.get_wl()
    assert checksum(sa["occupancy"]) == checksum(sb["occupancy"])
    assert checksum(wa["histogram"]) == checksum(wb["histogram"])
    np.testing.assert_allclose(wa["entropy"], wb["entropy"], rtol=0, atol=0)
    assert np.all(sa["n_steps"] == 3456)
    assert np.all(wa["histogram"].sum(axis=1) == 3456) and np.array_equal(wa["histogram"], wa["occurrences"])
    np.testing.assert_allclose(wa["entropy"], wa["mod_factor"][:, None] * wa["occurrences"], rtol=1e-12)
    charge = np.array([1.0, 3.0, 4.0])
    assert np.all(charge[sa["occupancy"][:, :P]].sum(axis=1) == 2.0 * P)
    assert np.all(sa["occupancy"][:, P:] == 0)
    n1 = np.stack([(sa["occupancy"][:, :P] == c).sum(axis=1) for c in range(3)], axis=1)
    n0 = np.stack([(occ[:, :P] == c).sum(axis=1) for c in range(3)], axis=1)
    kdir = (n1 - n0)[:, 0]
    assert np.array_equal(n1 - n0, kdir[:, None] * np.array([[1, -3, 2]]))
    acc = sa["n_accepted"].sum() / sa["n_steps"].sum()
    assert 0.05 < acc < 0.99, acc
    full = a.eval_full(sa["occupancy"])
    np.testing.assert_allclose(sa["features"], full, rtol=RTOL, atol=1e-6)
    assert_enthalpy_rel(sa["enthalpy"], full @ a.natural_parameters, "config5_wl_table_running_vs_from_scratch", record_property)
    assert np.all((sa["enthalpy"] >= lo) & (sa["enthalpy"] < hi))
    a.close()
    b.close()
