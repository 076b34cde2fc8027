"""The step counter is the position in a walker's Philox stream (counter words step & 0xffffffff,
step >> 32).  A production walker passes 2^32 steps after about half an hour, so every kernel
family is run ACROSS that boundary (and across 2^33, and from a counter far beyond) against the
oracle, whose counter arithmetic is plain 64-bit C: the 16- / 64-step random batches, the table
kernels' proposal batches and the Wang-Landau check period all derive their phase from the counter."""

import numpy as np
import pytest

from smol_amd import capi, ewald, moca, synth
from smol_amd.engine import Engine

pytestmark = pytest.mark.gpu

STARTS = [2**32 - 37, 2**33 - 5, 3 * 2**40 + 12345]


def _run_across(tab, cfg, occ, seeds, temps, start, expect_kernel, chunks=(1, 30, 64, 7, 200)):
    from oracle import oracle as orc

    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    assert eng.kernel_info().startswith(expect_kernel), eng.kernel_info()
    R = len(occ)
    n0 = np.full(R, start, dtype=np.uint64) + np.arange(R, dtype=np.uint64) * np.uint64(3)  # walkers out of phase
    for e in (eng, ora):
        e.set_state(occ, seeds, temps)
        e.set_counters(n0, np.zeros(R, dtype=np.uint64))
    done = 0
    for chunk in chunks:
        eng.run(chunk)
        ora.run(chunk)
        done += chunk
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["n_steps"], n0 + np.uint64(done)) and np.array_equal(b["n_steps"], a["n_steps"])
        assert np.array_equal(a["occupancy"], b["occupancy"]), f"start {start}: occupancies differ after {done} steps"
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-8)
    # device-side sampling across the boundary as well
    ring = eng.run_sampled(3, 23, occupancy=True)
    for i in range(3):
        ora.run(23)
        assert np.array_equal(ring["occupancy"][i], ora.get_state()["occupancy"])
    assert 0 < a["n_accepted"].sum()
    if cfg.kernel_type == capi.KERNEL_WANGLANDAU:
        wa, wb = eng.get_wl(), ora.get_wl()
        assert np.array_equal(wa["histogram"], wb["histogram"]) and np.array_equal(wa["occurrences"], wb["occurrences"])
        np.testing.assert_allclose(wa["entropy"], wb["entropy"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(wa["mod_factor"], wb["mod_factor"], rtol=0, atol=0)
    eng.close()


@pytest.fixture(scope="module")
def fcc():
    model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [5, 5, 5])
    return sc, capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=3))


@pytest.fixture(scope="module")
def rocksalt():
    model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 6.0, 3: 4.5})
    sc = synth.build_supercell(model, [3, 3, 3])
    return model, sc


def _fcc_occ(sc, R, seed):
    return (np.random.default_rng(seed).random((R, sc.num_sites)) < 0.5).astype(np.int32)


@pytest.mark.parametrize("start", STARTS)
@pytest.mark.parametrize("step", [capi.STEP_SWAP, capi.STEP_FLIP], ids=["swap", "flip"])
@pytest.mark.parametrize("force", [None, "SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL"], ids=["lean", "general", "universal"])
def test_metropolis_across_the_counter_boundary(fcc, start, step, force, monkeypatch):
    for name in ("SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL"):
        monkeypatch.delenv(name, raising=False)
    if force:
        monkeypatch.setenv(force, "1")
    sc, tab = fcc
    R = 5
    _run_across(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, step), _fcc_occ(sc, R, 1),
                np.arange(11, 11 + R, dtype=np.uint64), np.linspace(800.0, 3000.0, R), start,
                "lean" if force is None else force.split("_")[-1].lower())


@pytest.mark.parametrize("start", STARTS)
@pytest.mark.parametrize("force", [None, "SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL"], ids=["lean", "general", "universal"])
def test_wang_landau_across_the_counter_boundary(fcc, start, force, monkeypatch):
    """The flatness check fires where the COUNTER reaches a multiple of the check period: periods that
    do not divide 2^32 would show a counter truncated to 32 bits."""
    from oracle import oracle as orc

    for name in ("SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL"):
        monkeypatch.delenv(name, raising=False)
    if force:
        monkeypatch.setenv(force, "1")
    sc, tab = fcc
    R = 3
    occ = _fcc_occ(sc, R, 2)
    probe = orc.OracleEvaluator(tab)
    h = np.array([probe.natural_parameters() @ probe.feature_vector(o) for o in occ])
    cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, capi.STEP_SWAP, min_enthalpy=float(h.min()) - 4.0371,
                           max_enthalpy=float(h.max()) + 4.0113, bin_size=0.25, check_period=7, flatness=0.1)
    _run_across(tab, cfg, occ, np.arange(21, 21 + R, dtype=np.uint64), 0.0, start,
                "lean" if force is None else force.split("_")[-1].lower())


@pytest.mark.parametrize("start", STARTS[:2])
@pytest.mark.parametrize("two_sublattices", [False, True], ids=["cations", "cations+anions"])
def test_table_flip_and_ewald_across_the_counter_boundary(start, two_sublattices):
    """TableFlip proposal batches (64 steps, lane <-> step) and the lean-multi kernels with the Ewald field."""
    prim = synth.rocksalt_prim(anion_charges=(-2.0, -1.0)) if two_sublattices else synth.rocksalt_prim()
    model = synth.build_cluster_model(prim, {2: 6.0, 3: 4.5})
    sc = synth.build_supercell(model, [3, 3, 4])
    ens = moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model, seed=5), ewald_coefficient=0.15)
    table = ens.composition_space().flip_table
    tab = ens.make_tables(flip_table=table, swap_weight=0.2)
    R, P = 4, sc.size
    rng = np.random.default_rng(9)
    occ = np.zeros((R, sc.num_sites), dtype=np.int32)
    for r in range(R):  # charge neutral: n_Li + 3 n_Mn + 4 n_Ti = 2 P (anions all O2-)
        n_ti = 2 * int(rng.integers(1, P // 8))  # (even, so that P - 3 n_ti is)
        n_mn = (P - 3 * n_ti) // 2
        perm = rng.permutation(P)
        occ[r, perm[:n_mn]] = 1
        occ[r, perm[n_mn:n_mn + n_ti]] = 2
    _run_across(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP), occ,
                np.arange(31, 31 + R, dtype=np.uint64), np.linspace(3000.0, 9000.0, R), start,
                "lean", chunks=(1, 30, 64, 70, 130))


@pytest.mark.parametrize("kind", ["swap", "flip", "wang-landau", "table-flip", "universal"])
def test_many_walkers_of_a_small_cell(kind, monkeypatch):
    """10 007 walkers (a prime: the last workgroup is partial, the grid is thousands of workgroups) of
    a 64-site cell, every one compared with the oracle: walker -> launch slot -> wave mapping, per-walker
    seeds / temperatures / counters at scale, and the exchange-ladder launch order of the table kernel
    (walkers at different temperatures are dealt to the slots hottest-with-coldest)."""
    from oracle import oracle as orc

    for name in ("SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL"):
        monkeypatch.delenv(name, raising=False)
    R = 10007
    rng = np.random.default_rng(77)
    if kind == "table-flip":
        model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 4.5})
        sc = synth.build_supercell(model, [2, 2, 4])
        ens = moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model, seed=6), ewald_coefficient=0.1)
        tab = ens.make_tables(flip_table=[[1, -3, 2, 0]], swap_weight=0.3)
        P = sc.size
        occ = np.zeros((R, sc.num_sites), dtype=np.int32)
        for r in range(R):
            n_ti = 2 * int(rng.integers(1, 3))
            perm = rng.permutation(P)
            n_mn = (P - 3 * n_ti) // 2
            occ[r, perm[:n_mn]] = 1
            occ[r, perm[n_mn:n_mn + n_ti]] = 2
        cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP)
    else:
        if kind == "universal":
            monkeypatch.setenv("SMOLMC_FORCE_UNIVERSAL", "1")
        model = synth.build_cluster_model(synth.fcc_prim(), {2: 4.5, 3: 3.0})
        sc = synth.build_supercell(model, [4, 4, 4])
        mu = None
        if kind == "flip":
            mu = np.tile(np.array([[0.05, -0.05]]), (sc.num_sites, 1))
        tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=6), mu_table=mu)
        occ = (rng.random((R, sc.num_sites)) < 0.5).astype(np.int32)
        if kind == "wang-landau":
            probe = orc.OracleEvaluator(tab)
            h = np.array([probe.natural_parameters() @ probe.feature_vector(o) for o in occ[:200]])
            # (a window wide enough for every start: the enthalpy of 64 sites is bounded)
            cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, capi.STEP_SWAP, min_enthalpy=float(h.min()) - 40.0371,
                                   max_enthalpy=float(h.max()) + 40.0113, bin_size=0.5, check_period=13, flatness=0.1)
        else:
            cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_FLIP if kind == "flip" else capi.STEP_SWAP)
    seeds = rng.integers(1, 2**63, size=R).astype(np.uint64)
    temps = rng.uniform(500.0, 9000.0, size=R)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    for e in (eng, ora):
        e.set_state(occ, seeds, temps)
    for chunk in (1, 16, 40):
        eng.run(chunk)
        ora.run(chunk)
        a, b = eng.get_state(), ora.get_state()
        bad = np.flatnonzero((a["occupancy"] != b["occupancy"]).any(axis=1))
        assert len(bad) == 0, f"{len(bad)} walkers differ, first {bad[:5]} ({eng.kernel_info()})"
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-8)
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()
    eng.close()


@pytest.mark.parametrize("nspecies", [9, 12])
@pytest.mark.parametrize("mode", ["int", "corr"])
@pytest.mark.parametrize("step", [capi.STEP_SWAP, capi.STEP_FLIP], ids=["swap", "flip"])
def test_many_species_per_site(nspecies, mode, step):
    """Site spaces of 9 and 12 species (the lean families stop at 8 codes per sublattice): triplet tensors
    of 12^3 entries, 364 correlation functions -- beyond the 64 feature cells of the specialised kernels."""
    from oracle import oracle as orc

    model = synth.build_cluster_model(synth.fcc_prim(nspecies=nspecies), {2: 3.0, 3: 3.0})
    sc = synth.build_supercell(model, [3, 3, 4])
    mu = None
    rng = np.random.default_rng(nspecies)
    if step == capi.STEP_FLIP:
        mu = np.tile(rng.uniform(-0.2, 0.2, nspecies)[None, :], (sc.num_sites, 1))
    fm = capi.FEATURES_INTERACTIONS if mode == "int" else capi.FEATURES_CORRELATIONS
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=3), feature_mode=fm, mu_table=mu)
    R = 4
    occ = rng.integers(0, nspecies, (R, sc.num_sites)).astype(np.int32)
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    for e in (eng, ora):
        e.set_state(occ, np.arange(5, 5 + R, dtype=np.uint64), np.linspace(700.0, 4000.0, R))
    for chunk in (1, 20, 150):
        eng.run(chunk)
        ora.run(chunk)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"]), eng.kernel_info()
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(a["features"], b["features"], rtol=1e-10, atol=1e-8)
    np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=1e-9, atol=1e-7)
    assert not eng.kernel_info().startswith("lean")
    eng.close()


@pytest.mark.parametrize("nspecies", [2, 3])
@pytest.mark.parametrize("force", [None, "SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL"], ids=["auto", "general", "universal"])
@pytest.mark.parametrize("mode", ["int", "corr"])
def test_clusters_of_up_to_six_sites(nspecies, force, mode, monkeypatch):
    """Pairs to six-site clusters (SMOLMC_MAX_CLUSTER_SITES; fcc octahedra and their sub-clusters within
    4.2 A): strides over five other members, tensors of 3^6 entries."""
    from oracle import oracle as orc

    for name in ("SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL"):
        monkeypatch.delenv(name, raising=False)
    if force:
        monkeypatch.setenv(force, "1")
    model = synth.build_cluster_model(synth.fcc_prim(nspecies=nspecies), {2: 4.2, 3: 4.2, 4: 4.2, 5: 4.2, 6: 4.2})
    sc = synth.build_supercell(model, [4, 4, 4])
    fm = capi.FEATURES_INTERACTIONS if mode == "int" else capi.FEATURES_CORRELATIONS
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=8), feature_mode=fm)
    R = 5
    rng = np.random.default_rng(4)
    occ = rng.integers(0, nspecies, (R, sc.num_sites)).astype(np.int32)
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    for e in (eng, ora):
        e.set_state(occ, np.arange(5, 5 + R, dtype=np.uint64), np.linspace(700.0, 4000.0, R))
    for chunk in (1, 20, 150):
        eng.run(chunk)
        ora.run(chunk)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"]), eng.kernel_info()
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(a["features"], b["features"], rtol=1e-10, atol=1e-8)
    np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=1e-9, atol=1e-7)
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()
    eng.close()


def _neutral(sc, prim, rng):
    """Random charge-neutral occupancy by greedy single substitutions."""
    q = [[0.0 if c is None else float(c) for c in prim.charges[b]] for b in sc.site_b]
    nsp = np.array([prim.nspecies[b] for b in sc.site_b])
    for _ in range(100):
        occ = (rng.random(sc.num_sites) * nsp).astype(np.int32)
        tot = sum(q[i][occ[i]] for i in range(sc.num_sites))
        for _ in range(20 * sc.num_sites):
            if abs(tot) < 1e-9:
                return occ
            i = int(rng.integers(sc.num_sites))
            c = int(rng.integers(nsp[i]))
            new = tot - q[i][occ[i]] + q[i][c]
            if abs(new) < abs(tot):
                tot, occ[i] = new, c
    raise AssertionError("no neutral occupancy found")


@pytest.mark.parametrize("kind", ["swap", "flip", "table-flip", "wl-swap", "wl-table-flip"])
def test_three_active_sublattices(kind):
    """Rocksalt cations (Li+/Mn3+/Ti4+) and anions (O2-/F-) plus both tetrahedral interstitials
    (Li+/vacancy, one sublattice of two basis sites): four site classes, three active sublattices, a
    CompositionSpace table of three flip vectors across them, Ewald term with a vacancy species."""
    from oracle import oracle as orc

    a = 4.2
    lat = 0.5 * a * np.array([[0, 1, 1], [1, 0, 1], [1, 1, 0]], dtype=float)
    prim = synth.PrimCell(lat, [[0, 0, 0], [.5, .5, .5], [.25, .25, .25], [.75, .75, .75]], [3, 2, 2, 2],
                          charges=[[1.0, 3.0, 4.0], [-2.0, -1.0], [1.0, None], [1.0, None]],
                          species=[["Li+", "Mn3+", "Ti4+"], ["O2-", "F-"], ["Li+", "Vacancy"], ["Li+", "Vacancy"]])
    model = synth.build_cluster_model(prim, {2: 3.2, 3: 2.2})
    sc = synth.build_supercell(model, [3, 3, 2])
    ens = moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model, seed=1), ewald_coefficient=0.1)
    assert [len(s.species) for s in ens.sublattices] == [3, 2, 2] and len(ens.sublattices[2].sites) == 2 * sc.size
    if kind == "flip":
        ens.chemical_potentials = {sp: 0.03 * i for i, sp in enumerate(ens.species)}
    table = "table" in kind
    tab = ens.make_tables(**(dict(flip_table=ens.composition_space().flip_table, swap_weight=0.2) if table else {}))
    R = 4
    rng = np.random.default_rng(12)
    occ = np.array([_neutral(sc, prim, rng) for _ in range(R)])
    step = capi.STEP_TABLE_FLIP if table else capi.STEP_FLIP if kind == "flip" else capi.STEP_SWAP
    if kind.startswith("wl"):
        probe = orc.OracleEvaluator(tab)
        h = np.array([probe.natural_parameters() @ probe.feature_vector(o) for o in occ])
        cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, step, min_enthalpy=float(h.min()) - 6.0371,
                               max_enthalpy=float(h.max()) + 6.0113, bin_size=0.5, check_period=40, flatness=0.2)
    else:
        cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    for e in (eng, ora):
        e.set_state(occ, np.arange(40, 40 + R, dtype=np.uint64), np.linspace(1500.0, 9000.0, R))
    for chunk in (1, 30, 64, 300):
        eng.run(chunk)
        ora.run(chunk)
        x, y = eng.get_state(), ora.get_state()
        assert np.array_equal(x["occupancy"], y["occupancy"]), eng.kernel_info()
        assert np.array_equal(x["n_accepted"], y["n_accepted"])
        np.testing.assert_allclose(x["enthalpy"], y["enthalpy"], rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(x["features"], y["features"], rtol=1e-10, atol=1e-8)
    assert 0 < x["n_accepted"].sum() < x["n_steps"].sum()
    if kind in ("swap", "flip", "table-flip"):
        assert eng.kernel_info().startswith("lean-multi"), eng.kernel_info()
    if table:  # charge neutrality kept
        q = np.zeros((sc.num_sites, 3))
        for i, b in enumerate(sc.site_b):
            q[i, : prim.nspecies[b]] = [0.0 if c is None else c for c in prim.charges[b]]
        assert np.allclose(q[np.arange(sc.num_sites)[None, :], x["occupancy"]].sum(axis=1), 0.0)
    eng.close()
