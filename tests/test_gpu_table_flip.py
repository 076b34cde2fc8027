"""GPU TableFlip kernel against the CPU oracle (identical Philox streams -> identical
trajectories) and the detailed-balance histogram of tests/test_moca/test_mcushers.py:237-319."""

from math import comb

import numpy as np
import pytest

from smol_amd import capi
from tests.test_table_flip import FLIP_TABLE, _model, _neutral_occ

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [
    dict(),
    dict(coef_scale=0.05, mu=[0.1, -0.2, 0.05]),
    dict(coef_scale=0.05, mu=[0.1, -0.2, 0.05], ewald=True),
], ids=["zero-H", "ce+mu", "ce+mu+ewald"])
def test_table_flip_matches_oracle(kw):
    from oracle import oracle as orc
    from smol_amd.engine import Engine

    sc, tab = _model(3, **kw)
    R = 7
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP)
    rng = np.random.default_rng(4)
    occ = np.array([_neutral_occ(sc, 1 + 2 * (r % 5), rng) for r in range(R)])
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(100)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    eng.set_state(occ, seeds, 2500.0)
    ora.set_state(occ, seeds, 2500.0)
    for chunk in (1, 5, 16, 200):
        eng.run(chunk)
        ora.run(chunk)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        assert np.array_equal(a["accepted"], b["accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(a["features"], b["features"], rtol=1e-10, atol=1e-8)
    np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=1e-10, atol=1e-8)
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()
    # the composition moved along the flip direction only
    n = np.array([[(o[: sc.size] == c).sum() for c in range(3)] for o in a["occupancy"]])
    n0 = np.array([[(o[: sc.size] == c).sum() for c in range(3)] for o in occ])
    k = (n - n0)[:, 0]
    assert np.array_equal(n - n0, k[:, None] * FLIP_TABLE[0][None, :])


@pytest.mark.parametrize("dim,ewald", [(3, True), (6, False), (6, True)], ids=["3^3+ewald", "6^3", "6^3+ewald"])
def test_table_flip_proposal_batch_matches_oracle_over_many_blocks(dim, ewald):
    """The kernel proposes 64 consecutive steps at once (lane = step) and re-proposes step by step
    whatever an accepted step made stale (mc_lean.h, propose_batch); the oracle proposes one step
    at a time.  Thousands of steps in launches that start and end inside the 64-step blocks, on a
    small cell (the candidate stream names a site twice all the time: duplicate / fallback paths)
    and a larger one, at a temperature where a quarter of the steps is accepted."""
    from oracle import oracle as orc
    from smol_amd.engine import Engine

    sc, tab = _model(dim, coef_scale=0.05, mu=[0.1, -0.2, 0.05], ewald=ewald)
    R = 9
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP)
    rng = np.random.default_rng(21)
    occ = np.array([_neutral_occ(sc, (sc.size & 1) + 2 * (r % 3 + 1), rng) for r in range(R)])
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(4000)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    eng.set_state(occ, seeds, 6000.0)
    ora.set_state(occ, seeds, 6000.0)
    for chunk in (1, 62, 3, 700, 129, 1105):
        eng.run(chunk)
        ora.run(chunk)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-7)
    acc = a["n_accepted"].sum() / a["n_steps"].sum()
    assert 0.02 < acc < 0.9
    eng.close()


@pytest.mark.parametrize("dims", [(3, 12, 12), (4, 11, 12), (5, 9, 14), (3, 3, 6), (12, 12, 12)], ids=lambda d: "x".join(map(str, d)))
def test_table_flip_field_sweep_on_other_cell_shapes(dims, monkeypatch):
    """The three-flip potential-field sweep of an accepted TableFlip step (field_sweep_gx_multi<3>: chunks of batches
    of nine groups, a shifted tail batch for swaps only, group-by-group and ragged tails) on cells that are not cubes
    -- fields of 432, 528, 630 and 54 entries besides BASELINE config 5's 1728: the same chain as the oracle (dense
    Ewald rows, ewald.pyx:38-58) at temperatures where a third of the steps is accepted, in launches that start and
    end inside the 64-step proposal blocks; the field has not drifted from the occupancies (running trace ==
    from-scratch evaluation); SMOLMC_NO_EWALD_GX (rows of the full site kernel instead of the translation-compressed
    tables): the same chain again.  (Round 6 built two other sweeps on these cases -- a periodic 13.8 KB table walked
    plane by plane, and two entries per lane and gather with the offsets in LDS -- both bit-identical and both
    SLOWER on config 5: NOTES.md.)"""
    from oracle import oracle as orc
    from smol_amd.engine import Engine

    monkeypatch.delenv("SMOLMC_NO_EWALD_GX", raising=False)
    sc, tab = _model(dims, coef_scale=0.05, mu=[0.1, -0.2, 0.05], ewald=True)
    P = sc.size
    R = 4 if P > 1000 else 6
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP)
    rng = np.random.default_rng(7)
    occ = np.array([_neutral_occ(sc, (P & 1) + 2 * (P // 14 + r), rng) for r in range(R)])
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(900)
    temps = np.linspace(5000.0, 20000.0, R)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    assert eng.kernel_info().startswith("lean") and "gx=1x" + "x".join(map(str, dims)) in eng.kernel_info(), eng.kernel_info()
    monkeypatch.setenv("SMOLMC_NO_EWALD_GX", "1")
    rows = Engine(tab, cfg)
    monkeypatch.delenv("SMOLMC_NO_EWALD_GX")
    assert "gx=" not in rows.kernel_info() and "field=1" in rows.kernel_info(), rows.kernel_info()
    for e in (eng, ora, rows):
        e.set_state(occ, seeds, temps)
    for chunk in (1, 62, 3, 400) if P > 1000 else (1, 62, 3, 700, 129, 1105):
        for e in (eng, ora, rows):
            e.run(chunk)
        a, b, c = eng.get_state(), ora.get_state(), rows.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(a["features"], b["features"], rtol=1e-10, atol=1e-8)
        assert np.array_equal(a["occupancy"], c["occupancy"])
        np.testing.assert_allclose(a["enthalpy"], c["enthalpy"], rtol=1e-11, atol=1e-9)
    acc = a["n_accepted"].sum() / a["n_steps"].sum()
    assert 0.1 < acc < 0.9, acc
    np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=1e-10, atol=1e-7)
    eng.close()
    rows.close()


def test_launch_order_of_a_ladder_does_not_change_any_walker(monkeypatch):
    """On an exchange ladder the engine deals the walkers to launch slots hottest-with-coldest
    (engine.hip, update_walker_order); which slot runs a walker must not matter."""
    from smol_amd.engine import Engine

    sc, tab = _model(6, coef_scale=0.05, mu=[0.1, -0.2, 0.05], ewald=True)
    R = 24
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP)
    rng = np.random.default_rng(3)
    occ = np.array([_neutral_occ(sc, 2 * (r % 3 + 1), rng) for r in range(R)])
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(77)
    temps = rng.permutation(np.geomspace(1500.0, 9000.0, R))
    states = []
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("SMOLMC_WALKER_ORDER", mode)
        eng = Engine(tab, cfg)
        eng.set_state(occ, seeds, temps)
        eng.run(300)
        eng.set_temperature(temps[::-1].copy())  # (a new ladder assignment: the order is rebuilt)
        eng.run(300)
        states.append(eng.get_state())
        eng.close()
    for s in states[1:]:
        assert np.array_equal(s["occupancy"], states[0]["occupancy"])
        assert np.array_equal(s["n_accepted"], states[0]["n_accepted"])
        assert np.array_equal(s["enthalpy"], states[0]["enthalpy"])
    assert states[0]["n_accepted"].sum() > 0


def test_table_flip_detailed_balance_on_gpu():
    from smol_amd.engine import Engine

    sc, tab = _model(3)
    R = 2048
    eng = Engine(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP))
    rng = np.random.default_rng(6)
    occ = np.array([_neutral_occ(sc, 1 + 2 * (r % 5), rng) for r in range(R)])
    eng.set_state(occ, np.arange(R, dtype=np.uint64) + np.uint64(5), 1000.0)
    eng.run(2000)
    counts = np.zeros(10)
    for _ in range(10):
        eng.run(200)
        o = eng.get_state()["occupancy"]
        counts += np.bincount((o[:, : sc.size] == 2).sum(axis=1), minlength=10)
    P = sc.size
    w = {k: comb(P, k) * comb(P - k, (P - 3 * k) // 2) for k in (1, 3, 5, 7, 9)}
    tot = sum(w.values())
    for k, wt in w.items():
        p = wt / tot
        assert counts[k] / counts.sum() == pytest.approx(p, abs=max(0.01, 5 * np.sqrt(p / counts.sum())))
    assert counts[[0, 2, 4, 6, 8]].sum() == 0


def test_table_flip_through_sampler():
    from smol_amd import moca, synth

    model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 3.5})
    sc = synth.build_supercell(model, [3, 3, 3])
    ens = moca.Ensemble.from_cluster_expansion(
        sc, synth.random_coefs(model, seed=5, scale=0.05),
        chemical_potentials={"Li+": 0.1, "Mn3+": -0.2, "Ti4+": 0.05})
    auto = moca.Sampler.from_ensemble(ens, temperature=2000, step_type="table-flip")
    assert np.abs(auto.mckernels[0].usher_kwargs["flip_table"]).tolist() == [[1, 3, 2, 0]]  # CompositionSpace
    # reference-style table: columns for ALL sublattices (cations, then the inactive anions)
    sampler = moca.Sampler.from_ensemble(ens, temperature=2000, step_type="table-flip", nwalkers=3,
                                         seeds=[1, 2, 3], flip_table=[[1, -3, 2, 0]], swap_weight=0.2)
    rng = np.random.default_rng(8)
    occ = np.array([_neutral_occ(sc, 3, rng) for _ in range(3)])
    sampler.run(600, occ, thin_by=100)
    c = sampler.samples
    occs = c.get_occupancies(flat=False)
    charge = np.array([1, 3, 4])
    assert np.all(charge[occs[..., : sc.size]].sum(axis=-1) == 2 * sc.size)  # neutral throughout
    f = ens.compute_feature_vector(occs[-1, 0])
    np.testing.assert_allclose(c.get_feature_vectors(flat=False)[-1, 0], f, rtol=1e-10, atol=1e-8)


@pytest.mark.parametrize("ewald", [False, True], ids=["ce", "ce+ewald"])
def test_table_flip_across_two_sublattices(ewald):
    """TableFlip with flip vectors spanning the cation AND the anion sublattice (the shape of the
    reference's own TableFlip tests, tests/test_moca/test_mcushers.py:199-319): the table comes
    from CompositionSpace, the engine runs mc_table_multi_kernel, trajectories equal the
    oracle's, every sample stays charge neutral and the composition moves on both sublattices."""
    from oracle import oracle as orc
    from smol_amd import moca, synth
    from smol_amd.engine import Engine

    model = synth.build_cluster_model(synth.rocksalt_prim(anion_charges=(-2.0, -1.0)), {2: 4.5})
    sc = synth.build_supercell(model, [3, 3, 3])
    ens = moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model, seed=8, scale=0.02),
                                               ewald_coefficient=0.05 if ewald else None)
    cs = ens.composition_space(optimize_basis=True, table_ergodic=True)
    table = np.asarray(cs.flip_table)
    assert table.shape[1] == 5 and len(table) >= 2
    assert np.any(table[:, :3] != 0) and np.any(table[:, 3:] != 0)  # couples both sublattices
    tab = ens.make_tables(flip_table=table, swap_weight=0.15)
    R = 6
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP)
    P = sc.size
    rng = np.random.default_rng(13)
    occ = np.zeros((R, sc.num_sites), dtype=np.int32)
    for r in range(R):  # 17 Li+ + 8 Mn3+ + 2 Ti4+ = +49, 22 O2- + 5 F- = -49
        perm = rng.permutation(P)
        occ[r, perm[:8]] = 1
        occ[r, perm[8:10]] = 2
        occ[r, P + rng.permutation(P)[:5]] = 1
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(900)
    temps = np.linspace(1500.0, 6000.0, R)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    assert eng.kernel_info().startswith("lean-multi")
    eng.set_state(occ, seeds, temps)
    ora.set_state(occ, seeds, temps)
    q = np.zeros((sc.num_sites, 3))
    q[:P] = [1, 3, 4]
    q[P:, :2] = [-2, -1]
    comps = set()
    for chunk in (1, 9, 40, 300, 700):
        eng.run(chunk)
        ora.run(chunk)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(a["features"], b["features"], rtol=1e-10, atol=1e-8)
        charge = q[np.arange(sc.num_sites)[None, :], a["occupancy"]].sum(axis=1)
        assert np.all(charge == 0)
        for o in a["occupancy"]:
            comps.add((int((o[:P] == 1).sum()), int((o[:P] == 2).sum()), int((o[P:] == 1).sum())))
    np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=1e-10, atol=1e-8)
    assert len({c[2] for c in comps}) > 1 and len({c[0] for c in comps}) > 1  # F and Mn contents moved
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()


def test_two_sublattice_table_flip_detailed_balance():
    """Zero Hamiltonian on cation {Li+, Mn3+, Ti4+} + anion {O2-, F-} sublattices (27 sites each):
    the chain must visit a charge-neutral composition (n_Mn, n_Ti, n_F) with probability
    proportional to its number of configurations, 27!/(n_Li! n_Mn! n_Ti!) * C(27, n_F) -- the
    histogram test of tests/test_moca/test_mcushers.py:237-319 on the engine, with the flip table
    from CompositionSpace (ergodic completion) and 10 % canonical swaps."""
    from math import comb

    from smol_amd import moca, synth
    from smol_amd.engine import Engine

    model = synth.build_cluster_model(synth.rocksalt_prim(anion_charges=(-2.0, -1.0)), {2: 3.5})
    sc = synth.build_supercell(model, [3, 3, 3])
    ens = moca.Ensemble.from_cluster_expansion(sc, np.zeros(model.num_corr_functions))
    table = ens.composition_space(optimize_basis=True, table_ergodic=True).flip_table
    tab = ens.make_tables(flip_table=table, swap_weight=0.1)
    R, P = 256, sc.size
    eng = Engine(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP))
    assert eng.kernel_info().startswith("lean-multi")
    rng = np.random.default_rng(21)
    occ = np.zeros((R, sc.num_sites), dtype=np.int32)
    for r in range(R):  # start: 6 Mn3+, 2 Ti4+, 9 F-  (charge 19 + 18 + 8 = 45 = 2*18 + 9)
        perm = rng.permutation(P)
        occ[r, perm[:6]] = 1
        occ[r, perm[6:8]] = 2
        occ[r, P + rng.permutation(P)[:9]] = 1
    eng.set_state(occ, np.arange(R, dtype=np.uint64) + np.uint64(77), 1000.0)
    eng.run(4000)
    counts = {}
    nsamp = 60
    for _ in range(nsamp):
        eng.run(400)
        o = eng.get_state()["occupancy"]
        keys = zip((o[:, :P] == 1).sum(axis=1), (o[:, :P] == 2).sum(axis=1), (o[:, P:] == 1).sum(axis=1))
        for k in keys:
            k = tuple(int(x) for x in k)
            counts[k] = counts.get(k, 0) + 1
    w = {}
    for n_mn in range(P + 1):
        for n_ti in range(P + 1 - n_mn):
            n_f = 54 - ((P - n_mn - n_ti) + 3 * n_mn + 4 * n_ti)
            if 0 <= n_f <= P:
                w[(n_mn, n_ti, n_f)] = comb(P, n_mn) * comb(P - n_mn, n_ti) * comb(P, n_f)
    assert set(counts) <= set(w)  # only charge-neutral compositions are ever visited
    tot_w, tot_c = sum(w.values()), sum(counts.values())
    top = sorted(w, key=lambda k: -w[k])[:8]
    for k in top:
        p = w[k] / tot_w
        assert counts.get(k, 0) / tot_c == pytest.approx(p, abs=max(0.012, 5 * np.sqrt(p / tot_c)))


def _wl_window(tab, occ, seeds, R, nbins=23.5, update_period=1):
    """A Wang-Landau window the walkers start inside and try to leave: a third of the enthalpy range a hot
    Metropolis chain of the oracle covers, on either side of the starting enthalpies."""
    from oracle import oracle as orc

    probe = orc.OracleMC(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP))
    probe.set_state(occ, seeds, 1.0e4)
    h0 = probe.get_state()["enthalpy"]
    probe.run(300)
    h1 = probe.get_state()["enthalpy"]
    pad = (max(h0.max(), h1.max()) - min(h0.min(), h1.min())) / 3 + 1e-3
    lo, hi = h0.min() - pad, h0.max() + pad
    return capi.make_config(R, capi.KERNEL_WANGLANDAU, capi.STEP_TABLE_FLIP, min_enthalpy=lo, max_enthalpy=hi,
                            bin_size=(hi - lo) / nbins, check_period=97, flatness=0.2, update_period=update_period)


def _wl_table_chain(tab, cfg, occ, seeds, family, monkeypatch):
    """The lean kernel of `family`, the oracle and the universal kernel (SMOLMC_NO_TABLE_WL) on one Wang-Landau
    TableFlip chain: occupancies, accept flags, entropies, histograms, occurrences bit-exact, enthalpies / features /
    per-bin mean features to 1e-10; then a sampled trace against the oracle's states at the sample times."""
    from oracle import oracle as orc
    from smol_amd.engine import Engine

    monkeypatch.delenv("SMOLMC_NO_TABLE_WL", raising=False)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    monkeypatch.setenv("SMOLMC_NO_TABLE_WL", "1")
    univ = Engine(tab, cfg)
    monkeypatch.delenv("SMOLMC_NO_TABLE_WL")
    assert eng.kernel_info().startswith(family[0]) and family[1] in eng.kernel_info(), eng.kernel_info()
    assert univ.kernel_info().startswith("universal"), univ.kernel_info()
    for e in (eng, ora, univ):
        e.set_state(occ, seeds, 0.0)
    for chunk in (1, 62, 3, 700, 129, 1105):
        for e in (eng, ora, univ):
            e.run(chunk)
        a, b, c = eng.get_state(), ora.get_state(), univ.get_state()
        x, y, z = eng.get_wl(), ora.get_wl(), univ.get_wl()
        for s, w in ((b, y), (c, z)):
            assert np.array_equal(a["occupancy"], s["occupancy"])
            assert np.array_equal(a["n_accepted"], s["n_accepted"])
            assert np.array_equal(a["accepted"], s["accepted"])
            np.testing.assert_allclose(a["enthalpy"], s["enthalpy"], rtol=1e-10, atol=1e-8)
            np.testing.assert_allclose(a["features"], s["features"], rtol=1e-10, atol=1e-8)
            assert np.array_equal(x["histogram"], w["histogram"])
            assert np.array_equal(x["occurrences"], w["occurrences"])
            np.testing.assert_allclose(x["entropy"], w["entropy"], rtol=0, atol=0)
            np.testing.assert_allclose(x["mean_features"], w["mean_features"], rtol=1e-10, atol=1e-8)
            np.testing.assert_allclose(x["mod_factor"], w["mod_factor"])
    acc = a["n_accepted"].sum() / a["n_steps"].sum()
    assert 0.05 < acc < 0.98, acc
    assert (x["mod_factor"] < 1.0).any()  # the flatness branch fired
    np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=1e-10, atol=1e-7)
    rows = eng.run_sampled(5, 40)
    for k in range(5):
        ora.run(40)
        b = ora.get_state()
        assert np.array_equal(rows["accepted"][k], b["accepted"])
        np.testing.assert_allclose(rows["enthalpy"][k], b["enthalpy"], rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(rows["features"][k], b["features"], rtol=1e-10, atol=1e-8)
    y, x = ora.get_wl(), eng.get_wl()
    assert np.array_equal(x["histogram"], y["histogram"])
    np.testing.assert_allclose(x["mean_features"], y["mean_features"], rtol=1e-10, atol=1e-8)
    eng.close()
    univ.close()


@pytest.mark.parametrize("kw", [
    dict(coef_scale=0.05),
    dict(coef_scale=0.05, mu=[0.1, -0.2, 0.05]),
    dict(coef_scale=0.05, mu=[0.1, -0.2, 0.05], ewald=True),
], ids=["ce", "ce+mu", "ce+mu+ewald"])
@pytest.mark.parametrize("dim,update_period", [(3, 1), (6, 1), (3, 3), (6, 2)], ids=["3^3", "6^3", "3^3-update3", "6^3-update2"])
def test_wang_landau_table_flip_on_the_lean_table_kernel(dim, update_period, kw, monkeypatch):
    """Wang-Landau with TableFlip proposals (the reference composes any usher with any kernel, kernel/base.py:192-239;
    the accept rule takes the step's a-priori factor, wanglandau.py:197-198) on mc_table_kernel<..., WLT> (round 6;
    the universal kernel until then): the oracle's chain on the native stream in launches that start and end inside
    the 64-step proposal blocks, with flatness checks that fire, steps that leave the window, a sampled trace, and the
    same chain again from the universal kernel.  update_period > 1: entropies / histograms every few steps and the
    per-bin mean features as running means in a cache of rows (wanglandau.py:233-245), as mc_lean_multi_kernel's WLK."""
    sc, tab = _model(dim, **kw)
    R = 6
    rng = np.random.default_rng(31)
    occ = np.array([_neutral_occ(sc, (sc.size & 1) + 2 * (r % 3 + 1), rng) for r in range(R)])
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(7100)
    _wl_table_chain(tab, _wl_window(tab, occ, seeds, R, update_period=update_period), occ, seeds,
                    ("lean ", "wl=table" + ("-mean" if update_period > 1 else "")), monkeypatch)


@pytest.mark.parametrize("update_period", [1, 3])
@pytest.mark.parametrize("ewald", [False, True], ids=["ce", "ce+ewald"])
def test_wang_landau_table_flip_across_two_sublattices(ewald, update_period, monkeypatch):
    """... and with flip vectors that span the cation and the anion sublattice (the shape of the reference's own
    TableFlip tests, tests/test_moca/test_mcushers.py:199-319) on mc_table_multi_kernel<..., WLT>."""
    from smol_amd import moca, synth

    model = synth.build_cluster_model(synth.rocksalt_prim(anion_charges=(-2.0, -1.0)), {2: 4.5})
    sc = synth.build_supercell(model, [3, 3, 3])
    ens = moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model, seed=8, scale=0.02),
                                               ewald_coefficient=0.05 if ewald else None)
    table = np.asarray(ens.composition_space(optimize_basis=True, table_ergodic=True).flip_table)
    tab = ens.make_tables(flip_table=table, swap_weight=0.15)
    R, P = 6, sc.size
    rng = np.random.default_rng(13)
    occ = np.zeros((R, sc.num_sites), dtype=np.int32)
    for r in range(R):  # 17 Li+ + 8 Mn3+ + 2 Ti4+ = +49, 22 O2- + 5 F- = -49
        perm = rng.permutation(P)
        occ[r, perm[:8]] = 1
        occ[r, perm[8:10]] = 2
        occ[r, P + rng.permutation(P)[:5]] = 1
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(1900)
    _wl_table_chain(tab, _wl_window(tab, occ, seeds, R, update_period=update_period), occ, seeds,
                    ("lean-multi", "wl=multi" + ("-mean" if update_period > 1 else "")), monkeypatch)
