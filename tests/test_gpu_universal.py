"""The universal kernel (mc_univ.h) -- the backstop for everything the specialised kernels decline --
against the CPU oracle on the engine's own random stream (identical Philox words -> identical
chains): every step type x kernel type x feature mode x bias the reference composes
(kernel/base.py:192-239), occupancy in LDS and in HBM, models the other kernels cannot hold."""

import numpy as np
import pytest

from smol_amd import capi
from tests.cases import CASES, load_case, tables_for
from tests.v6_cases import SPECS, T6, build

pytestmark = pytest.mark.gpu
MODES = {"int": capi.FEATURES_INTERACTIONS, "corr": capi.FEATURES_CORRELATIONS}


def _pair(tab, cfg, occ, seeds, temp):
    from oracle import oracle as orc
    from smol_amd.engine import Engine

    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    eng.set_state(occ, seeds, temp)
    ora.set_state(occ, seeds, temp)
    return eng, ora


def _same_chain(eng, ora, chunks, wl=False, bias=False):
    for n in chunks:
        eng.run(n)
        ora.run(n)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        assert np.array_equal(a["accepted"], b["accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(a["features"], b["features"], rtol=1e-10, atol=1e-8)
        if bias:
            np.testing.assert_allclose(eng.get_bias(), ora.get_bias(), rtol=1e-10, atol=1e-9)
        if wl:
            x, y = eng.get_wl(), ora.get_wl()
            assert np.array_equal(x["histogram"], y["histogram"])
            assert np.array_equal(x["occurrences"], y["occurrences"])
            np.testing.assert_allclose(x["entropy"], y["entropy"], rtol=0, atol=0)
            np.testing.assert_allclose(x["mean_features"], y["mean_features"], rtol=1e-10, atol=1e-8)
            np.testing.assert_allclose(x["mod_factor"], y["mod_factor"])
    return a


def _rand_occ(sc, rng, R):
    nsp = np.array([sc.model.prim.nspecies[b] for b in sc.site_b])
    return (rng.random((R, sc.num_sites)) * nsp).astype(np.int32)


@pytest.mark.parametrize("occ_mem", ["lds", "hbm"])
@pytest.mark.parametrize("step", [capi.STEP_FLIP, capi.STEP_SWAP], ids=["flip", "swap"])
@pytest.mark.parametrize("mode", ["int", "corr"])
@pytest.mark.parametrize("name", list(CASES))
def test_flip_and_swap_chains_on_the_universal_kernel(name, mode, step, occ_mem, monkeypatch):
    """Every golden model (pairs, triplets, ternary indicator basis on a skew cell, aliased 2x2x2,
    vacancies, two active sublattices, Ewald): the universal kernel runs the oracle's chain."""
    monkeypatch.setenv("SMOLMC_FORCE_UNIVERSAL", "1")
    if occ_mem == "hbm":
        monkeypatch.setenv("SMOLMC_UNIV_OCC_HBM", "1")
    else:
        monkeypatch.delenv("SMOLMC_UNIV_OCC_HBM", raising=False)
    c = load_case(name)
    sc = c["sc"]
    mu = None
    if step == capi.STEP_FLIP:
        mu = np.zeros((sc.num_sites, 3))
        mu[:, :] = np.random.default_rng(5).uniform(-0.3, 0.3, 3)[None, :]
    tab = tables_for(name, MODES[mode], mu_table=mu)
    R = 5
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    occ = _rand_occ(sc, np.random.default_rng(8), R)
    eng, ora = _pair(tab, cfg, occ, np.arange(R, dtype=np.uint64) + np.uint64(50), 2500.0)
    info = eng.kernel_info()
    assert info.startswith("universal occ=" + occ_mem), info
    a = _same_chain(eng, ora, (1, 7, 150))
    np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=1e-10, atol=1e-8)
    assert 0 < a["n_accepted"].sum()
    eng.close()


@pytest.mark.parametrize("tag", ["TC_tf_int", "TC_tf_corr", "TC_tfw_int", "TC_tflim_int", "TC_tffug_int", "TC_tfwl_int",
                                 "TG_tf_int", "TG_tf_corr", "TG6_tf_int"])
@pytest.mark.parametrize("how", ["general-handle", "universal-handle", "hbm"])
def test_table_flip_chains_on_the_universal_kernel(tag, how, monkeypatch):
    """TableFlip outside the lean families (SMOLMC_FORCE_GENERAL: the reference's own TableFlip shapes
    -- cation table, cation + anion table -- with correlation features, Wang-Landau, a bias term) on
    the native stream: same chain as the oracle."""
    for k in ("SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL", "SMOLMC_UNIV_OCC_HBM"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv({"general-handle": "SMOLMC_FORCE_GENERAL", "universal-handle": "SMOLMC_FORCE_UNIVERSAL",
                        "hbm": "SMOLMC_FORCE_UNIVERSAL"}[how], "1")
    if how == "hbm":
        monkeypatch.setenv("SMOLMC_UNIV_OCC_HBM", "1")
    R = 4
    tab, cfg, occ0, temp = build(tag, n_replicas=R)
    sp = SPECS[tag]
    eng, ora = _pair(tab, cfg, np.tile(occ0, (R, 1)), np.arange(R, dtype=np.uint64) + np.uint64(900), temp)
    assert eng.kernel_info().startswith("universal"), eng.kernel_info()
    a = _same_chain(eng, ora, (1, 5, 64, 300), wl="wl" in sp, bias="bias" in sp)
    acc = a["n_accepted"].sum() / a["n_steps"].sum()
    assert 0.01 < acc < 0.99
    eng.close()


@pytest.mark.parametrize("tag", ["BC_fug_flip_int", "BC_sqc_flip_corr", "BG_hyp_flip_int", "BG_sqc_swap_int", "BG_fug_flip_corr"])
def test_biased_chains_on_the_universal_kernel(tag, monkeypatch):
    monkeypatch.setenv("SMOLMC_FORCE_UNIVERSAL", "1")
    R = 4
    tab, cfg, occ0, temp = build(tag, n_replicas=R)
    eng, ora = _pair(tab, cfg, np.tile(occ0, (R, 1)), np.arange(R, dtype=np.uint64) + np.uint64(77), temp)
    _same_chain(eng, ora, (1, 9, 400), bias=True)
    eng.close()


@pytest.mark.parametrize("update_period", [1, 3])
def test_wang_landau_on_the_universal_kernel(update_period, monkeypatch):
    monkeypatch.setenv("SMOLMC_FORCE_UNIVERSAL", "1")
    tab, cfg, occ0, _ = build("B_wlup3", n_replicas=3)
    cfg.wl_update_period = update_period
    eng, ora = _pair(tab, cfg, np.tile(occ0, (3, 1)), [4, 5, 6], 0.0)
    _same_chain(eng, ora, (1, 59, 61, 500), wl=True)
    assert (eng.get_wl()["mod_factor"] < 1.0).any()  # the flatness branch fired
    eng.close()


def test_wang_landau_update_period_on_the_general_kernel(monkeypatch):
    """update_period > 1 on mc_kernel (its path until round 5; SMOLMC_NO_WL_MULTI keeps it reachable):
    native stream vs the oracle."""
    monkeypatch.delenv("SMOLMC_FORCE_UNIVERSAL", raising=False)
    monkeypatch.setenv("SMOLMC_NO_WL_MULTI", "1")
    tab, cfg, occ0, _ = build("B_wlup3", n_replicas=3)
    eng, ora = _pair(tab, cfg, np.tile(occ0, (3, 1)), [4, 5, 6], 0.0)
    assert eng.kernel_info().startswith("general"), eng.kernel_info()
    _same_chain(eng, ora, (1, 59, 61, 500), wl=True)
    eng.close()


def test_device_side_sampling_on_the_universal_kernel(monkeypatch):
    monkeypatch.setenv("SMOLMC_FORCE_UNIVERSAL", "1")
    R = 3
    tab, cfg, occ0, temp = build("TG_tf_int", n_replicas=R)
    eng, ora = _pair(tab, cfg, np.tile(occ0, (R, 1)), [1, 2, 3], temp)
    smp = eng.run_sampled(6, 25)
    for i in range(6):
        ora.run(25)
        b = ora.get_state()
        assert np.array_equal(smp["occupancy"][i], b["occupancy"])
        assert np.array_equal(smp["accepted"][i], b["accepted"])
        np.testing.assert_allclose(smp["enthalpy"][i], b["enthalpy"], rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(smp["features"][i], b["features"], rtol=1e-10, atol=1e-8)
    eng.close()


def test_more_than_1024_clusters_per_site(monkeypatch):
    """1429 clusters per site (pairs <= 7 A, triplets <= 6.5 A on FCC): beyond the 16 slot groups of
    mc_kernel -- the reference has no such limit (processor/expansion.py:120-163); the universal
    kernel walks the reference's own per-site tables."""
    from smol_amd import synth

    monkeypatch.delenv("SMOLMC_FORCE_UNIVERSAL", raising=False)
    model = synth.build_cluster_model(synth.fcc_prim(), {2: 7.0, 3: 6.5})
    sc = synth.build_supercell(model, [7, 7, 7])
    assert sum(rows.shape[0] for _, rows, _ in sc.local_tables()[0]) > 1024
    coefs = synth.random_coefs(model, seed=3)
    R = 3
    occ = (np.random.default_rng(1).random((R, sc.num_sites)) < 0.4).astype(np.int32)
    for mode, step in (("int", capi.STEP_SWAP), ("corr", capi.STEP_FLIP)):
        tab = capi.TableSet.from_synth(sc, coefs, feature_mode=MODES[mode])
        eng, ora = _pair(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, step), occ, [7, 8, 9], 1500.0)
        assert "more than 1024 clusters per site" in eng.kernel_info(), eng.kernel_info()
        a = _same_chain(eng, ora, (1, 40, 160))
        np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=1e-10, atol=1e-7)
        assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()
        eng.close()


def test_occupancy_beyond_lds_runs_from_hbm(monkeypatch):
    """A 64^3 binary FCC cell (262 144 sites): the occupancy of one walker exceeds a workgroup's
    LDS, the universal kernel keeps it in HBM (one byte per site, L2-resident) -- three walkers
    against the oracle."""
    from smol_amd import synth

    monkeypatch.delenv("SMOLMC_FORCE_UNIVERSAL", raising=False)
    monkeypatch.delenv("SMOLMC_UNIV_OCC_HBM", raising=False)
    model = synth.build_cluster_model(synth.fcc_prim(), {2: 3.0})
    sc = synth.build_supercell(model, [64, 64, 64])
    assert sc.num_sites == 262144
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=4))
    R = 3
    occ = (np.random.default_rng(2).random((R, sc.num_sites)) < 0.5).astype(np.int32)
    eng, ora = _pair(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP), occ, [1, 2, 3], 900.0)
    info = eng.kernel_info()
    assert info.startswith("universal occ=hbm") and "does not fit LDS" in info, info
    a = _same_chain(eng, ora, (1, 64, 400))
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()
    assert np.array_equal(a["occupancy"].sum(axis=1), occ.sum(axis=1))  # canonical
    eng.close()


def test_more_than_eight_flip_vectors(monkeypatch):
    """Nine flip vectors (integer combinations of the two basis vectors of the cation + anion table,
    up to eight flips per step): beyond the eight vector registers of the lean TableFlip kernels --
    the reference takes any table (mcusher.py:397-551); oracle and universal kernel agree."""
    monkeypatch.delenv("SMOLMC_FORCE_UNIVERSAL", raising=False)
    monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    from tests.cases import load_case

    v1, v2 = np.array([1, -3, 2, 0, 0]), np.array([1, -1, 0, -2, 2])
    table = np.array([v1, v2, v1 + v2, 2 * v1, v1 - v2, 2 * v2, 2 * v1 - v2, v1 - 2 * v2, 2 * v1 - 2 * v2])
    assert len(table) == 9 and max(int(-r[r < 0].sum()) for r in table) == 8
    c = load_case("rocksalt333_two_sublattices")
    weights = np.linspace(0.5, 2.0, 18)
    tab = capi.TableSet.from_synth(c["sc"], c["coefs"], ewald=c["ewald"], ewald_coef=0.1, mu_table=T6["TG_mu"],
                                   flip_table=table, flip_weights=weights, swap_weight=0.15)
    R = 5
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP)
    eng, ora = _pair(tab, cfg, np.tile(T6["TG_occ0"], (R, 1)), np.arange(R, dtype=np.uint64) + np.uint64(31), 6000.0)
    assert eng.kernel_info().startswith("universal"), eng.kernel_info()
    a = _same_chain(eng, ora, (1, 30, 500))
    assert 0.01 < a["n_accepted"].sum() / a["n_steps"].sum() < 0.99
    eng.close()


def test_wang_landau_with_more_than_64_features(monkeypatch):
    """A five-species FCC model in correlation mode has 65 correlation functions: beyond the 64
    lane-indexed features of the Wang-Landau bookkeeping in mc_kernel / mc_wl_kernel (refused before
    round 4; the reference has no limit, wanglandau.py:117-118)."""
    from smol_amd import synth

    monkeypatch.delenv("SMOLMC_FORCE_UNIVERSAL", raising=False)
    model = synth.build_cluster_model(synth.fcc_prim(nspecies=5), {2: 6.0, 3: 3.0})
    assert model.num_corr_functions > 64
    sc = synth.build_supercell(model, [4, 4, 4])
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=6), feature_mode=capi.FEATURES_CORRELATIONS)
    R = 3
    occ = (np.random.default_rng(5).random((R, sc.num_sites)) * 5).astype(np.int32)
    from oracle import oracle as orc

    ev = orc.OracleEvaluator(tab)
    h = np.array([ev.natural_parameters() @ ev.feature_vector(o) for o in occ])
    cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, capi.STEP_FLIP, min_enthalpy=float(h.min()) - 20.37,
                           max_enthalpy=float(h.max()) + 20.11, bin_size=0.5, check_period=80)
    eng, ora = _pair(tab, cfg, occ, [11, 12, 13], 0.0)
    assert "more than 64 features" in eng.kernel_info(), eng.kernel_info()
    _same_chain(eng, ora, (1, 79, 2, 400), wl=True)
    eng.close()


@pytest.mark.parametrize("bias", [None, "fug", "sqc", "hyp"])
@pytest.mark.parametrize("mode", ["int", "corr"])
def test_replayed_records_of_arbitrary_flips(mode, bias, monkeypatch):
    """Records of one to eight flips drawn at random -- several sublattices in one step, the SAME site
    flipped more than once (what a MultiStep usher returns, mcusher.py:288-304) -- replayed on a Flip
    handle (the universal kernel takes them) against the oracle: sequential-flip semantics of the
    features (expansion.py:217-229), mu against the occupancy before the step (ensemble.py:368-374),
    'the last flip of a site counts' for the bias terms (bias.py:75-93, 188-206)."""
    from oracle import oracle as orc
    from smol_amd.engine import Engine

    monkeypatch.delenv("SMOLMC_FORCE_UNIVERSAL", raising=False)
    c = load_case("rocksalt333_two_sublattices")
    sc = c["sc"]
    tab = capi.TableSet.from_synth(sc, c["coefs"], feature_mode=MODES[mode], ewald=c["ewald"], ewald_coef=0.1,
                                   mu_table=T6["TG_mu"])
    if bias == "fug":
        tab.set_bias(capi.BIAS_FUGACITY, T6["BG_fug_table"])
    elif bias == "sqc":
        tab.set_bias(capi.BIAS_SQUARE_CHARGE, T6["BG_sqc_table"], 0.04)
    elif bias == "hyp":
        from tests.v6_cases import _hyperplane_tables

        tab.set_bias(capi.BIAS_SQUARE_HYPERPLANE, _hyperplane_tables(T6["BG_hyp_A"], T6["BG_hyp_dim_ids"]), 0.02,
                     intercepts=T6["BG_hyp_b"].astype(float))
    R, n = 4, 300
    rng = np.random.default_rng(17)
    nsp = np.array([sc.model.prim.nspecies[b] for b in sc.site_b])
    occ = (rng.random((R, sc.num_sites)) * nsp).astype(np.int32)
    steps = -np.ones((R, n, 16), dtype=np.int32)
    for r in range(R):
        for k in range(n):
            nf = int(rng.integers(0, 9))
            sites = rng.integers(0, sc.num_sites, nf)
            if nf >= 3 and rng.random() < 0.5:
                sites[-1] = sites[0]  # a site flipped twice in one step
            for j, s in enumerate(sites):
                steps[r, k, 2 * j], steps[r, k, 2 * j + 1] = s, rng.integers(0, nsp[s])
    us = rng.random((R, n))
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_FLIP)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    for e in (eng, ora):
        e.set_state(occ, np.arange(R, dtype=np.uint64), 3000.0)
    a_acc, a_H = eng.replay(steps, us)
    b_acc, b_H = ora.replay(steps, us)
    assert np.array_equal(a_acc, b_acc)
    np.testing.assert_allclose(a_H, b_H, rtol=1e-10, atol=1e-8)
    a, b = eng.get_state(), ora.get_state()
    assert np.array_equal(a["occupancy"], b["occupancy"])
    np.testing.assert_allclose(a["features"], b["features"], rtol=1e-10, atol=1e-8)
    if bias:
        np.testing.assert_allclose(eng.get_bias(), ora.get_bias(), rtol=1e-10, atol=1e-9)
    assert 0.05 < a_acc.mean() < 0.95
    # a one-step chain of the same records on the native kernels afterwards: the handle is still consistent
    eng.run(50)
    ora.run(50)
    assert np.array_equal(eng.get_state()["occupancy"], ora.get_state()["occupancy"])
    eng.close()
