"""Wang-Landau on the multi-class layout (mc_lean_multi_kernel<..., WLK>, round 5): several active
sublattices / site classes, any update_period, mu rows, the Ewald field in LDS or HBM -- the reference
composes any kernel with any ensemble (kernel/base.py:192-239, kernel/wanglandau.py:17-130), and its own
shipped model (LiNiO2: Li/vacancy + Ni3+/Ni4+) is of this class.  Every case: the engine's native stream
against the oracle on identical Philox words -- occupancies, counters, histograms, occurrences and
entropies bit-equal, enthalpies / features / mean features to 1e-10."""

import os

import numpy as np
import pytest

from smol_amd import capi
from tests.cases import load_case, tables_for
from tests.v6_cases import build

pytestmark = pytest.mark.gpu
MODES = {"int": capi.FEATURES_INTERACTIONS, "corr": capi.FEATURES_CORRELATIONS}
ENV = ("SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL", "SMOLMC_NO_WL_MULTI", "SMOLMC_NO_LEAN_MULTI",
       "SMOLMC_MULTI_PHI_HBM", "SMOLMC_MULTI_PHI_LDS", "SMOLMC_WL_RUNNING_MEAN", "SMOLMC_REPLAY_GENERAL",
       "SMOLMC_REPLAY_UNIVERSAL", "SMOLMC_LAUNCH_CHUNK")


def _clean(monkeypatch):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)


def _pair(tab, cfg, occ, seeds):
    from oracle import oracle as orc
    from smol_amd.engine import Engine

    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    eng.set_state(occ, seeds, 0.0)
    ora.set_state(occ, seeds, 0.0)
    return eng, ora


def _compare(eng, ora):
    a, b = eng.get_state(), ora.get_state()
    assert np.array_equal(a["occupancy"], b["occupancy"])
    assert np.array_equal(a["n_accepted"], b["n_accepted"])
    assert np.array_equal(a["accepted"], b["accepted"])
    np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-8)
    np.testing.assert_allclose(a["features"], b["features"], rtol=1e-10, atol=1e-8)
    x, y = eng.get_wl(), ora.get_wl()
    assert np.array_equal(x["histogram"], y["histogram"])
    assert np.array_equal(x["occurrences"], y["occurrences"])
    np.testing.assert_allclose(x["entropy"], y["entropy"], rtol=0, atol=0)
    np.testing.assert_allclose(x["mean_features"], y["mean_features"], rtol=1e-10, atol=1e-8)
    np.testing.assert_allclose(x["mod_factor"], y["mod_factor"])
    return a, x


def _same_chain(eng, ora, chunks):
    for n in chunks:
        eng.run(n)
        ora.run(n)
        a, x = _compare(eng, ora)
    return a, x


def _rand_occ(sc, rng, R):
    nsp = np.array([sc.model.prim.nspecies[b] for b in sc.site_b])
    return (rng.random((R, sc.num_sites)) * nsp).astype(np.int32)


def _window(tab, occ, below, above):
    from oracle import oracle as orc

    ev = orc.OracleEvaluator(tab)
    h = np.array([ev.feature_vector(o) @ ev.natural_parameters() for o in occ])
    return float(h.min() - below), float(h.max() + above)


def _mu_two(c):
    """one chemical-potential row per active sublattice (cations, anions)"""
    sc = c["sc"]
    rng = np.random.default_rng(12)
    rows = {b: rng.uniform(-0.4, 0.4, 3) for b in set(sc.site_b)}
    mu = np.zeros((sc.num_sites, 3))
    for s, b in enumerate(sc.site_b):
        mu[s] = rows[b]
    return mu


@pytest.mark.parametrize("phi", ["lds", "hbm"])
@pytest.mark.parametrize("update_period", [1, 3])
@pytest.mark.parametrize("step", [capi.STEP_SWAP, capi.STEP_FLIP], ids=["swap", "flip"])
@pytest.mark.parametrize("mode", ["int", "corr"])  # (corr: three cation species, K = 3 / 4 / 6 functions per orbit -- the KFW instantiations)
def test_two_sublattices_with_ewald(mode, step, update_period, phi, monkeypatch):
    """Disorder on the cation AND the anion sublattice + Ewald term (+ one mu row per sublattice for the
    semigrand flips): narrow bins (many bin changes, the row cache of 32 bins evicts), a check period
    that lets the flatness branch fire, launches of 1 ... 1500 steps."""
    _clean(monkeypatch)
    monkeypatch.setenv("SMOLMC_MULTI_PHI_HBM" if phi == "hbm" else "SMOLMC_MULTI_PHI_LDS", "1")
    name = "rocksalt333_two_sublattices"
    c = load_case(name)
    tab = tables_for(name, MODES[mode], mu_table=_mu_two(c) if step == capi.STEP_FLIP else None)
    R = 6
    occ = _rand_occ(c["sc"], np.random.default_rng(21), R)
    lo, hi = _window(tab, occ, 3.371, 2.193)
    cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, step, min_enthalpy=lo, max_enthalpy=hi, bin_size=0.0517,
                           check_period=50, update_period=update_period, flatness=0.2)
    eng, ora = _pair(tab, cfg, occ, np.arange(R, dtype=np.uint64) + np.uint64(300))
    info = eng.kernel_info()
    assert info.startswith("lean-multi") and ("wl=multi" in info) and (("wl=multi-mean" in info) == (update_period != 1)), info
    assert f"field={1 if phi == 'lds' else 2}" in info, info
    assert ("kf=1" in info) == (mode == "corr"), info
    a, x = _same_chain(eng, ora, (1, 2, 13, 64, 400, 1500))
    assert a["n_accepted"].sum() > 200
    assert (x["occurrences"] > 0).sum(axis=1).max() > 32  # more visited bins than cached rows
    np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=1e-9, atol=1e-7)
    eng.close()


@pytest.mark.parametrize("name,step,mukind,mode", [
    ("fcc_prim666_triplets", capi.STEP_SWAP, None, "int"),
    ("fcc_prim666_triplets", capi.STEP_FLIP, "mu", "corr"),
    ("rocksalt444_ewald", capi.STEP_FLIP, "mu", "int"),
    ("rocksalt333_vacancy_ewald", capi.STEP_SWAP, None, "int"),
])
@pytest.mark.parametrize("variant", ["update3", "running-mean"])
def test_one_class_with_update_period_or_running_means(name, step, mukind, mode, variant, monkeypatch):
    """One site class: mc_wl_kernel takes update_period 1 with per-bin feature SUMS; update_period > 1 (the
    running-mean recurrence of wanglandau.py:235-239 with occurrences that move every third step) and
    SMOLMC_WL_RUNNING_MEAN (the same recurrence at update_period 1) take the multi-class kernel."""
    _clean(monkeypatch)
    if variant == "running-mean":
        monkeypatch.setenv("SMOLMC_WL_RUNNING_MEAN", "1")
    c = load_case(name)
    mu = None
    if mukind:
        mu = np.zeros((c["sc"].num_sites, 3))
        mu[:, :] = np.random.default_rng(5).uniform(-0.3, 0.3, 3)[None, :]
    tab = tables_for(name, MODES[mode], mu_table=mu)
    R = 5
    occ = _rand_occ(c["sc"], np.random.default_rng(31), R)
    lo, hi = _window(tab, occ, 4.77, 3.11)
    cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, step, min_enthalpy=lo, max_enthalpy=hi, bin_size=0.113,
                           check_period=64, update_period=3 if variant == "update3" else 1, flatness=0.3)
    eng, ora = _pair(tab, cfg, occ, np.arange(R, dtype=np.uint64) + np.uint64(41))
    info = eng.kernel_info()
    assert info.startswith("lean-multi") and "wl=multi-mean" in info, info
    a, _ = _same_chain(eng, ora, (1, 17, 63, 700, 1300))
    assert a["n_accepted"].sum() > 100
    eng.close()


def test_general_kernel_still_serves_wang_landau_when_asked(monkeypatch):
    """SMOLMC_NO_WL_MULTI: the round-4 path (mc_kernel) of the same models, same chain."""
    _clean(monkeypatch)
    monkeypatch.setenv("SMOLMC_NO_WL_MULTI", "1")
    tab, cfg, occ0, _ = build("B_wlup3", n_replicas=3)
    eng, ora = _pair(tab, cfg, np.tile(occ0, (3, 1)), [4, 5, 6])
    assert eng.kernel_info().startswith("general"), eng.kernel_info()
    _same_chain(eng, ora, (1, 59, 61, 500))
    eng.close()


@pytest.mark.parametrize("update_period", [1, 3])
def test_device_side_sampling_and_chunked_launches(update_period, monkeypatch):
    """run_sampled on the Wang-Landau multi kernel = the oracle's state every thin_by steps; launches split
    at 37 steps (SMOLMC_LAUNCH_CHUNK: the Wang-Landau state crosses launch boundaries through HBM -- entropy,
    step-count deltas folded into histogram / occurrences, cached rows) leave the chain where it was."""
    _clean(monkeypatch)
    name = "rocksalt333_two_sublattices"
    c = load_case(name)
    tab = tables_for(name, MODES["int"])
    R = 4
    occ = _rand_occ(c["sc"], np.random.default_rng(77), R)
    lo, hi = _window(tab, occ, 6.2, 4.4)
    cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, capi.STEP_SWAP, min_enthalpy=lo, max_enthalpy=hi, bin_size=0.21,
                           check_period=40, update_period=update_period, flatness=0.25)
    eng, ora = _pair(tab, cfg, occ, [9, 8, 7, 6])
    smp = eng.run_sampled(7, 23)
    for i in range(7):
        ora.run(23)
        b = ora.get_state()
        assert np.array_equal(smp["occupancy"][i], b["occupancy"])
        assert np.array_equal(smp["accepted"][i], b["accepted"])
        np.testing.assert_allclose(smp["enthalpy"][i], b["enthalpy"], rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(smp["features"][i], b["features"], rtol=1e-10, atol=1e-8)
    _compare(eng, ora)
    monkeypatch.setenv("SMOLMC_LAUNCH_CHUNK", "37")
    eng.run(500)
    ora.run(500)
    _compare(eng, ora)
    assert (eng.get_wl()["mod_factor"] < 1.0).any()  # the flatness branch fired
    eng.close()


def test_linio2_reference_model_under_wang_landau(monkeypatch):
    """The model the reference ships (docs/src/notebooks/data/basic_ce_ewald.mson: Li+/vacancy and
    Ni3+/Ni4+ disorder, Ewald term) in a 4x4x4 cell under Wang-Landau, swaps and flips, update_period 1 / 2."""
    from smol_amd import mson

    _clean(monkeypatch)
    ce = mson.load_mson(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lno_ce_ewald.mson.json.gz"))
    tab_of = {m: ce.tables(np.diag([4, 4, 4]), feature_mode=m) for m in MODES.values()}
    tab = tab_of[capi.FEATURES_INTERACTIONS]
    cell = tab.supercell
    P, R = cell.size, 5
    rng = np.random.default_rng(3)
    occ = np.ones((R, cell.num_sites), dtype=np.int32)
    occ[:, 2 * P:] = 0
    for r in range(R):
        occ[r, rng.permutation(P)[:P // 2]] = 0
        occ[r, P + rng.permutation(P)[:P // 2]] = 0
    lo, hi = _window(tab, occ, 30.3, 20.7)
    # (binary site spaces: one correlation function per orbit, the correlation features run on the same kernel)
    for step, upd, mode in ((capi.STEP_SWAP, 1, "int"), (capi.STEP_FLIP, 2, "int"), (capi.STEP_SWAP, 2, "corr"), (capi.STEP_FLIP, 1, "corr")):
        tab = tab_of[MODES[mode]]
        cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, step, min_enthalpy=lo, max_enthalpy=hi, bin_size=0.5,
                               check_period=100, update_period=upd, flatness=0.3)
        eng, ora = _pair(tab, cfg, occ, np.arange(R, dtype=np.uint64) + np.uint64(11))
        assert eng.kernel_info().startswith("lean-multi") and "wl=multi" in eng.kernel_info(), eng.kernel_info()
        a, _ = _same_chain(eng, ora, (1, 30, 800))
        assert a["n_accepted"].sum() > 100
        eng.close()
