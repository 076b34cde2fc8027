"""GPU parity: the HIP engine (through the C-ABI) against the CPU oracle and the
committed reference-core fixtures.  Tolerances: accept masks / occupancies bit-exact;
float64 1e-10 relative (BASELINE.json north_star)."""

import ctypes as C
import os

import numpy as np
import pytest

from smol_amd import capi
from tests.cases import CASES, GOLD, load_case, tables_for

pytestmark = pytest.mark.gpu

T = np.load(os.path.join(GOLD, "trajectories.npz"))
MODES = {"int": capi.FEATURES_INTERACTIONS, "corr": capi.FEATURES_CORRELATIONS}
RTOL, ATOL = 1e-10, 1e-9


def _engine(tab, cfg):
    from smol_amd.engine import Engine

    return Engine(tab, cfg)


def _max_rel(a, b, floor):
    """largest |a - b| / |b| over the entries with |b| > floor (north_star's bar is 1e-10 RELATIVE;
    assert_allclose's atol would let a relative 1e-8 through on a value of 0.1)"""
    a, b = np.asarray(a, float), np.asarray(b, float)
    m = np.abs(b) > floor
    return float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m]))) if m.any() else 0.0


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("mode", ["int", "corr"])
def test_eval_full_and_delta_vs_reference_fixtures(name, mode, record_property):
    c = load_case(name)
    g = c["gold"]
    tab = tables_for(name, MODES[mode])
    eng = _engine(tab, capi.make_config(1))
    nce = c["model"].num_corr_functions if mode == "corr" else c["model"].num_orbits
    full = eng.eval_full(g["occ"])
    np.testing.assert_allclose(full[:, :nce], g["full_corr" if mode == "corr" else "full_int"],
                               rtol=RTOL, atol=ATOL)
    if c["ewald"] is not None:
        np.testing.assert_allclose(full[:, nce], g["full_ewald"], rtol=RTOL)
    nper = len(g["flips"]) // len(g["occ"])
    worst = dict(full=0.0, delta=0.0, delta_ewald=0.0)
    for k, occ in enumerate(g["occ"]):
        rows = g["flips"][k * nper:(k + 1) * nper]
        d = eng.eval_delta(occ, rows, single_step=False)  # (rows of (site, code): one single-flip step each)
        np.testing.assert_allclose(
            d[:, :nce], g["delta_corr" if mode == "corr" else "delta_int"][k * nper:(k + 1) * nper],
            rtol=RTOL, atol=ATOL)
        if c["ewald"] is not None:
            np.testing.assert_allclose(d[:, nce], g["delta_ewald"][k * nper:(k + 1) * nper],
                                       rtol=RTOL, atol=1e-8)
        # the relative bar itself, wherever the reference value is not (nearly) zero
        gd = g["delta_corr" if mode == "corr" else "delta_int"][k * nper:(k + 1) * nper]
        worst["delta"] = max(worst["delta"], _max_rel(d[:, :nce], gd, 1e-6))
        if c["ewald"] is not None:
            ge = g["delta_ewald"][k * nper:(k + 1) * nper]
            # (an Ewald delta is a sum over N sites of terms of order max|delta|: cancellation-free entries only)
            worst["delta_ewald"] = max(worst["delta_ewald"], _max_rel(d[:, nce], ge, 1e-3 * float(np.abs(ge).max())))
    worst["full"] = _max_rel(full[:, :nce], g["full_corr" if mode == "corr" else "full_int"], 1e-6)
    print(f"max relative errors vs the reference core [{name}, {mode}]: " +
          ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    for k, v in worst.items():
        record_property(f"max_rel_{k}", v)
        assert v < 1e-10, (k, v)


REPLAY_KERNELS = pytest.mark.parametrize("replay_kernel", ["auto", "general-kernel"])


def _replay_env(monkeypatch, replay_kernel):
    """"auto": a lean handle replays on its own kernel (REPLAY instantiations of mc_lean_kernel /
    mc_wl_kernel -- the kernels the BASELINE configurations run); "general-kernel": mc_kernel."""
    monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    if replay_kernel == "general-kernel":
        monkeypatch.setenv("SMOLMC_REPLAY_GENERAL", "1")
    else:
        monkeypatch.delenv("SMOLMC_REPLAY_GENERAL", raising=False)


def _check_replay(eng, key, R=1):
    steps = np.tile(T[f"{key}_steps"][None], (R, 1, 1))
    us = np.tile(T[f"{key}_u"][None], (R, 1))
    h_start = eng.get_state(occupancy=False)["enthalpy"].copy()
    acc, H = eng.replay(steps, us)
    st = eng.get_state()
    for r in range(R):
        assert np.array_equal(acc[r], T[f"{key}_accepted"])
        np.testing.assert_allclose(H[r], T[f"{key}_H"], rtol=RTOL, atol=ATOL)
        if f"{key}_dH" in T.files:
            # per-step enthalpy change of every ACCEPTED step (the running enthalpy moved by it),
            # purely relative: north_star's 1e-10 where the change is not itself rounding noise
            dH = np.diff(np.concatenate(([h_start[r]], H[r])))
            want = T[f"{key}_dH"]
            sel = T[f"{key}_accepted"].astype(bool) & (np.abs(want) > 1e-6)
            assert sel.sum() > 10
            rel = np.abs(dH[sel] - want[sel]) / np.abs(want[sel])
            assert rel.max() < 1e-10, (key, rel.max())
        assert np.array_equal(st["occupancy"][r], T[f"{key}_occ_final"])
        np.testing.assert_allclose(st["features"][r], T[f"{key}_feat_final"], rtol=RTOL, atol=1e-8)
        assert st["n_accepted"][r] == T[f"{key}_accepted"].sum()
    return st


@REPLAY_KERNELS
@pytest.mark.parametrize("mode", ["int", "corr"])
def test_replay_metropolis_swap_vs_reference_trajectory(mode, replay_kernel, monkeypatch):
    _replay_env(monkeypatch, replay_kernel)
    tab = tables_for("fcc_prim666_triplets", MODES[mode])
    R = 5
    eng = _engine(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP))
    assert eng.kernel_info().startswith("lean")
    eng.set_state(np.tile(T["B_occ0"], (R, 1)), temperature=T["B_T"][0])
    _check_replay(eng, f"B_swap_{mode}", R)


@REPLAY_KERNELS
@pytest.mark.parametrize("mode", ["int", "corr"])
def test_replay_semigrand_flip_ewald_mu(mode, replay_kernel, monkeypatch):
    _replay_env(monkeypatch, replay_kernel)
    tab = tables_for("rocksalt444_ewald", MODES[mode], mu_table=T["C_mu"])
    eng = _engine(tab, capi.make_config(2, capi.KERNEL_METROPOLIS, capi.STEP_FLIP))
    assert eng.kernel_info().startswith("lean")
    eng.set_state(np.tile(T["C_occ0"], (2, 1)), temperature=T["C_T"][0])
    _check_replay(eng, f"C_flip_{mode}", 2)


@REPLAY_KERNELS
def test_replay_two_sublattices(replay_kernel, monkeypatch):
    _replay_env(monkeypatch, replay_kernel)
    tab = tables_for("rocksalt333_two_sublattices", MODES["int"])
    eng = _engine(tab, capi.make_config(2, capi.KERNEL_METROPOLIS, capi.STEP_SWAP))
    assert eng.kernel_info().startswith("lean-multi")
    eng.set_state(np.tile(T["G_occ0"], (2, 1)), temperature=T["G_T"][0])
    _check_replay(eng, "G_swap_int", 2)
    tab = tables_for("rocksalt333_two_sublattices", MODES["corr"], mu_table=T["G_mu"])
    eng = _engine(tab, capi.make_config(2, capi.KERNEL_METROPOLIS, capi.STEP_FLIP))
    eng.set_state(np.tile(T["G_occ0"], (2, 1)), temperature=T["G_T"][0])
    _check_replay(eng, "G_flip_corr", 2)


def test_empty_swap_steps_on_gpu():
    """No site of another species: empty, 'accepted' steps that change nothing
    (mcusher.py:197-199); with one solute the partner search falls back to the long
    candidate stream."""
    from oracle import oracle as orc

    tab = tables_for("fcc_prim666_triplets", MODES["int"])
    cfg = capi.make_config(3, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
    occ = np.zeros((3, tab.num_sites), dtype=np.int32)
    occ[1, 0] = 1
    occ[2, :3] = 1
    eng, ora = _engine(tab, cfg), orc.OracleMC(tab, cfg)
    eng.set_state(occ, [3, 4, 5], 800.0)
    ora.set_state(occ, [3, 4, 5], 800.0)
    eng.run(400)
    ora.run(400)
    a, b = eng.get_state(), ora.get_state()
    assert np.array_equal(a["occupancy"], b["occupancy"])
    assert np.array_equal(a["n_accepted"], b["n_accepted"])
    assert np.array_equal(a["occupancy"][0], occ[0]) and a["n_accepted"][0] == 400


@REPLAY_KERNELS
def test_replay_swap_ewald(replay_kernel, monkeypatch):
    _replay_env(monkeypatch, replay_kernel)
    tab = tables_for("rocksalt444_ewald", MODES["int"], mu_table=T["C_mu"])
    eng = _engine(tab, capi.make_config(1, capi.KERNEL_METROPOLIS, capi.STEP_SWAP))
    eng.set_state(T["C_occ0"][None], temperature=T["C_T"][0])
    _check_replay(eng, "C_swap_int")


@REPLAY_KERNELS
@pytest.mark.parametrize("tag", ["B_wl", "B_wlflat"])
def test_replay_wang_landau(tag, replay_kernel, monkeypatch):
    _replay_env(monkeypatch, replay_kernel)
    tab = tables_for("fcc_prim666_triplets", MODES["int"])
    w = T[f"{tag}_window"]
    cfg = capi.make_config(3, capi.KERNEL_WANGLANDAU, capi.STEP_SWAP, min_enthalpy=w[0],
                           max_enthalpy=w[1], bin_size=w[2], check_period=int(T[f"{tag}_check"][0]))
    eng = _engine(tab, cfg)
    assert eng.L == len(T[f"{tag}_levels"])
    eng.set_state(np.tile(T["B_occ0"], (3, 1)))
    _check_replay(eng, tag, 3)
    wl = eng.get_wl()
    for r in range(3):
        np.testing.assert_allclose(wl["entropy"][r], T[f"{tag}_entropy"], rtol=0, atol=0)
        assert np.array_equal(wl["histogram"][r], T[f"{tag}_histogram"])
        assert np.array_equal(wl["occurrences"][r], T[f"{tag}_occurrences"])
        np.testing.assert_allclose(wl["mean_features"][r], T[f"{tag}_mean_features"], rtol=1e-10,
                                   atol=1e-9)
    np.testing.assert_allclose(wl["mod_factor"], np.tile(T[f"{tag}_mod_factor"], 3))


CONFIGS = [
    ("fcc_conv444_pairs", "int", capi.STEP_SWAP, None),
    ("fcc_prim666_triplets", "int", capi.STEP_SWAP, None),
    ("fcc_prim666_triplets", "int", capi.STEP_FLIP, "mu2"),
    ("fcc3_indicator_skew", "int", capi.STEP_SWAP, None),
    ("fcc3_indicator_skew", "int", capi.STEP_FLIP, "mu3"),
    ("fcc_prim666_triplets", "corr", capi.STEP_FLIP, "mu2"),
    ("rocksalt444_ewald", "int", capi.STEP_FLIP, "mu3"),
    ("rocksalt444_ewald", "corr", capi.STEP_SWAP, None),
    ("fcc3_indicator_skew", "corr", capi.STEP_SWAP, None),
    ("fcc_prim222_aliased", "int", capi.STEP_FLIP, "mu2"),
    ("rocksalt333_vacancy_ewald", "int", capi.STEP_FLIP, "mu3"),
    ("rocksalt333_vacancy_ewald", "corr", capi.STEP_SWAP, None),
    ("rocksalt333_two_sublattices", "int", capi.STEP_SWAP, None),
    ("rocksalt333_two_sublattices", "corr", capi.STEP_FLIP, "muG"),
    ("rocksalt333_two_sublattices", "int", capi.STEP_FLIP, "muG"),
]


def _mu(kind, c):
    if kind is None:
        return None
    if kind == "muG":
        return T["G_mu"]
    nsp = 2 if kind == "mu2" else 3
    mu = np.zeros((c["sc"].num_sites, nsp))
    act = np.array([c["model"].prim.nspecies[b] for b in c["sc"].site_b]) > 1
    mu[act] = np.linspace(-0.3, 0.4, nsp)[None, :]
    return mu


@pytest.mark.parametrize("general", [False, True], ids=["auto", "general-kernel"])
@pytest.mark.parametrize("name,mode,step,mukind", CONFIGS)
def test_native_stream_matches_oracle(name, mode, step, mukind, general, monkeypatch):
    """Same Philox streams on CPU oracle and GPU: identical trajectories (occupancies,
    accept counts bit-exact; enthalpy/features 1e-10), across chunked run() calls.
    "auto" lets the engine pick the lean kernel where eligible; "general-kernel" forces
    mc_kernel so both code paths are pinned."""
    from oracle import oracle as orc

    if general:
        monkeypatch.setenv("SMOLMC_FORCE_GENERAL", "1")
    else:
        monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)

    c = load_case(name)
    tab = tables_for(name, MODES[mode], mu_table=_mu(mukind, c))
    R = 9
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    rng = np.random.default_rng(5)
    nsp = np.array([c["model"].prim.nspecies[b] for b in c["sc"].site_b])
    occ0 = (rng.random((R, c["sc"].num_sites)) * nsp).astype(np.int32)
    seeds = np.arange(100, 100 + R, dtype=np.uint64) * np.uint64(7919)
    temps = np.linspace(400.0, 4000.0, R)
    eng = _engine(tab, cfg)
    ora = orc.OracleMC(tab, cfg)
    eng.set_state(occ0, seeds, temps)
    ora.set_state(occ0, seeds, temps)
    s0, o0 = eng.get_state(), ora.get_state()
    np.testing.assert_allclose(s0["features"], o0["features"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(s0["enthalpy"], o0["enthalpy"], rtol=RTOL, atol=ATOL)
    for chunk in (1, 7, 16, 33, 500):
        eng.run(chunk)
        ora.run(chunk)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        assert np.array_equal(a["n_steps"], b["n_steps"])
        assert np.array_equal(a["accepted"], b["accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(a["features"], b["features"], rtol=RTOL, atol=1e-8)
    # trace consistency: running features == recomputed features (test_sampler.py:59-84)
    np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=RTOL, atol=1e-8)
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()


@pytest.mark.parametrize("scale", ["1", "3e3", "1e9"], ids=["default-band", "wide-band", "always-exact"])
@pytest.mark.parametrize("step,mukind", [(capi.STEP_SWAP, None), (capi.STEP_FLIP, "mu2")])
def test_fast_accept_pretest_never_changes_a_decision(step, mukind, scale, monkeypatch):
    """The lean kernel decides most steps on a float32 wave sum and falls back to the exact
    float64 rule inside an error band.  Widening the band (test hook) makes the two decision
    paths interleave at different rates; the trajectory must not depend on it and must equal
    the oracle's (metropolis.py:46-48 evaluated in float64) over many steps and temperatures."""
    from oracle import oracle as orc

    monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    monkeypatch.setenv("SMOLMC_FAST_EPS_SCALE", scale)
    name = "fcc_prim666_triplets"
    c = load_case(name)
    tab = tables_for(name, MODES["int"], mu_table=_mu(mukind, c))
    R = 64
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    rng = np.random.default_rng(17)
    occ0 = (rng.random((R, c["sc"].num_sites)) < 0.5).astype(np.int32)
    seeds = np.arange(1, R + 1, dtype=np.uint64) * np.uint64(104729)
    temps = np.geomspace(30.0, 30000.0, R)  # from almost-always-reject to almost-always-accept
    eng = _engine(tab, cfg)
    ora = orc.OracleMC(tab, cfg)
    eng.set_state(occ0, seeds, temps)
    ora.set_state(occ0, seeds, temps)
    eng.run(4000)
    ora.run(4000)
    a, b = eng.get_state(), ora.get_state()
    assert np.array_equal(a["occupancy"], b["occupancy"])
    assert np.array_equal(a["n_accepted"], b["n_accepted"])
    np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(a["features"], b["features"], rtol=RTOL, atol=1e-8)


@pytest.mark.parametrize("general", [False, True], ids=["auto", "general-kernel"])
def test_native_wang_landau_matches_oracle(general, monkeypatch):
    from oracle import oracle as orc

    if general:
        monkeypatch.setenv("SMOLMC_FORCE_GENERAL", "1")
    else:
        monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)

    tab = tables_for("fcc_prim666_triplets", MODES["int"])
    w = T["B_wlflat_window"]
    R = 6
    cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, capi.STEP_SWAP, min_enthalpy=w[0],
                           max_enthalpy=w[1], bin_size=0.5, check_period=100)
    eng, ora = _engine(tab, cfg), orc.OracleMC(tab, cfg)
    occ0 = np.tile(T["B_occ0"], (R, 1))
    seeds = np.arange(1, R + 1, dtype=np.uint64)
    eng.set_state(occ0, seeds)
    ora.set_state(occ0, seeds, 0.0)
    for chunk in (3, 50, 1000):
        eng.run(chunk)
        ora.run(chunk)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=RTOL, atol=ATOL)
        wa, wb = eng.get_wl(), ora.get_wl()
        np.testing.assert_allclose(wa["entropy"], wb["entropy"], rtol=0, atol=0)
        assert np.array_equal(wa["histogram"], wb["histogram"])
        assert np.array_equal(wa["occurrences"], wb["occurrences"])
        np.testing.assert_allclose(wa["mean_features"], wb["mean_features"], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(wa["mod_factor"], wb["mod_factor"])


@pytest.mark.parametrize("scale", ["1", "40", "3000"])
def test_wang_landau_bin_pretest_is_decision_neutral(scale, monkeypatch):
    """The lean Wang-Landau kernel takes the bin of the proposed enthalpy from a float32 wave sum
    and a carried enthalpy with a rigorous error bound, and falls back to the exact float64 path near
    bin edges / window ends and whenever the bound has grown to 0.5 % of a bin (mc_wl.h).  With the
    bound scaled up the two paths interleave at every rate (x3000: every accepted step is followed
    by an exact rebuild); histograms, entropies and occupancies must not notice."""
    from oracle import oracle as orc

    monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    monkeypatch.setenv("SMOLMC_FAST_EPS_SCALE", scale)
    tab = tables_for("fcc_prim666_triplets", MODES["int"])
    c = load_case("fcc_prim666_triplets")
    R = 8
    rng = np.random.default_rng(77)
    occ0 = (rng.random((R, c["sc"].num_sites)) < 0.5).astype(np.int32)
    ev = orc.OracleEvaluator(tab)
    h0 = np.array([ev.feature_vector(o) @ ev.natural_parameters() for o in occ0])
    # a narrow window (walkers bounce off both ends) with 0.11 eV bins: many bin changes per walker
    cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, capi.STEP_SWAP, min_enthalpy=h0.min() - 1.337,
                           max_enthalpy=h0.max() + 1.219, bin_size=0.11, check_period=64)
    eng, ora = _engine(tab, cfg), orc.OracleMC(tab, cfg)
    assert eng.kernel_info().startswith("lean")
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(4242)
    eng.set_state(occ0, seeds)
    ora.set_state(occ0, seeds, 0.0)
    for chunk in (1, 17, 700, 2500):
        eng.run(chunk)
        ora.run(chunk)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(a["features"], b["features"], rtol=RTOL, atol=1e-8)
        wa, wb = eng.get_wl(), ora.get_wl()
        np.testing.assert_allclose(wa["entropy"], wb["entropy"], rtol=0, atol=0)
        assert np.array_equal(wa["histogram"], wb["histogram"])
        assert np.array_equal(wa["occurrences"], wb["occurrences"])
        np.testing.assert_allclose(wa["mean_features"], wb["mean_features"], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(wa["mod_factor"], wb["mod_factor"])
    assert wa["histogram"].sum() > 0 and (wa["occurrences"] > 0).sum(axis=1).min() > 3


def test_wang_landau_group_rotation_runs_the_same_chains(monkeypatch):
    """More walkers than mc_wl_kernel keeps resident (4096 against 3072 on 256 CUs): the launch is split into
    sub-launches that each fill the chip with a rotating subset of walker groups (mc_wl.h, launch_wl_kern).  Every
    walker still takes exactly nsteps steps of its own chain: equal to one plain launch (SMOLMC_NO_ROTATE) in
    occupancies, counters, histograms, occurrences and entropies bit for bit, and to the oracle on a few walkers."""
    from oracle import oracle as orc

    monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    monkeypatch.delenv("SMOLMC_NO_ROTATE", raising=False)
    tab = tables_for("fcc_prim666_triplets", MODES["int"])
    c = load_case("fcc_prim666_triplets")
    R = 4096
    rng = np.random.default_rng(99)
    occ0 = (rng.random((R, c["sc"].num_sites)) < 0.5).astype(np.int32)
    ev = orc.OracleEvaluator(tab)
    h0 = np.array([ev.feature_vector(o) @ ev.natural_parameters() for o in occ0])  # (every start lies inside the window)
    kw = dict(min_enthalpy=h0.min() - 6.37, max_enthalpy=h0.max() + 6.11, bin_size=0.25, check_period=50, flatness=0.3)
    cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, capi.STEP_SWAP, **kw)
    seeds = np.arange(R, dtype=np.uint64) * np.uint64(977) + np.uint64(5)
    rot = _engine(tab, cfg)
    assert "wl=v3" in rot.kernel_info()
    monkeypatch.setenv("SMOLMC_NO_ROTATE", "1")
    plain = _engine(tab, cfg)
    pick = np.array([0, 1023, 1024, 3071, 3072, 4095])
    ora = orc.OracleMC(tab, capi.make_config(len(pick), capi.KERNEL_WANGLANDAU, capi.STEP_SWAP, **kw))
    rot.set_state(occ0, seeds)
    plain.set_state(occ0, seeds)
    ora.set_state(occ0[pick], seeds[pick], 0.0)
    for n in (700, 65, 1001):  # (1001 = 3 x 333 + 2: the remainder runs as one launch of all walkers)
        monkeypatch.delenv("SMOLMC_NO_ROTATE", raising=False)
        rot.run(n)
        monkeypatch.setenv("SMOLMC_NO_ROTATE", "1")
        plain.run(n)
        ora.run(n)
        a, b, o = rot.get_state(), plain.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"]) and np.array_equal(a["n_accepted"], b["n_accepted"])
        assert np.array_equal(a["n_steps"], b["n_steps"])
        assert np.array_equal(a["occupancy"][pick], o["occupancy"]) and np.array_equal(a["n_accepted"][pick], o["n_accepted"])
        wa, wb, wo = rot.get_wl(), plain.get_wl(), ora.get_wl()
        for k in ("histogram", "occurrences"):
            assert np.array_equal(wa[k], wb[k]) and np.array_equal(wa[k][pick], wo[k])
        np.testing.assert_allclose(wa["entropy"], wb["entropy"], rtol=0, atol=0)
        np.testing.assert_allclose(wa["entropy"][pick], wo["entropy"], rtol=0, atol=0)
        np.testing.assert_allclose(wa["mean_features"], wb["mean_features"], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(wa["mean_features"][pick], wo["mean_features"], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(wa["mod_factor"], wb["mod_factor"])


def test_errors_surface_as_exceptions(monkeypatch):
    tab = tables_for("fcc_prim222_aliased", MODES["int"])
    eng = _engine(tab, capi.make_config(1))
    with pytest.raises(ValueError):
        eng.eval_full(np.zeros((1, 8), dtype=np.float64))
    with pytest.raises(ValueError):
        _engine(tab, capi.make_config(1, capi.KERNEL_WANGLANDAU, min_enthalpy=2.0, max_enthalpy=1.0))
    # a Wang-Landau walker must start inside the window: the bin of the current enthalpy is used
    # unchecked afterwards (the reference raises IndexError above it, wanglandau.py:175-180)
    for lean in (True, False):
        if not lean:
            monkeypatch.setenv("SMOLMC_FORCE_GENERAL", "1")
        wl = _engine(tab, capi.make_config(2, capi.KERNEL_WANGLANDAU, min_enthalpy=1.0e3,
                                           max_enthalpy=1.1e3, bin_size=1.0))
        with pytest.raises(ValueError, match="outside the Wang-Landau window"):
            wl.set_state(np.zeros((2, 8), dtype=np.int32), None, 0.0)


def test_table_and_occupancy_bounds_are_checked_at_the_boundary():
    """SURVEY 5 (bounds on gather indices): the reference's compiled core runs with boundscheck=False
    and reads garbage for a bad index; here every cluster-site / Ewald index is checked once at
    smolmc_create and every occupancy code against the species of its site at set_state / eval."""
    from smol_amd.engine import EngineError

    c = load_case("rocksalt444_ewald")
    N = c["sc"].num_sites
    # (the flattened arrays may be views of the cached case: every corrupted entry is put back)
    for key, pos, bad, msg in (("loc_idx", 7, N + 5, "local table"), ("full_idx", 7, N + 5, "full table"),
                               ("ewald_inds", 3, None, "Ewald index")):
        tab = tables_for("rocksalt444_ewald", MODES["int"])
        arr = tab._keep[key].reshape(-1)
        old = int(arr[pos])
        arr[pos] = tab.struct.ewald_dim if bad is None else bad
        try:
            with pytest.raises((EngineError, ValueError), match=msg):
                _engine(tab, capi.make_config(1))
        finally:
            arr[pos] = old
    tab = tables_for("rocksalt444_ewald", MODES["int"])
    eng = _engine(tab, capi.make_config(1))
    occ = np.zeros((1, N), dtype=np.int32)
    occ[0, 2] = 3  # ternary cation site: codes 0..2
    with pytest.raises(ValueError, match="occupancy code 3 out of range on site 2"):
        eng.eval_full(occ)
    with pytest.raises(ValueError, match="out of range"):
        eng.set_state(occ, None, 500.0)


def test_bounds_debug_build_runs_clean():
    """libsmolmc_hip_bounds.so (make -C smol_amd/csrc bounds): the lean kernels with a trap at every
    gather whose LDS address leaves the walker's occupancy.  A short run of every lean step type
    must finish (a trap would surface as a HIP error) and equal the normal build."""
    import os
    import subprocess
    import sys

    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "smol_amd", "libsmolmc_hip_bounds.so")
    if not os.path.exists(lib):
        pytest.skip("debug build not made (make -C smol_amd/csrc bounds)")
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from smol_amd import capi, workloads\n"
        "from smol_amd.engine import Engine\n"
        "wl = workloads.config2(count=64, dim=6)\n"
        "out = []\n"
        "for step in (capi.STEP_SWAP, capi.STEP_FLIP):\n"
        "    eng = Engine(wl.tables, capi.make_config(64, capi.KERNEL_METROPOLIS, step))\n"
        "    eng.set_state(wl.occupancy, wl.seeds, 1500.0)\n"
        "    eng.run(700, sync=True)\n"
        "    st = eng.get_state()\n"
        "    out.append(int(st['occupancy'].astype(np.int64).sum() * 7 + st['n_accepted'].sum()))\n"
        "print('CHECK', out)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    res = {}
    for tag, env in (("normal", {}), ("bounds", {"SMOLMC_LIB": lib})):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True,
                           timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[tag] = [l for l in p.stdout.splitlines() if l.startswith("CHECK")][-1]
    assert res["normal"] == res["bounds"]


@pytest.mark.parametrize("kernel", ["metropolis", "wang-landau"])
@pytest.mark.parametrize("offset,thin", [(0, 1), (5, 3), (63, 16), (15, 17), (1, 64), (40, 65)])
def test_sample_rows_at_any_phase_of_the_random_batches(offset, thin, kernel, monkeypatch):
    """The lean step loop runs in chunks that end at the next 16-step random batch, the next sample
    row or the end of the launch: sample rows with periods below, at and above the batch lengths
    (16 and 64 steps), started at arbitrary stream positions, equal step-wise runs and the oracle."""
    from oracle import oracle as orc

    monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    c = load_case("fcc_prim666_triplets")
    tab = tables_for("fcc_prim666_triplets", MODES["int"])
    R = 3
    rng = np.random.default_rng(offset + thin)
    occ0 = (rng.random((R, c["sc"].num_sites)) < 0.5).astype(np.int32)
    if kernel == "metropolis":
        cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
    else:
        # the window must contain every walker's starting enthalpy (set_state refuses otherwise)
        ev = orc.OracleEvaluator(tab)
        h0 = np.array([ev.feature_vector(o) @ ev.natural_parameters() for o in occ0])
        cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, capi.STEP_SWAP, min_enthalpy=h0.min() - 40.3,
                               max_enthalpy=h0.max() + 40.7, bin_size=0.5, check_period=37)
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(7000 + thin)
    eng, ora = _engine(tab, cfg), orc.OracleMC(tab, cfg)
    assert eng.kernel_info().startswith("lean")
    for e in (eng, ora):
        e.set_state(occ0, seeds, 1800.0)
        if offset:
            e.run(offset)
    ns = 5
    smp = eng.run_sampled(ns, thin, occupancy=True)
    for i in range(ns):
        ora.run(thin)
        st = ora.get_state()
        assert np.array_equal(smp["occupancy"][i], st["occupancy"])
        assert np.array_equal(smp["accepted"][i], st["accepted"])
        np.testing.assert_allclose(smp["enthalpy"][i], st["enthalpy"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(smp["features"][i], st["features"], rtol=RTOL, atol=1e-8)
    a, b = eng.get_state(), ora.get_state()
    assert np.array_equal(a["n_steps"], b["n_steps"]) and np.array_equal(a["n_accepted"], b["n_accepted"])


@pytest.mark.parametrize("name", ["fcc_prim666_triplets", "fcc3_indicator_skew", "rocksalt333_vacancy_ewald"])
def test_packed_sample_download_equals_int32_download(name):
    """smolmc_get_samples_u8 (occupancies as the ring's bytes; row pitch Npad on the device, N on
    the host -- 216 / 27 / 54 sites are not multiples of the padding) == smolmc_get_samples."""
    c = load_case(name)
    tab = tables_for(name, MODES["int"])
    R = 5
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_FLIP)
    nsp = np.array([c["model"].prim.nspecies[b] for b in c["sc"].site_b])
    occ0 = (np.random.default_rng(3).random((R, c["sc"].num_sites)) * nsp).astype(np.int32)
    out = []
    for packed in (False, True):
        eng = _engine(tab, cfg)
        eng.set_state(occ0, np.arange(R, dtype=np.uint64) + np.uint64(3), 2500.0)
        out.append(eng.run_sampled(7, 13, occupancy=True, packed=packed))
        if packed:  # the int32 entry point still serves the same ring
            again = np.zeros((7, R, eng.N), dtype=np.int32)
            eng._chk(eng._lib.smolmc_get_samples(eng._h, None, None, None,
                                                 again.ctypes.data_as(C.POINTER(C.c_int32))))
            assert np.array_equal(again, out[0]["occupancy"])
    assert out[0]["occupancy"].dtype == np.int32 and out[1]["occupancy"].dtype == np.uint8
    assert out[1]["occupancy"].shape == (7, R, c["sc"].num_sites)
    assert np.array_equal(out[0]["occupancy"], out[1]["occupancy"])
    assert len(np.unique(out[1]["occupancy"], axis=0)) > 1
    for k in ("enthalpy", "features", "accepted"):
        assert np.array_equal(out[0][k], out[1][k])


@pytest.mark.parametrize("general", [False, True], ids=["auto", "general-kernel"])
@pytest.mark.parametrize("name,mode,step,mukind", [
    ("fcc_prim666_triplets", "int", capi.STEP_SWAP, None),
    ("fcc3_indicator_skew", "int", capi.STEP_FLIP, "mu3"),
    ("rocksalt444_ewald", "corr", capi.STEP_FLIP, "mu3"),
])
def test_device_side_sampling_equals_stepwise(name, mode, step, mukind, general, monkeypatch):
    """smolmc_run_sampled records exactly what repeated run(thin_by) + get_state returns
    (the rows Sampler.sample yields, sampler.py:195-210)."""
    if general:
        monkeypatch.setenv("SMOLMC_FORCE_GENERAL", "1")
    else:
        monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    c = load_case(name)
    tab = tables_for(name, MODES[mode], mu_table=_mu(mukind, c))
    R = 6
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    rng = np.random.default_rng(11)
    nsp = np.array([c["model"].prim.nspecies[b] for b in c["sc"].site_b])
    occ0 = (rng.random((R, c["sc"].num_sites)) * nsp).astype(np.int32)
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(42)
    a, b = _engine(tab, cfg), _engine(tab, cfg)
    a.set_state(occ0, seeds, 1500.0)
    b.set_state(occ0, seeds, 1500.0)
    a.run(5)
    b.run(5)  # samples must continue from a non-trivial state / stream position
    ns, thin = 7, 23
    smp = a.run_sampled(ns, thin, occupancy=True)
    for i in range(ns):
        b.run(thin)
        st = b.get_state()
        assert np.array_equal(smp["occupancy"][i], st["occupancy"])
        assert np.array_equal(smp["accepted"][i], st["accepted"])
        np.testing.assert_allclose(smp["enthalpy"][i], st["enthalpy"], rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(smp["features"][i], st["features"], rtol=1e-12, atol=1e-9)
    fa, fb = a.get_state(), b.get_state()
    assert np.array_equal(fa["occupancy"], fb["occupancy"])
    np.testing.assert_allclose(fa["features"], fb["features"], rtol=1e-12, atol=1e-9)
    assert np.array_equal(fa["n_steps"], fb["n_steps"])
    smp2 = a.run_sampled(2, 10, occupancy=False)
    assert smp2["occupancy"] is None and smp2["enthalpy"].shape == (2, R)


@pytest.mark.parametrize("case", ["rocksalt444_ewald", "rocksalt333_vacancy_ewald"])
@pytest.mark.parametrize("step", [capi.STEP_FLIP, capi.STEP_SWAP])
def test_compact_ewald_matches_dense_rows_and_oracle(step, case, monkeypatch):
    """The factorised Ewald delta (site kernel G, enabled when ewald_charges are given and
    the matrix is of product form) against the dense two-row form of ewald.pyx:38-58 and
    the CPU oracle: same accept decisions, enthalpies to 1e-10."""
    from oracle import oracle as orc

    c = load_case(case)  # (the vacancy case: Ewald index -1 entries in the batched dense gather)
    tab = tables_for(case, MODES["int"], mu_table=_mu("mu3", c))
    assert tab.struct.ewald_charges  # charges travel with the synthetic tables
    R = 5
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    rng = np.random.default_rng(21)
    nsp = np.array([c["model"].prim.nspecies[b] for b in c["sc"].site_b])
    occ0 = (rng.random((R, c["sc"].num_sites)) * nsp).astype(np.int32)
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(9)
    compact = _engine(tab, cfg)
    monkeypatch.setenv("SMOLMC_DENSE_EWALD", "1")
    dense = _engine(tab, cfg)
    monkeypatch.delenv("SMOLMC_DENSE_EWALD")
    ora = orc.OracleMC(tab, cfg)
    for e in (compact, dense, ora):
        e.set_state(occ0, seeds, 2500.0)
        e.run(400)
    a, b, o = compact.get_state(), dense.get_state(), ora.get_state()
    for x in (b, o):
        assert np.array_equal(a["occupancy"], x["occupancy"])
        assert np.array_equal(a["n_accepted"], x["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], x["enthalpy"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(a["features"], x["features"], rtol=RTOL, atol=1e-8)
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()


@pytest.mark.parametrize("step", [capi.STEP_FLIP, capi.STEP_SWAP])
def test_ewald_potential_field_matches_row_sums_and_oracle(step, monkeypatch):
    """The lean kernel keeps the Ewald potential phi[j] = sum_k q_k G[j][k] of every walker in
    LDS (O(1) per proposal, one row update per accepted flip; HBM copy between launches).
    Over many launches of uneven length its trajectory must equal the per-step row sums
    (SMOLMC_NO_EWALD_FIELD) and the CPU oracle, and the Ewald feature must not drift from a
    from-scratch evaluation (ewald.pyx:38-58 applied to the final occupancy)."""
    from oracle import oracle as orc

    c = load_case("rocksalt444_ewald")
    tab = tables_for("rocksalt444_ewald", MODES["int"], mu_table=_mu("mu3", c))
    R = 6
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    rng = np.random.default_rng(33)
    nsp = np.array([c["model"].prim.nspecies[b] for b in c["sc"].site_b])
    occ0 = (rng.random((R, c["sc"].num_sites)) * nsp).astype(np.int32)
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(77)
    temps = np.linspace(800.0, 6000.0, R)
    field = _engine(tab, cfg)
    monkeypatch.setenv("SMOLMC_NO_EWALD_FIELD", "1")
    rows = _engine(tab, cfg)
    monkeypatch.delenv("SMOLMC_NO_EWALD_FIELD")
    ora = orc.OracleMC(tab, cfg)
    for e in (field, rows, ora):
        e.set_state(occ0, seeds, temps)
    for chunk in (1, 2, 37, 500, 3, 2000):
        for e in (field, rows, ora):
            e.run(chunk)
        a, b, o = field.get_state(), rows.get_state(), ora.get_state()
        for x in (b, o):
            assert np.array_equal(a["occupancy"], x["occupancy"])
            assert np.array_equal(a["n_accepted"], x["n_accepted"])
            np.testing.assert_allclose(a["enthalpy"], x["enthalpy"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(a["features"], field.eval_full(a["occupancy"]), rtol=RTOL, atol=1e-8)
    # a continued set_state (new occupancies, same streams) rebuilds the field
    field.set_state(a["occupancy"][::-1].copy(), seeds, temps, reset_aux=False)
    ora.set_state(a["occupancy"][::-1].copy(), seeds, temps, reset_aux=False)
    field.run(300)
    ora.run(300)
    assert np.array_equal(field.get_state()["occupancy"], ora.get_state()["occupancy"])


@pytest.mark.parametrize("general", [False, True], ids=["auto", "general-kernel"])
@pytest.mark.parametrize("cutoffs,step", [
    ({2: 6.0, 3: 5.0, 4: 4.2}, capi.STEP_SWAP),   # 183 clusters/site, quadruplets: NSLOT=4, MM=3
    ({2: 7.5, 3: 5.8}, capi.STEP_FLIP),            # 717 clusters/site: NSLOT=16 (general kernel)
])
def test_large_cluster_sets_match_oracle(cutoffs, step, general, monkeypatch):
    """Models with many clusters per site (more lanes-slots per wave, 4-body clusters)."""
    from oracle import oracle as orc
    from smol_amd import synth

    if general:
        monkeypatch.setenv("SMOLMC_FORCE_GENERAL", "1")
    else:
        monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    model = synth.build_cluster_model(synth.fcc_prim(), cutoffs)
    sc = synth.build_supercell(model, [6, 6, 6])
    mu = None
    if step == capi.STEP_FLIP:
        mu = np.tile(np.array([0.05, -0.1]), (sc.num_sites, 1))
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=9, scale=0.005), mu_table=mu)
    R = 5
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    rng = np.random.default_rng(31)
    occ0 = (rng.random((R, sc.num_sites)) < 0.5).astype(np.int32)
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(77)
    eng, ora = _engine(tab, cfg), orc.OracleMC(tab, cfg)
    eng.set_state(occ0, seeds, 1800.0)
    ora.set_state(occ0, seeds, 1800.0)
    np.testing.assert_allclose(eng.get_state()["features"], ora.get_state()["features"], rtol=RTOL, atol=1e-8)
    eng.run(300)
    ora.run(300)
    a, b = eng.get_state(), ora.get_state()
    assert np.array_equal(a["occupancy"], b["occupancy"])
    assert np.array_equal(a["n_accepted"], b["n_accepted"])
    np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(a["features"], b["features"], rtol=RTOL, atol=1e-8)
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()


def test_model_loaded_from_npz_runs_identically(tmp_path):
    """A model that went through the wire format (smol_amd.io) drives the engine to the same
    trajectory as the in-memory tables."""
    from smol_amd import io

    tab = tables_for("rocksalt444_ewald", MODES["int"], mu_table=T["C_mu"])
    path = str(tmp_path / "m.npz")
    io.save_tables(path, tab)
    tab2 = io.load_tables(path)
    cfg = capi.make_config(3, capi.KERNEL_METROPOLIS, capi.STEP_FLIP)
    a, b = _engine(tab, cfg), _engine(tab2, cfg)
    occ0 = np.tile(T["C_occ0"], (3, 1))
    for e in (a, b):
        e.set_state(occ0, [1, 2, 3], 2000.0)
        e.run(250)
    sa, sb = a.get_state(), b.get_state()
    assert np.array_equal(sa["occupancy"], sb["occupancy"])
    np.testing.assert_array_equal(sa["enthalpy"], sb["enthalpy"])


def test_more_than_65535_sites_uses_32bit_rows():
    """N > 65535: index rows no longer fit 16 bits (general kernel with int32 rows)."""
    from oracle import oracle as orc
    from smol_amd import synth

    model = synth.build_cluster_model(synth.fcc_prim(), {2: 4.5})
    sc = synth.build_supercell(model, [41, 41, 41])  # 68921 sites
    assert sc.num_sites > 65535
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=2))
    R = 3
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
    rng = np.random.default_rng(3)
    occ0 = (rng.random((R, sc.num_sites)) < 0.5).astype(np.int32)
    eng, ora = _engine(tab, cfg), orc.OracleMC(tab, cfg)
    eng.set_state(occ0, [7, 8, 9], 1500.0)
    ora.set_state(occ0, [7, 8, 9], 1500.0)
    np.testing.assert_allclose(eng.get_state()["features"], ora.get_state()["features"], rtol=RTOL)
    eng.run(500)
    ora.run(500)
    a, b = eng.get_state(), ora.get_state()
    assert np.array_equal(a["occupancy"], b["occupancy"])
    np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=RTOL, atol=1e-8)


@pytest.mark.parametrize("name,mode,step,mukind,kernel,expected", [
    ("fcc_conv444_pairs", "int", capi.STEP_SWAP, None, "metropolis", "lean"),
    ("fcc_prim666_triplets", "int", capi.STEP_SWAP, None, "metropolis", "lean"),
    ("fcc_prim666_triplets", "int", capi.STEP_FLIP, "mu2", "metropolis", "lean"),
    ("fcc_prim666_triplets", "int", capi.STEP_SWAP, None, "wang-landau", "lean"),
    ("rocksalt444_ewald", "int", capi.STEP_FLIP, "mu3", "metropolis", "lean"),        # compact Ewald + field
    ("rocksalt333_vacancy_ewald", "int", capi.STEP_FLIP, "mu3", "metropolis", "lean"),
    ("fcc_prim666_triplets", "corr", capi.STEP_SWAP, None, "metropolis", "lean"),     # correlation features, K = 1
    ("fcc_prim666_triplets", "corr", capi.STEP_SWAP, None, "wang-landau", "lean"),
    ("fcc_conv444_pairs", "corr", capi.STEP_FLIP, "mu2", "metropolis", "lean"),
    ("rocksalt444_ewald", "corr", capi.STEP_FLIP, "mu3", "metropolis", "lean"),       # K = 3 / 4 / 6 functions per orbit
    ("rocksalt444_ewald", "corr", capi.STEP_SWAP, None, "metropolis", "lean"),
    ("fcc3_indicator_skew", "corr", capi.STEP_FLIP, "mu3", "metropolis", "lean"),
    ("rocksalt444_ewald", "corr", capi.STEP_SWAP, None, "wang-landau", "lean-multi"), # WL with K > 1: the KFW kernel (round 5)
    ("rocksalt333_two_sublattices", "int", capi.STEP_SWAP, None, "wang-landau", "lean-multi"),   # WL + two sublattices (round 5)
    ("rocksalt333_two_sublattices", "int", capi.STEP_FLIP, "muG", "wang-landau", "lean-multi"),
    ("rocksalt333_two_sublattices", "corr", capi.STEP_FLIP, "muG", "wang-landau", "lean-multi"),  # K > 1 of a multi-class model
    ("fcc_prim222_aliased", "int", capi.STEP_FLIP, "mu2", "metropolis", "lean"),      # aliased cell: the site's own positions folded into the slot's table (round 6)
    ("fcc_prim222_aliased", "corr", capi.STEP_SWAP, None, "metropolis", "lean"),
    ("fcc_prim222_aliased", "int", capi.STEP_SWAP, None, "wang-landau", "lean"),
    ("rocksalt333_two_sublattices", "int", capi.STEP_SWAP, None, "metropolis", "lean-multi"),
    ("rocksalt333_two_sublattices", "int", capi.STEP_FLIP, "muG", "metropolis", "lean-multi"),
    ("rocksalt444_ewald", "int", capi.STEP_SWAP, None, "wang-landau", "lean"),        # WL + Ewald: field in LDS (round 4)
    ("rocksalt444_ewald", "int", capi.STEP_FLIP, "mu3", "wang-landau", "lean"),       # semigrand WL + Ewald
    ("fcc_prim666_triplets", "int", capi.STEP_FLIP, "mu2", "wang-landau", "lean"),    # semigrand WL
])
def test_dispatch_goes_where_the_design_says(name, mode, step, mukind, kernel, expected, monkeypatch):
    """DESIGN.md section 4 dispatch rules, asserted through smolmc_kernel_info: the parity tests
    above must not silently run everything on the general kernel."""
    monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    c = load_case(name)
    tab = tables_for(name, MODES[mode], mu_table=_mu(mukind, c))
    if kernel == "metropolis":
        cfg = capi.make_config(3, capi.KERNEL_METROPOLIS, step)
    else:
        cfg = capi.make_config(3, capi.KERNEL_WANGLANDAU, step, min_enthalpy=-50.0, max_enthalpy=50.0, bin_size=0.5)
    info = _engine(tab, cfg).kernel_info()
    assert info.startswith(expected), info
    if name == "rocksalt444_ewald" and kernel == "metropolis":
        assert "field=1" in info
    if mode == "corr" and expected == "lean-multi" and kernel == "wang-landau":
        assert "kf=1" in info, info
    if mode == "corr" and expected == "lean":
        assert ("kf=1" in info) == (name not in ("fcc_prim666_triplets", "fcc_conv444_pairs", "fcc_prim222_aliased"))  # (binary: K = 1)


@pytest.mark.parametrize("name,step,mukind", [("fcc_prim666_triplets", capi.STEP_SWAP, None),
                                              ("fcc3_indicator_skew", capi.STEP_FLIP, "mu3"),
                                              ("fcc_conv444_pairs", capi.STEP_SWAP, None)])
def test_one_wave_per_workgroup_layout_equals_shared_layout(name, step, mukind, monkeypatch):
    """mc_lean_kernel in its SOLO layout (occupancy at LDS address 0, 32-bit index rows, private
    tables; chosen for Metropolis models without Ewald term) against the four-waves-per-workgroup
    layout (SMOLMC_NO_SOLO) and the oracle: same occupancies, counters, samples."""
    from oracle import oracle as orc

    monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    monkeypatch.delenv("SMOLMC_NO_SOLO", raising=False)
    c = load_case(name)
    tab = tables_for(name, MODES["int"], mu_table=_mu(mukind, c))
    R = 9  # (not a multiple of the four waves of the shared layout)
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    rng = np.random.default_rng(77)
    nsp = np.array([c["model"].prim.nspecies[b] for b in c["sc"].site_b])
    occ0 = (rng.random((R, c["sc"].num_sites)) * nsp).astype(np.int32)
    seeds = np.arange(R, dtype=np.uint64) * np.uint64(31) + np.uint64(9)
    temps = np.linspace(600.0, 4000.0, R)
    solo = _engine(tab, cfg)
    assert "solo=1" in solo.kernel_info(), solo.kernel_info()
    monkeypatch.setenv("SMOLMC_NO_SOLO", "1")
    shared = _engine(tab, cfg)
    assert shared.kernel_info().startswith("lean") and "solo" not in shared.kernel_info()
    ora = orc.OracleMC(tab, cfg)
    for e in (solo, shared, ora):
        e.set_state(occ0, seeds, temps)
    for chunk in (1, 15, 16, 64, 65, 300):
        for e in (solo, shared, ora):
            e.run(chunk)
        a = solo.get_state()
        for x in (shared.get_state(), ora.get_state()):
            assert np.array_equal(a["occupancy"], x["occupancy"])
            assert np.array_equal(a["n_accepted"], x["n_accepted"])
            assert np.array_equal(a["accepted"], x["accepted"])
            np.testing.assert_allclose(a["enthalpy"], x["enthalpy"], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(a["features"], x["features"], rtol=RTOL, atol=1e-8)
    sa, sb = solo.run_sampled(5, 21, occupancy=True), shared.run_sampled(5, 21, occupancy=True)
    assert np.array_equal(sa["occupancy"], sb["occupancy"]) and np.array_equal(sa["accepted"], sb["accepted"])
    np.testing.assert_allclose(sa["enthalpy"], sb["enthalpy"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("name,step,mukind", [("fcc_prim666_triplets", capi.STEP_SWAP, None),
                                              ("fcc3_indicator_skew", capi.STEP_FLIP, "mu3")])
def test_six_waves_per_simd_instantiation_equals_default_and_oracle(name, step, mukind, monkeypatch):
    """More walkers than 4 waves per SIMD keep resident (16 x CU count) select the SOLO
    instantiation whose registers are held to 6 waves per SIMD (kernel_info "occ=6"): same
    trajectories as the default instantiation (SMOLMC_NO_OCC6) and the oracle."""
    import torch
    from oracle import oracle as orc

    for v in ("SMOLMC_FORCE_GENERAL", "SMOLMC_NO_SOLO", "SMOLMC_NO_OCC6"):
        monkeypatch.delenv(v, raising=False)
    c = load_case(name)
    tab = tables_for(name, MODES["int"], mu_table=_mu(mukind, c))
    R = 16 * torch.cuda.get_device_properties(0).multi_processor_count + 37
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    rng = np.random.default_rng(78)
    nsp = np.array([c["model"].prim.nspecies[b] for b in c["sc"].site_b])
    occ0 = (rng.random((R, c["sc"].num_sites)) * nsp).astype(np.int32)
    seeds = np.arange(R, dtype=np.uint64) * np.uint64(13) + np.uint64(5)
    temps = np.linspace(500.0, 5000.0, R)
    six = _engine(tab, cfg)
    assert "solo=1 occ=6" in six.kernel_info(), six.kernel_info()
    monkeypatch.setenv("SMOLMC_NO_OCC6", "1")
    four = _engine(tab, cfg)
    assert "solo=1" in four.kernel_info() and "occ" not in four.kernel_info()
    ora = orc.OracleMC(tab, cfg)
    for e in (six, four, ora):
        e.set_state(occ0, seeds, temps)
    for chunk in (1, 63, 200):
        for e in (six, four, ora):
            e.run(chunk)
        a = six.get_state()
        for x in (four.get_state(), ora.get_state()):
            assert np.array_equal(a["occupancy"], x["occupancy"])
            assert np.array_equal(a["n_accepted"], x["n_accepted"])
            np.testing.assert_allclose(a["enthalpy"], x["enthalpy"], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(a["features"], x["features"], rtol=RTOL, atol=1e-8)


@pytest.mark.parametrize("kernel", ["metropolis", "wang-landau"])
def test_split_launches_equal_one_launch(kernel, monkeypatch):
    """The lean kernels count steps in 32 bits, so the host splits long runs into several
    launches (on sample boundaries).  With the split forced to tiny chunks the samples and the
    final state must equal those of a single launch."""
    monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    c = load_case("fcc_prim666_triplets")
    tab = tables_for("fcc_prim666_triplets", MODES["int"])
    R = 5
    rng = np.random.default_rng(5)
    occ0 = (rng.random((R, c["sc"].num_sites)) < 0.5).astype(np.int32)
    if kernel == "metropolis":
        cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
    else:
        probe = _engine(tab, capi.make_config(1))
        h0 = probe.eval_full(occ0) @ probe.natural_parameters
        cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, capi.STEP_SWAP, min_enthalpy=h0.min() - 40.3,
                               max_enthalpy=h0.max() + 40.7, bin_size=0.5, check_period=40)
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(500)
    one = _engine(tab, cfg)
    one.set_state(occ0, seeds, 1500.0)
    ref = one.run_sampled(12, 25, occupancy=True)
    ref_state = one.get_state()
    monkeypatch.setenv("SMOLMC_LAUNCH_CHUNK", "60")  # -> chunks of 50 steps = 2 samples
    many = _engine(tab, cfg)
    assert many.kernel_info().startswith("lean")
    many.set_state(occ0, seeds, 1500.0)
    got = many.run_sampled(12, 25, occupancy=True)
    st = many.get_state()
    assert np.array_equal(got["occupancy"], ref["occupancy"])
    assert np.array_equal(got["accepted"], ref["accepted"])
    np.testing.assert_allclose(got["enthalpy"], ref["enthalpy"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(got["features"], ref["features"], rtol=RTOL, atol=1e-8)
    assert np.array_equal(st["occupancy"], ref_state["occupancy"])
    assert np.array_equal(st["n_steps"], ref_state["n_steps"])
    many.run(130)  # unsampled run, split at 60 steps
    one.run(130)
    assert np.array_equal(many.get_state()["occupancy"], one.get_state()["occupancy"])
    if kernel == "wang-landau":
        wa, wb = many.get_wl(), one.get_wl()
        assert np.array_equal(wa["histogram"], wb["histogram"]) and np.array_equal(wa["occurrences"], wb["occurrences"])
        np.testing.assert_allclose(wa["mean_features"], wb["mean_features"], rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("variant", ["lds-field", "hbm-field", "no-ewald"])
@pytest.mark.parametrize("step,mukind", [(capi.STEP_SWAP, None), (capi.STEP_FLIP, "muG")], ids=["swap", "flip+mu"])
def test_lean_multi_kernel_two_sublattices(step, mukind, variant, monkeypatch):
    """mc_lean_multi_kernel (several active sublattices): trajectories equal the oracle's and the
    general kernel's, with the Ewald potential field in LDS, forced into HBM, and without the
    Ewald term; thinned device-side samples equal step-wise runs."""
    from oracle import oracle as orc

    monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    if variant == "hbm-field":
        monkeypatch.setenv("SMOLMC_MULTI_PHI_HBM", "1")
    name = "rocksalt333_two_sublattices"
    c = load_case(name)
    if variant == "no-ewald":
        tab = capi.TableSet.from_synth(c["sc"], c["coefs"], feature_mode=MODES["int"], mu_table=_mu(mukind, c))
    else:
        tab = tables_for(name, MODES["int"], mu_table=_mu(mukind, c))
    R = 11
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    rng = np.random.default_rng(41)
    nsp = np.array([c["model"].prim.nspecies[b] for b in c["sc"].site_b])
    occ0 = (rng.random((R, c["sc"].num_sites)) * nsp).astype(np.int32)
    seeds = np.arange(R, dtype=np.uint64) * np.uint64(977) + np.uint64(3)
    temps = np.linspace(700.0, 5000.0, R)
    eng = _engine(tab, cfg)
    info = eng.kernel_info()
    assert info.startswith("lean-multi")
    assert ("field=1" in info) == (variant == "lds-field") and ("field=2" in info) == (variant == "hbm-field")
    monkeypatch.setenv("SMOLMC_FORCE_GENERAL", "1")
    gen = _engine(tab, cfg)
    assert gen.kernel_info().startswith("general")
    monkeypatch.delenv("SMOLMC_FORCE_GENERAL")
    ora = orc.OracleMC(tab, cfg)
    for e in (eng, gen, ora):
        e.set_state(occ0, seeds, temps)
    for chunk in (1, 15, 16, 17, 400):
        for e in (eng, gen, ora):
            e.run(chunk)
        a, g, o = eng.get_state(), gen.get_state(), ora.get_state()
        for x in (g, o):
            assert np.array_equal(a["occupancy"], x["occupancy"])
            assert np.array_equal(a["n_accepted"], x["n_accepted"])
            assert np.array_equal(a["accepted"], x["accepted"])
            np.testing.assert_allclose(a["enthalpy"], x["enthalpy"], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(a["features"], x["features"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=RTOL, atol=1e-8)
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()
    smp = eng.run_sampled(4, 30, occupancy=True)
    for i in range(4):
        ora.run(30)
        so = ora.get_state()
        assert np.array_equal(smp["occupancy"][i], so["occupancy"])
        np.testing.assert_allclose(smp["enthalpy"][i], so["enthalpy"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(smp["features"][i], so["features"], rtol=RTOL, atol=1e-8)


@pytest.mark.parametrize("step", [capi.STEP_SWAP, capi.STEP_FLIP])
def test_lean_multi_kernel_many_clusters(step):
    """257-512 clusters per site (one class) take mc_lean_multi_kernel with NSLOT = 8."""
    from oracle import oracle as orc
    from smol_amd import synth

    model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.5, 3: 5.2})  # 451 clusters / site
    sc = synth.build_supercell(model, [6, 6, 6])
    mu = None
    if step == capi.STEP_FLIP:
        mu = np.tile(np.array([0.0, 0.15]), (sc.num_sites, 1))
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=9), mu_table=mu)
    R = 6
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    eng, ora = _engine(tab, cfg), orc.OracleMC(tab, cfg)
    assert eng.kernel_info().startswith("lean-multi nslot=8")
    rng = np.random.default_rng(2)
    occ0 = (rng.random((R, sc.num_sites)) < 0.5).astype(np.int32)
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(60)
    for e in (eng, ora):
        e.set_state(occ0, seeds, 3000.0)
        e.run(37)
        e.run(300)
    a, o = eng.get_state(), ora.get_state()
    assert np.array_equal(a["occupancy"], o["occupancy"])
    np.testing.assert_allclose(a["enthalpy"], o["enthalpy"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(a["features"], o["features"], rtol=RTOL, atol=1e-8)


@pytest.mark.parametrize("what", ["fcc222", "fcc234", "fcc233-ternary", "rocksalt222-ewald", "fcc222-wl", "rocksalt322-two-sublattices",
                                  "fcc223-corr-k3"])
def test_aliased_cells_on_the_lean_kernels(what, monkeypatch):
    """Supercells shorter than their clusters: a cluster row holds a site twice (the reference keeps such rows,
    clusterspace.py:1353-1359, and flips every position of the site at once, evaluator.pyx:258-259).  Round 6: the
    lean families fold the flipped site's own positions into the slot's delta table (engine.hip, `LSlot`); until
    then these cells ran on mc_kernel's GENERIC rows.  Same chains as the oracle: binary / ternary fcc, rocksalt with
    the Ewald term (field in LDS), Wang-Landau, two active sublattices (mc_lean_multi_kernel), several correlation
    functions per orbit (lazy features); SMOLMC_NO_LEAN_ALIASED puts the cell back on mc_kernel: the same chain."""
    from oracle import oracle as orc
    from smol_amd import ewald as ew
    from smol_amd import synth

    monkeypatch.delenv("SMOLMC_NO_LEAN_ALIASED", raising=False)
    monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    kw, kernel, step, mode = {}, capi.KERNEL_METROPOLIS, capi.STEP_SWAP, MODES["int"]
    if what.startswith("fcc"):
        nsp = 3 if "ternary" in what or "k3" in what else 2
        model = synth.build_cluster_model(synth.fcc_prim(nspecies=nsp), {2: 6.0, 3: 5.0})
        dims = {"fcc222": [2, 2, 2], "fcc234": [2, 3, 4], "fcc233": [2, 3, 3], "fcc223": [2, 2, 3]}[what.split("-")[0]]
        sc = synth.build_supercell(model, dims)
        if "ternary" in what:
            step = capi.STEP_FLIP
            mu = np.zeros((sc.num_sites, 3))
            mu[:] = [0.0, 0.05, -0.04]
            kw["mu_table"] = mu
        if "corr" in what:
            mode = MODES["corr"]
    else:
        two = "two" in what
        prim = synth.rocksalt_prim(anion_charges=(-2.0, -1.0)) if two else synth.rocksalt_prim()
        model = synth.build_cluster_model(prim, {2: 6.0, 3: 4.5})
        sc = synth.build_supercell(model, [3, 2, 2] if two else [2, 2, 2])
        if "ewald" in what:
            kw.update(ewald=ew.supercell_ewald(sc), ewald_coef=0.1)
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=11, scale=0.03), feature_mode=mode, **kw)
    R = 6
    rng = np.random.default_rng(5)
    nspc = np.array([sc.model.prim.nspecies[b] for b in sc.site_b])
    occ = (rng.random((R, sc.num_sites)) * nspc).astype(np.int32)
    cfgkw = {}
    if what.endswith("-wl"):
        kernel = capi.KERNEL_WANGLANDAU
        ev = orc.OracleEvaluator(tab)
        h0 = np.array([ev.feature_vector(o) @ ev.natural_parameters() for o in occ])
        cfgkw = dict(min_enthalpy=h0.min() - 3.37, max_enthalpy=h0.max() + 3.11, bin_size=0.25, check_period=40, flatness=0.3)
    cfg = capi.make_config(R, kernel, step, **cfgkw)
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(300)
    temps = np.linspace(1500.0, 6000.0, R)
    eng, ora = _engine(tab, cfg), orc.OracleMC(tab, cfg)
    info = eng.kernel_info()
    assert info.startswith("lean"), info
    if "two" in what:
        assert info.startswith("lean-multi"), info
    if "k3" in what:
        assert "lazy-features" in info or "kf=1" in info, info
    monkeypatch.setenv("SMOLMC_NO_LEAN_ALIASED", "1")
    old = _engine(tab, cfg)
    monkeypatch.delenv("SMOLMC_NO_LEAN_ALIASED")
    assert old.kernel_info().startswith("general") and "aliased" in old.kernel_info(), old.kernel_info()
    for e in (eng, ora, old):
        e.set_state(occ, seeds, temps)
    for n in (1, 17, 64, 300):
        for e in (eng, ora, old):
            e.run(n)
        a, b, c = eng.get_state(), ora.get_state(), old.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"]), (what, n)
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(a["features"], b["features"], rtol=1e-10, atol=1e-8)
        assert np.array_equal(a["occupancy"], c["occupancy"])
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()
    np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=1e-10, atol=1e-8)
    if kernel == capi.KERNEL_WANGLANDAU:
        wa, wb = eng.get_wl(), ora.get_wl()
        assert np.array_equal(wa["histogram"], wb["histogram"]) and np.array_equal(wa["occurrences"], wb["occurrences"])
        np.testing.assert_allclose(wa["entropy"], wb["entropy"], rtol=0, atol=0)
        np.testing.assert_allclose(wa["mean_features"], wb["mean_features"], rtol=1e-10, atol=1e-9)
    # a device-sampled block on the same handle
    if kernel == capi.KERNEL_METROPOLIS:
        s = eng.run_sampled(3, 20, occupancy=True)
        for j in range(3):
            ora.run(20)
            assert np.array_equal(s["occupancy"][j], ora.get_state()["occupancy"])
