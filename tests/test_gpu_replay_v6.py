"""ABI-v6 replay on the GPU: the reference-order trajectories of tests/golden/trajectories_v6.npz
(TableFlip on one and two sublattices, unequal weights, the composition limit, TableFlip + bias,
TableFlip + Wang-Landau, the three MCBias terms, Wang-Landau with update_period 3) replayed through
smolmc_replay on the handle's own kernel, on the universal kernel and -- where mc_kernel takes the
step type -- on the general kernel.  Accept flags, final occupancies, counters: bit-exact;
enthalpies / features / bias / a-priori factors: 1e-10 relative."""

import numpy as np
import pytest

from smol_amd import capi
from tests.v6_cases import SPECS, T6, build, check_replay, check_wl

pytestmark = pytest.mark.gpu
TABLE = [t for t, s in SPECS.items() if s["step"] == "table"]


def _engine(tab, cfg):
    from smol_amd.engine import Engine

    return Engine(tab, cfg)


def _env(monkeypatch, kernel):
    for k in ("SMOLMC_REPLAY_GENERAL", "SMOLMC_REPLAY_UNIVERSAL", "SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL"):
        monkeypatch.delenv(k, raising=False)
    if kernel == "universal":
        monkeypatch.setenv("SMOLMC_REPLAY_UNIVERSAL", "1")
    elif kernel == "general":
        monkeypatch.setenv("SMOLMC_REPLAY_GENERAL", "1")
    elif kernel == "universal-handle":
        monkeypatch.setenv("SMOLMC_FORCE_UNIVERSAL", "1")
    elif kernel == "general-handle":
        monkeypatch.setenv("SMOLMC_FORCE_GENERAL", "1")


def _replay(tag, kernel, monkeypatch, derive):
    _env(monkeypatch, kernel)
    monkeypatch.setenv("SMOLMC_DEBUG", "1")  # the engine names the replay path on stderr
    R = 3
    tab, cfg, occ0, temp = build(tag, n_replicas=R)
    eng = _engine(tab, cfg)
    info = eng.kernel_info()
    eng.set_state(np.tile(occ0, (R, 1)), temperature=temp)
    h0 = float(eng.get_state(occupancy=False)["enthalpy"][0])
    steps = np.tile(T6[f"{tag}_steps"][None], (R, 1, 1))
    us = np.tile(T6[f"{tag}_u"][None], (R, 1))
    lp = None if derive else np.tile(T6[f"{tag}_log_priori"][None], (R, 1))
    acc, H, lpo = eng.replay(steps, us, log_priori=lp, with_priori=True)
    for r in range(R):
        assert np.array_equal(acc[r], acc[0]) and np.array_equal(H[r], H[0])
    check_replay(eng, tag, acc[0], H[0], lp_out=lpo[0] if SPECS[tag]["step"] == "table" else None, h0=h0)
    if "wl" in SPECS[tag]:
        check_wl(eng, tag)
    eng.close()
    return info


# kernel family the handle of a trajectory gets by default, and the replay path that goes with it
LEAN_TABLE = {"TC_tf_int", "TC_tfw_int", "TC_tflim_int", "TG_tf_int", "TG6_tf_int"}
LEAN_PLAIN = {"BC_fug_flip_int", "BG_sqc_swap_int"}  # (lean / lean-multi with an MCBias term)


@pytest.mark.parametrize("kernel", ["auto", "universal", "general", "universal-handle", "general-handle"])
@pytest.mark.parametrize("tag", sorted(SPECS))
def test_reference_trajectories_replay(tag, kernel, monkeypatch, capfd):
    info = _replay(tag, kernel, monkeypatch, derive=SPECS[tag]["step"] == "table")
    err = capfd.readouterr().err
    path = [ln.split("path=")[1].split()[0] for ln in err.splitlines() if "replay path=" in ln][-1]
    sp = SPECS[tag]
    if kernel == "universal-handle" or (kernel == "general-handle" and sp["step"] == "table"):
        assert info.startswith("universal") and path == "universal", (info, path)
    if kernel == "universal":
        assert path == "universal"
    if kernel in ("general", "general-handle") and sp["step"] != "table":
        assert path == "general", (info, path)
    if kernel == "auto" and tag in LEAN_TABLE:
        assert info.startswith("lean") and path == "lean-table", (info, path)  # the TableFlip kernels of config 5
    if kernel == "auto" and tag in LEAN_PLAIN:
        assert info.startswith("lean") and path == "lean", (info, path)
    if kernel == "auto" and tag == "B_wlup3":  # update_period 3: the Wang-Landau variant of the multi-class kernel (round 5)
        assert info.startswith("lean-multi") and "wl=multi-mean" in info and path == "lean", (info, path)


@pytest.mark.parametrize("kernel", ["auto", "universal"])
@pytest.mark.parametrize("tag", TABLE)
def test_table_replay_with_the_reference_priori_factor(tag, kernel, monkeypatch):
    _replay(tag, kernel, monkeypatch, derive=False)


def test_step_outside_the_flip_table_fails_like_the_reference(monkeypatch):
    _env(monkeypatch, "auto")
    tab, cfg, occ0, temp = build("TC_tf_int")
    eng = _engine(tab, cfg)
    eng.set_state(occ0[None], temperature=temp)
    site = int(np.flatnonzero(occ0[:64] == 0)[0])
    with pytest.raises((RuntimeError, ValueError), match="not in flip table"):
        eng.replay(np.array([[[site, 1]]], dtype=np.int32), np.array([[0.5]]))
    anion = 64 + 3
    with pytest.raises((RuntimeError, ValueError), match="not changeable"):
        eng.replay(np.array([[[anion, 0]]], dtype=np.int32), np.array([[0.5]]))


def test_eval_delta_takes_whole_table_steps():
    """smolmc_eval_delta with records of up to eight flips: every recorded TableFlip step of the
    reference trajectory in one call, against the oracle's compute_feature_vector_change."""
    from oracle import oracle as orc

    for tag in ("TG6_tf_int", "TC_tf_corr"):
        tab, cfg, occ0, _ = build(tag)
        eng, ev = _engine(tab, capi.make_config(1)), orc.OracleEvaluator(tab)
        steps = T6[f"{tag}_steps"][:200]
        d = eng.eval_delta(occ0, steps)
        for k, row in enumerate(steps):
            fl = [(int(row[2 * j]), int(row[2 * j + 1])) for j in range(8) if row[2 * j] >= 0]
            want = ev.feature_vector_change(occ0, fl) if fl else np.zeros(eng.F)
            np.testing.assert_allclose(d[k], want, rtol=1e-10, atol=1e-9)
