"""Pin the oracle's TableFlip usher, a-priori factor, MCBias terms and Wang-Landau update_period
against trajectories replayed from the reference (tests/golden/trajectories_v6.npz): proposals in
the reference's PCG64 call order (mcusher.py:553-639), a-priori factors from
compute_log_priori_factor (:656-711, scipy gammaln), bias changes from bias.py:75-93,188-206,
every feature delta from the reference's compiled core."""

import numpy as np
import pytest

from oracle import oracle as orc
from tests.v6_cases import SPECS, T6, build, check_replay, check_wl

TABLE = [t for t, s in SPECS.items() if s["step"] == "table"]


@pytest.mark.parametrize("tag", sorted(SPECS))
def test_replay_with_the_reference_priori_factor(tag):
    tab, cfg, occ0, temp = build(tag)
    mc = orc.OracleMC(tab, cfg)
    mc.set_state(occ0[None], [0], temp)
    h0 = float(mc.get_state()["enthalpy"][0])
    lp = T6[f"{tag}_log_priori"][None]
    acc, H = mc.replay(T6[f"{tag}_steps"][None], T6[f"{tag}_u"][None], log_priori=lp)
    check_replay(mc, tag, acc[0], H[0], h0=h0)
    if "wl" in SPECS[tag]:
        check_wl(mc, tag)


@pytest.mark.parametrize("tag", TABLE)
def test_replay_with_the_oracles_own_priori_factor(tag):
    """log_priori = NULL: the factor is derived from the step (_get_flip_id over
    delta_counts_from_step) -- it must equal the reference's number and lead to the same chain."""
    tab, cfg, occ0, temp = build(tag)
    mc = orc.OracleMC(tab, cfg)
    mc.set_state(occ0[None], [0], temp)
    acc, H, lp = mc.replay(T6[f"{tag}_steps"][None], T6[f"{tag}_u"][None], with_priori=True)
    check_replay(mc, tag, acc[0], H[0], lp_out=lp[0])
    # both kinds of step occur, and both signs of every direction
    ref = T6[f"{tag}_log_priori"]
    assert (ref[~np.isnan(ref)] != 0).any() and (ref == 0).any()


def test_step_outside_the_flip_table_is_an_error():
    tab, cfg, occ0, temp = build("TC_tf_int")
    mc = orc.OracleMC(tab, cfg)
    mc.set_state(occ0[None], [0], temp)
    site = int(np.flatnonzero(occ0[:64] == 0)[0])
    with pytest.raises(ValueError, match="not in flip table"):
        mc.replay(np.array([[[site, 1]]], dtype=np.int32), np.array([[0.5]]))  # Li+ -> Mn3+ alone


def test_table_proposals_have_the_shape_the_reference_draws():
    """Structure of the recorded reference steps (what the native-stream proposals of the oracle and
    the kernels must also satisfy, tests/test_table_flip.py): distinct sites, picked sites hold the
    depleted species, count change = +-(a table row) or a canonical swap."""
    for tag in ("TC_tf_int", "TG_tf_int", "TG6_tf_int"):
        tab, cfg, occ0, temp = build(tag)
        ev = orc.OracleEvaluator(tab)
        occ = occ0.copy()
        steps, acc = T6[f"{tag}_steps"], T6[f"{tag}_accepted"]
        seen = set()
        for k in range(len(steps)):
            fl = [(int(steps[k, 2 * j]), int(steps[k, 2 * j + 1])) for j in range(8) if steps[k, 2 * j] >= 0]
            assert len({s for s, _ in fl}) == len(fl)
            lp = ev.table_log_priori(occ, fl)
            np.testing.assert_allclose(lp, T6[f"{tag}_log_priori"][k], rtol=1e-10, atol=1e-10)
            seen.add(len(fl))
            if acc[k]:
                for s, c in fl:
                    occ[s] = c
        assert seen == ({2, 3, 6} if tag == "TG6_tf_int" else {2, 3})
