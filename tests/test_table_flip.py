"""TableFlip (charge-neutral semigrand steps, smol/moca/kernel/mcusher.py:397-711).

The reference's own checks are (i) known a-priori factors for a flip table produced by
CompositionSpace (tests/test_moca/test_mcushers.py:199-234 -- reproduced on the host in
tests/test_composition.py, which also ties the oracle's factor to that host formula) and (ii) a detailed-
balance histogram over compositions (test_mcushers.py:237-319).  (ii) is restated here for
the oracle on CPU and, in tests/test_gpu_table_flip.py, for the engine; the a-priori
formula is additionally checked against hand-computed values."""

from math import comb, factorial, log

import numpy as np
import pytest

from oracle import oracle as orc
from smol_amd import capi, synth

# Li+ / Mn3+ / Ti4+ on the cation sublattice of rocksalt, fixed O2-: site count and charge
# conservation leave ONE flip direction: 3 Mn3+ -> 1 Li+ + 2 Ti4+
FLIP_TABLE = np.array([[1, -3, 2]])


def _model(dim, coef_scale=0.0, mu=None, ewald=False):
    from smol_amd import ewald as ew

    model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 3.5})
    sc = synth.build_supercell(model, [dim] * 3 if np.isscalar(dim) else list(dim))
    coefs = synth.random_coefs(model, seed=5, scale=coef_scale)
    mu_table = None
    if mu is not None:
        mu_table = np.zeros((sc.num_sites, 3))
        mu_table[: sc.size] = np.asarray(mu)[None, :]
    tab = capi.TableSet.from_synth(sc, coefs, ewald=ew.supercell_ewald(sc) if ewald else None,
                                   ewald_coef=0.05, mu_table=mu_table, flip_table=FLIP_TABLE,
                                   swap_weight=0.2)
    return sc, tab


def _neutral_occ(sc, n_ti, rng):
    P = sc.size  # cations 0..P-1; 2 n_Mn + 3 n_Ti = P for neutrality with O2-
    n_mn = (P - 3 * n_ti) // 2
    assert 2 * n_mn + 3 * n_ti == P
    occ = np.zeros(sc.num_sites, dtype=np.int32)
    perm = rng.permutation(P)
    occ[perm[:n_mn]] = 1
    occ[perm[n_mn:n_mn + n_ti]] = 2
    return occ


def test_log_priori_matches_hand_computation():
    sc, tab = _model(3)
    mc = orc.OracleMC(tab, capi.make_config(1, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP))
    rng = np.random.default_rng(0)
    occ = _neutral_occ(sc, 3, rng)  # n = (Li 15, Mn 9, Ti 3): both directions feasible
    mc.set_state(occ[None], [7], 1000.0)
    seen = set()
    for step in range(400):
        nf, fl, lp = mc.propose(0, step, with_priori=True)
        flips = [(int(fl[2 * i]), int(fl[2 * i + 1])) for i in range(nf)]
        dn = np.zeros(3, int)
        for s, c in flips:
            dn[occ[s]] -= 1
            dn[c] += 1
        if nf == 2 and not dn.any():  # canonical swap branch
            assert lp == 0.0 and occ[flips[0][0]] != occ[flips[1][0]]
            seen.add("swap")
            continue
        assert nf == 3 and len({s for s, _ in flips}) == 3
        n = np.array([15, 9, 3])
        sign = 1 if dn[0] == 1 else -1
        assert np.array_equal(dn, sign * FLIP_TABLE[0])
        depleted = {1} if sign == 1 else {0, 2}
        assert all(int(occ[s]) in depleted for s, _ in flips)  # picked sites hold depleted species
        # at n both directions are feasible (p_now = 0.8 * 1/2); at n_next too unless Mn < 3
        n_next = n + dn
        feas_next = sum(int(np.all(n_next + d >= 0)) for d in (FLIP_TABLE[0], -FLIP_TABLE[0]))
        expect = log((1 / feas_next) / (1 / 2))
        expect += sum(log(factorial(int(a))) - log(factorial(int(b))) for a, b in zip(n, n_next))
        assert lp == pytest.approx(expect, rel=1e-12)
        seen.add(sign)
    assert seen == {"swap", 1, -1}


def test_detailed_balance_over_compositions():
    """Zero Hamiltonian: the chain must visit composition k = n_Ti with probability
    proportional to the number of configurations, P!/(n_Li! n_Mn! n_Ti!)
    (the histogram test of tests/test_moca/test_mcushers.py:237-319)."""
    sc, tab = _model(3)  # 27 cations: n_Ti in {1,3,5,7,9}
    R = 16
    mc = orc.OracleMC(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP))
    rng = np.random.default_rng(1)
    occ = np.array([_neutral_occ(sc, 1 + 2 * (r % 5), rng) for r in range(R)])
    mc.set_state(occ, np.arange(R, dtype=np.uint64) + np.uint64(3), 1000.0)
    mc.run(300)
    counts = np.zeros(10)
    nsamp = 2500
    for _ in range(nsamp):
        mc.run(8)
        st = mc.get_state()
        for r in range(R):
            n_ti = int((st["occupancy"][r][: sc.size] == 2).sum())
            counts[n_ti] += 1
        assert np.all(st["occupancy"][:, sc.size:] == 0)  # anions untouched
    P = sc.size
    w = {}
    for n_ti in (1, 3, 5, 7, 9):
        n_mn = (P - 3 * n_ti) // 2
        w[n_ti] = comb(P, n_ti) * comb(P - n_ti, n_mn)
    tot = sum(w.values())
    for n_ti, wt in w.items():
        p = wt / tot
        got = counts[n_ti] / counts.sum()
        assert got == pytest.approx(p, abs=max(0.02, 6 * np.sqrt(p * (1 - p) / (nsamp * R / 10))))
    assert counts[[0, 2, 4, 6, 8]].sum() == 0  # charge neutrality never violated


def test_table_flip_trace_consistency():
    """Accepted multi-flip steps keep features / enthalpy equal to recomputed values."""
    sc, tab = _model(3, coef_scale=0.05, mu=[0.1, -0.2, 0.05], ewald=True)
    R = 4
    mc = orc.OracleMC(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP))
    rng = np.random.default_rng(2)
    occ = np.array([_neutral_occ(sc, 3, rng) for _ in range(R)])
    mc.set_state(occ, np.arange(R, dtype=np.uint64) + np.uint64(11), 3000.0)
    mc.run(600)
    st = mc.get_state()
    ev = orc.OracleEvaluator(tab)
    nat = ev.natural_parameters()
    for r in range(R):
        f = ev.feature_vector(st["occupancy"][r])
        np.testing.assert_allclose(st["features"][r], f, rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(st["enthalpy"][r], nat @ f, rtol=1e-10, atol=1e-8)
    assert 0 < st["n_accepted"].sum() < st["n_steps"].sum()


def test_composition_line_cache_never_serves_another_position():
    """The bookkeeping of mc_table_kernel's a-priori factors for a single flip vector (mc_lean.h, lp_line), restated:
    lane (k & 63) holds F(k) = f(k, +u), direction d reads slot (kpos - (d & 1)) & 63 with the sign of d, and an accepted
    step moves kpos by one and drops the residues (kpos + 32) & 63 and (kpos + 33) & 63.  Over long random walks --
    drifting, oscillating, reversing after 31 / 32 / 33 steps -- a slot that is marked valid always holds the factor of
    the position it is read for."""
    rng = np.random.default_rng(11)

    def walk(moves):
        kpos, valid, held = 0, 0, [None] * 64  # (Python ints) held[lane] = the position whose factor the lane holds
        for mv in moves:
            for d in (0, 1):  # both directions are looked up at every position
                want = kpos - (d & 1)
                slot = want & 63
                if (valid >> slot) & 1:
                    assert held[slot] == want  # served from the cache: it must be THIS position's factor
                else:
                    held[slot], valid = want, valid | (1 << slot)
            kpos += int(mv)
            valid &= ~((1 << ((kpos + 32) & 63)) | (1 << ((kpos + 33) & 63)))
        return kpos

    walk(rng.choice([-1, 1], size=20000))                      # diffusive
    walk(rng.choice([-1, 1], size=20000, p=[0.3, 0.7]))        # drifting up (wraps the window many times)
    walk(rng.choice([-1, 1], size=20000, p=[0.7, 0.3]))        # drifting down
    for span in (30, 31, 32, 33, 34, 63, 64, 65):              # sawtooth around the window size
        walk(np.tile(np.concatenate([np.ones(span, int), -np.ones(span, int)]), 20))
