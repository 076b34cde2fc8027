"""CompositionSpace / flip tables / TableFlip a-priori factors on the host, pinned by the
reference's own known answers:
  tests/test_moca/test_comp_space.py:26-113,243-271   bases and flip tables of four systems
  tests/test_moca/test_mcushers.py:199-234            TableFlip.compute_log_priori_factor
plus the reference's generic invariants (test_comp_space.py:116-167) and the agreement of the
C oracle's single-sublattice a-priori factor with the multi-sublattice host version."""

import numpy as np
import pytest

from smol_amd import composition as cmp
from smol_amd.composition import CompositionSpace

LI, MN, TI, ZR, NI = ("Li+", 1), ("Mn3+", 3), ("Ti4+", 4), ("Zr4+", 4), ("Ni2+", 2)
O, P, F = ("O2-", -2), ("P3-", -3), ("F-", -1)


def table_set(a):
    a = np.asarray(a)
    return sorted(map(tuple, np.concatenate([a, -a]).tolist()))  # tests/utils.py:16-25


def lmtpo(**kw):
    return CompositionSpace([[LI, MN, TI], [P, O]], [1, 1], charge_neutral=True, optimize_basis=True,
                            table_ergodic=True, **kw)


STD1 = [[0, -1, 1, 1, -1], [-1, 1, 0, 2, -2]]
STD2 = [[-3, 1, 1, 1, 6, -6]]
STD3 = [[-1, 2, -1, 1, -1]]


def test_known_bases_and_flip_tables():
    """test_comp_space.py:248-271 ("Pre-computed optimal basis")."""
    c1 = lmtpo()
    c2 = CompositionSpace([[LI, NI, MN, TI], [O, F]], [1, 1], optimize_basis=True, table_ergodic=True,
                          other_constraints=[([0, 1, -1, 0, 0, 0], 0, "eq"), ([0, 0, 1, -1, 0, 0], 0, "eq")])
    c3 = lmtpo(other_constraints=[([2, 1, 0, 0, 0], 7 / 6, "eq")])
    c4 = lmtpo(other_constraints=[
        ([0, 1, 0, 0, 0], 5 / 6, "<="), ([0, 0, 0, 1, 0], 5 / 6, "<="), ([2, 1, 0, 0, 0], 8 / 6, "<="),
        ([0, 1, 0, 0, 0], 1 / 6, ">="), ([0, 0, 0, 1, 0], 1 / 6, ">="), ([2, 1, 0, 0, 0], 5 / 6, ">="),
    ])
    for c, std in ((c1, STD1), (c2, STD2), (c3, STD3), (c4, STD1)):
        assert table_set(c.basis) == table_set(std)
        assert table_set(c.flip_table) == table_set(std)
    assert c1.num_unconstrained_compositions == 7776  # test_comp_space.py:243-245
    assert c2.num_unconstrained_compositions == 46656
    assert c1.min_supercell_size == 6


def test_table_flip_log_priori_known_answers():
    """test_mcushers.py:199-234: 3 cation sites {Li+, Zr4+, Mn3+} + 3 anion sites {O2-, F-}."""
    cs = CompositionSpace([[LI, ZR, MN], [O, F]], [1, 1], optimize_basis=True, table_ergodic=True)
    table, w, max_n = cs.flip_table, np.ones(2 * len(cs.flip_table)), [3] * 5

    def counts(occ):
        occ = np.asarray(occ)
        return np.array([(occ[:3] == c).sum() for c in range(3)] + [(occ[3:] == c).sum() for c in range(2)])

    def factor(occ, step):
        new = list(occ)
        for s, c in step:
            new[s] = c
        return cmp.table_log_priori_factor(table, w, 0.1, counts(occ), counts(new) - counts(occ), max_n)

    assert np.isclose(factor([0, 0, 1, 0, 0, 0], [(2, 2), (4, 1)]), np.log(3 / 2))
    assert np.isclose(factor([0, 0, 2, 1, 0, 0], [(2, 1), (3, 0)]), np.log(2 / 3))
    assert np.isclose(factor([0, 0, 2, 1, 0, 0], [(2, 0), (4, 1), (5, 1)]), np.log(2 / 9))
    assert np.isclose(factor([0, 0, 0, 1, 1, 1], [(0, 2), (4, 0), (5, 0)]), np.log(9 / 2))
    assert np.isclose(factor([0, 0, 2, 1, 0, 0], [(2, 0), (0, 2)]), 0.0)
    with pytest.raises(ValueError, match="not in flip table"):
        cmp.table_log_priori_factor(table, w, 0.1, [1, 1, 1, 3, 0], [1, -1, 0, 0, 0], max_n)


@pytest.mark.parametrize("spaces,sizes", [
    ([[LI, MN, TI], [P, O]], [1, 1]),
    ([[LI, MN, TI], [O]], [1, 1]),
    ([[LI, NI, MN, TI], [O, F]], [1, 1]),
    ([[LI, ("Vacancy", 0), MN], [O, F]], [2, 1]),
])
def test_generic_invariants(spaces, sizes):
    """test_comp_space.py:116-167: A has one charge row + one row per sublattice, vertices sit on
    the polytope boundary, every grid point solves A n = b * size with n >= 0, the un-optimised
    flip table is the basis."""
    cs = CompositionSpace(spaces, sizes)
    A, b = cs._A, cs._b
    assert len(A) == 1 + len(spaces) and b[0] == 0
    np.testing.assert_array_equal(b[1:], sizes)
    v = cs.prim_vertices
    np.testing.assert_allclose(A @ v.T - b[:, None], 0, atol=1e-9)
    assert np.all(np.any(np.isclose(v, 0), axis=-1)) and np.all(v >= -1e-12)
    msc = cs.min_supercell_size
    for size in (1, msc, 2 * msc):
        if not np.allclose(size * b, np.round(size * b)):
            continue
        try:
            n0 = cs.get_supercell_base_solution(size)
        except ValueError:
            continue
        ns = cs.get_composition_grid(size) @ cs.basis + n0
        np.testing.assert_array_equal(A @ ns.T - np.round(b * size)[:, None].astype(int), 0)
        assert np.all(ns >= 0)
        assert len({tuple(r) for r in ns.tolist()}) == len(ns)
    np.testing.assert_array_equal(cs.flip_table, cs.basis)
    assert np.all(A @ cs.basis.T == 0) and np.linalg.matrix_rank(cs.basis) == len(cs.basis)
    # the grid at the minimal size is complete: brute-force count of natural solutions
    import itertools

    total = 0
    ranges = [range(s * msc + 1) for sl, s in zip(spaces, sizes) for _ in sl]
    if np.prod([len(r) for r in ranges]) <= 2_000_000:
        for n in itertools.product(*ranges):
            total += bool(np.array_equal(A @ np.array(n), np.round(b * msc).astype(int)))
        assert total == len(cs.get_composition_grid(msc))
    assert len(cs.flip_reactions) == len(cs.flip_table)
    assert all("->" in r for r in cs.flip_reactions)


def test_grid_step_and_scaling():
    """Grids at a multiple of the size with the same multiple as step are the scaled grid
    (space.py:292-296 comment, test_comp_space.py:285-330)."""
    cs = lmtpo()
    g1 = cs.get_composition_grid(6, step=1)
    g2 = cs.get_composition_grid(12, step=2)
    n1 = g1 @ cs.basis + cs.get_supercell_base_solution(6)
    n2 = g2 @ cs.basis + cs.get_supercell_base_solution(12)
    assert {tuple(r) for r in (2 * n1).tolist()} <= {tuple(r) for r in
                                                      (cs.get_composition_grid(12) @ cs.basis
                                                       + cs.get_supercell_base_solution(12)).tolist()}
    assert len(g2) <= len(cs.get_composition_grid(12)) and len(n2)
    with pytest.raises(ValueError):  # 7/6 per prim has no integer solution in 4 prims
        lmtpo(other_constraints=[([2, 1, 0, 0, 0], 7 / 6, "eq")]).get_supercell_base_solution(4)


def test_errors():
    with pytest.raises(ValueError, match="Sub-lattice number"):
        CompositionSpace([[LI, MN], [O]], [1])
    with pytest.raises(ValueError, match="does not match dimensions"):
        CompositionSpace([[LI, MN], [O]], [1, 1], other_constraints=[([1, 0], 0.5, "eq")])
    with pytest.raises(ValueError, match="more than number of dimensions"):
        CompositionSpace([[LI, MN], [O]], [1, 1],
                         other_constraints=[([1, 0, 0], 0.5, "eq")])
    with pytest.raises(ValueError, match="not feasible"):
        cmp.solve_diophantines([[2, 4]], [3])
    assert cmp.as_species("Mn3+") == ("Mn3+", 3) and cmp.as_species("O2-") == ("O2-", -2)
    assert cmp.as_species("F-") == ("F-", -1) and cmp.as_species("Vacancy").oxi_state == 0


def test_solve_diophantines_random():
    """smol tests/test_utils/test_math_utils.py:207-224 style: A n0 = b, A v = 0, full lattice."""
    rng = np.random.default_rng(0)
    for _ in range(40):
        m, d = rng.integers(1, 4), rng.integers(3, 7)
        A = rng.integers(-4, 5, size=(m, d))
        n_true = rng.integers(-5, 6, size=d)
        n0, vs = cmp.solve_diophantines(A, A @ n_true)
        np.testing.assert_array_equal(A @ n0, A @ n_true)
        assert np.all(A @ vs.T == 0)
        assert len(vs) == d - np.linalg.matrix_rank(A)
        # n_true - n0 must be an INTEGER combination of the basis (the basis spans the lattice)
        if len(vs):
            x, *_ = np.linalg.lstsq(vs.T.astype(float), (n_true - n0).astype(float), rcond=None)
            np.testing.assert_allclose(x, np.round(x), atol=1e-8)
            np.testing.assert_array_equal(np.round(x).astype(int) @ vs + n0, n_true)


def test_config5_table_and_oracle_factor_agree():
    """The flip table of bench config 5 (Li+/Mn3+/Ti4+ on a rocksalt cation sublattice, fixed O2-)
    comes out as 3 Mn3+ <-> Li+ + 2 Ti4+, and the C oracle's single-sublattice a-priori factor
    equals the host's multi-sublattice one."""
    from oracle import oracle as orc

    cs = CompositionSpace([[LI, MN, TI], [O]], [1, 1])
    assert table_set(cs.flip_table) == table_set([[1, -3, 2, 0]])
    sub = cs.sublattice_flip_table(0)
    assert table_set(sub) == table_set([[1, -3, 2]])
    with pytest.raises(ValueError, match="couples several sublattices"):
        lmtpo().sublattice_flip_table(0)
    # oracle proposals with a two-row table and uneven weights: its a-priori factor must be
    # the host formula evaluated on the counts before the step
    from smol_amd import capi, synth

    model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 3.5})
    sc = synth.build_supercell(model, [3, 3, 3])
    table = np.array([[1, -3, 2], [1, -1, 0]])
    w = np.array([1.0, 2.0, 0.5, 1.5])
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=5), flip_table=table,
                                   flip_weights=w, swap_weight=0.2)
    mc = orc.OracleMC(tab, capi.make_config(1, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP))
    rng = np.random.default_rng(3)
    occ = np.zeros(sc.num_sites, dtype=np.int32)
    occ[: sc.size] = rng.integers(0, 3, size=sc.size)
    mc.set_state(occ[None], [11], 1000.0)
    n = np.bincount(occ[: sc.size], minlength=3)
    kinds = set()
    for step in range(300):
        nf, fl, lp = mc.propose(0, step, with_priori=True)
        dn = np.zeros(3, dtype=np.int64)
        for i in range(nf):
            dn[occ[fl[2 * i]]] -= 1
            dn[fl[2 * i + 1]] += 1
        ref = cmp.table_log_priori_factor(table, w, 0.2, n, dn, max_n=sc.size)
        np.testing.assert_allclose(lp, ref, rtol=1e-12, atol=1e-12)
        kinds.add(tuple(dn.tolist()))
    assert len(kinds) >= 4  # swap + several table directions were proposed
