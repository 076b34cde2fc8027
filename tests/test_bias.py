"""MCBias terms (smol/moca/kernel/bias.py) -- host records and the CPU oracle.

Restates the reference's own checks (tests/test_moca/test_bias.py): compute_bias_change equals
compute_bias(after) - compute_bias(before) for random steps; argument validation
(fractions add to one, species must match, penalty > 0); plus the physics the terms exist
for: with a zero Hamiltonian a FugacityBias makes single flips sample the fugacity
fractions, and a SquareChargeBias confines the net charge."""

import numpy as np
import pytest

from oracle import oracle as orc
from smol_amd import capi, moca, synth


@pytest.fixture(scope="module")
def rocksalt():
    model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 3.5})
    sc = synth.build_supercell(model, [3, 3, 3])
    return model, sc


def _ensemble(rocksalt, scale=0.02):
    model, sc = rocksalt
    return moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model, seed=5, scale=scale))


def _rand_occ(sc, rng):
    occ = np.zeros(sc.num_sites, dtype=np.int32)
    occ[: sc.size] = rng.integers(0, 3, size=sc.size)
    return occ


def test_bias_tables_and_validation(rocksalt):
    ens = _ensemble(rocksalt)
    names = ens.active_sublattices[0].species
    fb = moca.FugacityBias(ens.sublattices, [{names[0]: 0.2, names[1]: 0.3, names[2]: 0.5}])
    sc = rocksalt[1]
    assert fb._table.shape == (sc.num_sites, 3)
    np.testing.assert_allclose(fb._table[: sc.size], [[0.2, 0.3, 0.5]] * sc.size)
    np.testing.assert_allclose(fb._table[sc.size:], 1.0)  # inactive rows are ones (bias.py:221)
    with pytest.raises(ValueError, match="add to one"):
        moca.FugacityBias(ens.sublattices, [{names[0]: 0.2, names[1]: 0.3, names[2]: 0.6}])
    with pytest.raises(ValueError, match="missing or not valid"):
        moca.FugacityBias(ens.sublattices, [{names[0]: 0.5, names[1]: 0.5}])
    default = moca.FugacityBias(ens.sublattices)
    np.testing.assert_allclose(default._table[: sc.size], 1 / 3)
    cb = moca.SquareChargeBias(ens.sublattices, penalty=0.7)
    np.testing.assert_allclose(cb._table[0], [1, 3, 4])
    np.testing.assert_allclose(cb._table[sc.size], [-2, 0, 0])
    with pytest.raises(ValueError, match="Penalty factor"):
        moca.SquareChargeBias(ens.sublattices, penalty=0.0)
    with pytest.raises(NotImplementedError):
        moca.mcbias_factory("nonexistent-bias", ens.sublattices)
    k = moca.Metropolis(ens, "flip", 1000.0, bias_type="square-charge", bias_kwargs={"penalty": 0.3})
    assert isinstance(k.bias, moca.SquareChargeBias) and k.spec["bias"]["penalty"] == 0.3
    with pytest.raises(ValueError, match="Wang-Landau"):  # wanglandau.py:127-128
        moca.WangLandau(ens, "swap", 0.0, 10.0, 0.5, bias_type="fugacity")
    s = moca.Sampler.from_ensemble(ens, temperature=900, step_type="flip", bias_type="fugacity", nwalkers=2)
    assert "bias" in s.samples.traced_values


@pytest.mark.parametrize("kind", ["fugacity", "square-charge"])
def test_oracle_bias_change_is_difference(rocksalt, kind):
    """tests/test_moca/test_bias.py: delta == bias(next) - bias(now) for random steps."""
    ens = _ensemble(rocksalt)
    sc = rocksalt[1]
    names = ens.active_sublattices[0].species
    bias = (moca.FugacityBias(ens.sublattices, [{names[0]: 0.15, names[1]: 0.25, names[2]: 0.6}])
            if kind == "fugacity" else moca.SquareChargeBias(ens.sublattices, penalty=0.4))
    tab = ens.make_tables().set_bias(bias.bias_type, bias._table, bias.penalty)
    ev = orc.OracleEvaluator(tab)
    rng = np.random.default_rng(2)
    for _ in range(50):
        occ = _rand_occ(sc, rng)
        np.testing.assert_allclose(ev.bias(occ), bias.compute_bias(occ), rtol=1e-13)
        sites = rng.choice(sc.size, size=rng.integers(1, 4), replace=False)
        flips = [(int(s), int((occ[s] + rng.integers(1, 3)) % 3)) for s in sites]
        new = occ.copy()
        for s, c in flips:
            new[s] = c
        np.testing.assert_allclose(ev.bias_change(occ, flips), bias.compute_bias(new) - bias.compute_bias(occ),
                                   rtol=1e-10, atol=1e-10)
    # a site flipped twice in one step only counts its last flip (bias.py:199-200)
    occ = _rand_occ(sc, rng)
    twice = [(4, int((occ[4] + 1) % 3)), (4, int((occ[4] + 2) % 3))]
    new = occ.copy()
    new[4] = twice[-1][1]
    np.testing.assert_allclose(ev.bias_change(occ, twice), bias.compute_bias(new) - bias.compute_bias(occ),
                               rtol=1e-10, atol=1e-10)


def test_oracle_fugacity_bias_samples_the_fractions(rocksalt):
    """Zero Hamiltonian + single flips + FugacityBias: species fractions -> fugacity fractions."""
    model, sc = rocksalt
    ens = moca.Ensemble.from_cluster_expansion(sc, np.zeros(model.num_corr_functions))
    names = ens.active_sublattices[0].species
    fus = {names[0]: 0.2, names[1]: 0.3, names[2]: 0.5}
    bias = moca.FugacityBias(ens.sublattices, [fus])
    tab = ens.make_tables().set_bias(bias.bias_type, bias._table)
    R = 16
    mc = orc.OracleMC(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_FLIP))
    rng = np.random.default_rng(4)
    mc.set_state(np.array([_rand_occ(sc, rng) for _ in range(R)]), np.arange(R, dtype=np.uint64) + 1, 700.0)
    mc.run(400)
    counts = np.zeros(3)
    for _ in range(200):
        mc.run(30)
        o = mc.get_state()["occupancy"][:, : sc.size]
        counts += np.bincount(o.ravel(), minlength=3)
    np.testing.assert_allclose(counts / counts.sum(), [0.2, 0.3, 0.5], atol=0.01)
    st = mc.get_state()
    ev = orc.OracleEvaluator(tab)
    np.testing.assert_allclose(mc.get_bias(), [ev.bias(o) for o in st["occupancy"]], rtol=1e-10, atol=1e-9)


def test_oracle_square_charge_bias_confines_the_charge(rocksalt):
    """Random start (net charge ~ +19) relaxes to |charge| <= a few e under the penalty and the
    running bias stays -penalty * charge^2."""
    model, sc = rocksalt
    ens = moca.Ensemble.from_cluster_expansion(sc, np.zeros(model.num_corr_functions))
    bias = moca.SquareChargeBias(ens.sublattices, penalty=0.5)
    tab = ens.make_tables().set_bias(bias.bias_type, bias._table, bias.penalty)
    R = 8
    mc = orc.OracleMC(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_FLIP))
    rng = np.random.default_rng(5)
    occ0 = np.array([_rand_occ(sc, rng) for _ in range(R)])
    mc.set_state(occ0, np.arange(R, dtype=np.uint64) + 9, 1000.0)
    q = bias._table
    c0 = np.array([q[np.arange(sc.num_sites), o].sum() for o in occ0])
    np.testing.assert_allclose(mc.get_bias(), -0.5 * c0**2)
    mc.run(3000)
    st = mc.get_state()
    c = np.array([q[np.arange(sc.num_sites), o].sum() for o in st["occupancy"]])
    assert np.abs(c0).mean() > 8 and np.abs(c).max() <= 3
    np.testing.assert_allclose(mc.get_bias(), -0.5 * c**2, atol=1e-9)


def _hyperplanes(sc):
    """Two composition constraints on the counts vector (Li, Mn, Ti | O): n_Mn = P/3 and
    n_Li - n_Ti = 1 (per supercell, as the reference defines A n = b)."""
    return [[0, 1, 0, 0], [1, 0, -1, 0]], [sc.size // 3, 1]


def test_square_hyperplane_bias_host_and_oracle(rocksalt):
    """SquareHyperplaneBias (bias.py:290-366): the host record reproduces the reference formula
    -penalty * ||A n - b||^2 on the species counts, the oracle's table form equals it, and its
    change equals bias(after) - bias(before) (tests/test_moca/test_bias.py)."""
    ens = _ensemble(rocksalt)
    sc = rocksalt[1]
    A, b = _hyperplanes(sc)
    bias = moca.mcbias_factory("square-hyperplane", ens.sublattices, hyperplane_normals=A,
                               hyperplane_intercepts=b, penalty=0.3)
    assert isinstance(bias, moca.SquareHyperplaneBias) and bias.d == 4
    assert bias._table.shape == (2, sc.num_sites, 3)
    np.testing.assert_array_equal(bias._table[1, 0], [1, 0, -1])  # cation site: A[1][Li, Mn, Ti]
    np.testing.assert_array_equal(bias._table[1, sc.size], [0, 0, 0])  # anion site: A[1][O] = 0, unused codes 0
    with pytest.raises(ValueError, match="Penalty factor"):
        moca.SquareHyperplaneBias(ens.sublattices, A, b, penalty=-1.0)
    with pytest.raises(ValueError, match="one column per"):
        moca.SquareHyperplaneBias(ens.sublattices, [[1, 0, 0]], [0])
    tab = ens.make_tables().set_bias(bias.bias_type, bias._table, bias.penalty, intercepts=bias.intercepts)
    assert tab.struct.bias_rows == 2
    ev = orc.OracleEvaluator(tab)
    rng = np.random.default_rng(3)
    for _ in range(40):
        occ = _rand_occ(sc, rng)
        n = np.array([(occ[: sc.size] == c).sum() for c in range(3)] + [sc.size])
        want = -0.3 * float(np.sum((np.array(A) @ n - np.array(b)) ** 2))
        np.testing.assert_array_equal(bias.counts(occ), n)
        assert bias.compute_bias(occ) == want
        np.testing.assert_allclose(ev.bias(occ), want, rtol=1e-13)
        sites = rng.choice(sc.size, size=rng.integers(1, 4), replace=False)
        flips = [(int(s), int((occ[s] + rng.integers(1, 3)) % 3)) for s in sites]
        new = occ.copy()
        for s_, c in flips:
            new[s_] = c
        np.testing.assert_allclose(ev.bias_change(occ, flips), bias.compute_bias(new) - bias.compute_bias(occ),
                                   rtol=1e-10, atol=1e-10)
    # the reference's SquareChargeBias is the one-hyperplane special case A = charges, b = 0
    q = moca.SquareChargeBias(ens.sublattices, penalty=0.3)
    hq = moca.SquareHyperplaneBias(ens.sublattices, [[1, 3, 4, -2]], [0], penalty=0.3)
    occ = _rand_occ(sc, rng)
    assert hq.compute_bias(occ) == q.compute_bias(occ)


def test_oracle_square_hyperplane_bias_pulls_onto_the_constraints(rocksalt):
    model, sc = rocksalt
    ens = moca.Ensemble.from_cluster_expansion(sc, np.zeros(model.num_corr_functions))
    A, b = _hyperplanes(sc)
    bias = moca.SquareHyperplaneBias(ens.sublattices, A, b, penalty=0.5)
    tab = ens.make_tables().set_bias(bias.bias_type, bias._table, bias.penalty, intercepts=bias.intercepts)
    R = 6
    mc = orc.OracleMC(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_FLIP))
    rng = np.random.default_rng(8)
    occ0 = np.zeros((R, sc.num_sites), dtype=np.int32)  # all Li: far from both hyperplanes
    mc.set_state(occ0, np.arange(R, dtype=np.uint64) + 3, 1000.0)
    np.testing.assert_allclose(mc.get_bias(), [bias.compute_bias(o) for o in occ0])
    mc.run(4000)
    st = mc.get_state()
    for o in st["occupancy"]:
        n = bias.counts(o)
        assert abs(n[1] - sc.size // 3) <= 2 and abs(n[0] - n[2] - 1) <= 3
    np.testing.assert_allclose(mc.get_bias(), [bias.compute_bias(o) for o in st["occupancy"]], atol=1e-9)
