"""MCBias terms on the engine: trajectories equal the CPU oracle's (same Philox streams) with
either bias, the running trace.bias equals a recomputation, Sampler traces carry `bias`, and
the argument errors of the reference surface through the C ABI."""

import numpy as np
import pytest

from smol_amd import capi, moca, synth
from smol_amd.engine import Engine

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-10, 1e-9


@pytest.fixture(scope="module")
def rocksalt():
    model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 6.0, 3: 4.5})
    sc = synth.build_supercell(model, [3, 3, 3])
    return model, sc, synth.random_coefs(model, seed=4)


def _occ(sc, rng, R):
    occ = np.zeros((R, sc.num_sites), dtype=np.int32)
    occ[:, : sc.size] = rng.integers(0, 3, size=(R, sc.size))
    return occ


@pytest.mark.parametrize("general", [False, True], ids=["auto", "general-kernel"])
@pytest.mark.parametrize("ewald", [False, True], ids=["ce", "ce+ewald"])
@pytest.mark.parametrize("step", [capi.STEP_FLIP, capi.STEP_SWAP], ids=["flip", "swap"])
@pytest.mark.parametrize("kind", ["fugacity", "square-charge"])
def test_biased_trajectories_match_oracle(rocksalt, kind, step, ewald, general, monkeypatch):
    from oracle import oracle as orc

    if general:
        monkeypatch.setenv("SMOLMC_FORCE_GENERAL", "1")
    else:
        monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)

    model, sc, coefs = rocksalt
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs, ewald_coefficient=0.2 if ewald else None)
    names = ens.active_sublattices[0].species
    bias = (moca.FugacityBias(ens.sublattices, [{names[0]: 0.15, names[1]: 0.25, names[2]: 0.6}])
            if kind == "fugacity" else moca.SquareChargeBias(ens.sublattices, penalty=0.05))
    tab = ens.make_tables().set_bias(bias.bias_type, bias._table, bias.penalty)
    R = 7
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    rng = np.random.default_rng(21)
    occ0 = _occ(sc, rng, R)
    seeds = np.arange(1, R + 1, dtype=np.uint64) * np.uint64(6151)
    temps = np.linspace(600.0, 5000.0, R)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    assert eng.kernel_info().startswith("general" if general else "lean")  # both paths are pinned
    eng.set_state(occ0, seeds, temps)
    ora.set_state(occ0, seeds, temps)
    np.testing.assert_allclose(eng.get_bias(), ora.get_bias(), rtol=RTOL, atol=ATOL)
    for chunk in (1, 16, 300):
        eng.run(chunk)
        ora.run(chunk)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(eng.get_bias(), ora.get_bias(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(eng.get_bias(), [bias.compute_bias(o) for o in a["occupancy"]],
                               rtol=RTOL, atol=1e-8)
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()
    if step == capi.STEP_SWAP and kind == "square-charge":  # swaps conserve the charge
        np.testing.assert_allclose(eng.get_bias(), [bias.compute_bias(o) for o in occ0], atol=1e-9)


@pytest.fixture(scope="module")
def oxyfluoride():
    """Two active sublattices: Li+ / Mn3+ / Ti4+ cations, O2- / F- anions."""
    model = synth.build_cluster_model(synth.rocksalt_prim(anion_charges=(-2.0, -1.0)), {2: 6.0, 3: 4.5})
    sc = synth.build_supercell(model, [3, 3, 3])
    return model, sc, synth.random_coefs(model, seed=14)


@pytest.mark.parametrize("ewald", [False, True], ids=["ce", "ce+ewald"])
@pytest.mark.parametrize("step", [capi.STEP_FLIP, capi.STEP_SWAP], ids=["flip", "swap"])
@pytest.mark.parametrize("kind", ["fugacity", "square-charge"])
def test_biased_trajectories_two_sublattices(oxyfluoride, kind, step, ewald, monkeypatch):
    """MCBias on the multi-sublattice lean kernel (one bias row per sublattice): same chain as the
    oracle and as the general kernel; trace.bias equals a recomputation."""
    from oracle import oracle as orc

    monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    model, sc, coefs = oxyfluoride
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs, ewald_coefficient=0.15 if ewald else None)
    cat, an = ens.active_sublattices[0].species, ens.active_sublattices[1].species
    bias = (moca.FugacityBias(ens.sublattices, [{cat[0]: 0.15, cat[1]: 0.25, cat[2]: 0.6}, {an[0]: 0.7, an[1]: 0.3}])
            if kind == "fugacity" else moca.SquareChargeBias(ens.sublattices, penalty=0.05))
    tab = ens.make_tables().set_bias(bias.bias_type, bias._table, bias.penalty)
    R = 6
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    rng = np.random.default_rng(22)
    occ0 = np.zeros((R, sc.num_sites), dtype=np.int32)
    occ0[:, : sc.size] = rng.integers(0, 3, size=(R, sc.size))
    occ0[:, sc.size:] = rng.integers(0, 2, size=(R, sc.size))
    seeds = np.arange(1, R + 1, dtype=np.uint64) * np.uint64(7919)
    temps = np.linspace(700.0, 5000.0, R)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    assert eng.kernel_info().startswith("lean-multi"), eng.kernel_info()
    monkeypatch.setenv("SMOLMC_FORCE_GENERAL", "1")
    gen = Engine(tab, cfg)
    assert gen.kernel_info().startswith("general")
    for e in (eng, gen, ora):
        e.set_state(occ0, seeds, temps)
    for chunk in (1, 16, 17, 300):
        for e in (eng, gen, ora):
            e.run(chunk)
        a = eng.get_state()
        for x in (gen, ora):
            b = x.get_state()
            assert np.array_equal(a["occupancy"], b["occupancy"])
            assert np.array_equal(a["n_accepted"], b["n_accepted"])
            np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(a["features"], b["features"], rtol=RTOL, atol=1e-8)
            np.testing.assert_allclose(eng.get_bias(), x.get_bias(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(eng.get_bias(), [bias.compute_bias(o) for o in a["occupancy"]],
                               rtol=RTOL, atol=1e-8)
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()
    smp = eng.run_sampled(3, 11)  # the sample rows of a biased multi-sublattice walker
    for e in (gen, ora):
        e.run(33)
    np.testing.assert_allclose(smp["enthalpy"][-1], ora.get_state()["enthalpy"], rtol=RTOL, atol=ATOL)
    assert np.array_equal(smp["occupancy"][-1], gen.get_state()["occupancy"])


def test_bias_errors_surface(rocksalt):
    model, sc, coefs = rocksalt
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs)
    bias = moca.SquareChargeBias(ens.sublattices, penalty=0.5)
    tab = ens.make_tables().set_bias(bias.bias_type, bias._table, bias.penalty)
    with pytest.raises((RuntimeError, ValueError), match="Wang-Landau"):
        Engine(tab, capi.make_config(2, capi.KERNEL_WANGLANDAU, capi.STEP_SWAP, min_enthalpy=-5.0,
                                     max_enthalpy=5.0, bin_size=0.5))
    plain = Engine(ens.make_tables(), capi.make_config(2))
    with pytest.raises((RuntimeError, ValueError), match="no bias"):
        plain.get_bias()


def test_sampler_with_bias_traces_it(rocksalt):
    """Sampler.from_ensemble(..., bias_type=...) (kernel/base.py:229-235): the container holds
    a `bias` trace equal to compute_bias of the sampled occupancies."""
    model, sc, coefs = rocksalt
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs)
    nw = 4
    sampler = moca.Sampler.from_ensemble(ens, temperature=2500, step_type="flip", nwalkers=nw,
                                         seeds=[3, 4, 5, 6], bias_type="square-charge",
                                         bias_kwargs={"penalty": 0.1})
    occ = _occ(sc, np.random.default_rng(0), nw)
    sampler.run(1200, occ, thin_by=200)
    c = sampler.samples
    b = c.get_trace_value("bias", flat=False)
    occs = c.get_occupancies(flat=False)
    bias = sampler.mckernels[0].bias
    assert b.shape == (6, nw, 1)
    for i in range(6):
        np.testing.assert_allclose(b[i, :, 0], [bias.compute_bias(o) for o in occs[i]], atol=1e-8)
    q = bias._table
    charge = np.array([q[np.arange(sc.num_sites), o].sum() for o in occs[-1]])
    charge0 = np.array([q[np.arange(sc.num_sites), o].sum() for o in occ])
    assert np.abs(charge).mean() < np.abs(charge0).mean()


def test_table_flip_without_table_uses_composition_space():
    """TableFlip with no flip_table builds it from the sublattices (mcusher.py:489-518): for
    Li+/Mn3+/Ti4+ over fixed O2- that is 3 Mn3+ <-> Li+ + 2 Ti4+, charge stays zero."""
    model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 3.5})
    sc = synth.build_supercell(model, [3, 3, 3])
    ens = moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model, seed=5, scale=0.01))
    sampler = moca.Sampler.from_ensemble(ens, temperature=3000, step_type="table-flip", nwalkers=3,
                                         seeds=[1, 2, 3])
    table = np.asarray(sampler.mckernels[0].usher_kwargs["flip_table"])
    assert sorted(map(tuple, np.concatenate([table, -table]).tolist())) == [(-1, 3, -2, 0), (1, -3, 2, 0)]
    P = sc.size  # 27 cations: neutral with n_Ti = 3, n_Mn = 9, n_Li = 15
    rng = np.random.default_rng(2)
    occ = np.zeros((3, sc.num_sites), dtype=np.int32)
    for r in range(3):
        perm = rng.permutation(P)
        occ[r, perm[:9]] = 1
        occ[r, perm[9:12]] = 2
    sampler.run(3000, occ, thin_by=500)
    occs = sampler.samples.get_occupancies(flat=False)[:, :, :P]
    q = np.array([1, 3, 4])
    assert np.all((q[occs].sum(axis=-1) - 2 * P) == 0)  # charge neutral at every sample
    assert len({int((o == 2).sum()) for o in occs.reshape(-1, P)}) > 1  # composition moved


@pytest.mark.parametrize("general", [False, True], ids=["auto", "general-kernel"])
@pytest.mark.parametrize("step", [capi.STEP_FLIP, capi.STEP_SWAP], ids=["flip", "swap"])
def test_square_hyperplane_bias_matches_oracle(rocksalt, step, general, monkeypatch):
    """SquareHyperplaneBias (bias.py:290-366, two hyperplanes) on the engine == oracle, running
    bias == recomputation from the species counts, through the smol-shaped Sampler as well.  On the biased lean
    kernel since round 5 (one running sum per hyperplane), and on mc_kernel as before."""
    from oracle import oracle as orc

    if general:
        monkeypatch.setenv("SMOLMC_FORCE_GENERAL", "1")
    else:
        monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    model, sc, coefs = rocksalt
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs, ewald_coefficient=0.2)
    A, b = [[0, 1, 0, 0], [1, 0, -1, 0]], [sc.size // 3, 1]
    bias = moca.SquareHyperplaneBias(ens.sublattices, A, b, penalty=0.05)
    tab = ens.make_tables().set_bias(bias.bias_type, bias._table, bias.penalty, intercepts=bias.intercepts)
    R = 7
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    rng = np.random.default_rng(23)
    occ0 = _occ(sc, rng, R)
    seeds = np.arange(1, R + 1, dtype=np.uint64) * np.uint64(977)
    temps = np.linspace(600.0, 5000.0, R)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    assert eng.kernel_info().startswith("general" if general else "lean"), eng.kernel_info()  # both paths are pinned
    eng.set_state(occ0, seeds, temps)
    ora.set_state(occ0, seeds, temps)
    np.testing.assert_allclose(eng.get_bias(), [bias.compute_bias(o) for o in occ0], rtol=RTOL, atol=ATOL)
    for chunk in (1, 16, 400):
        eng.run(chunk)
        ora.run(chunk)
        a, b_ = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b_["occupancy"])
        assert np.array_equal(a["n_accepted"], b_["n_accepted"])
        np.testing.assert_allclose(eng.get_bias(), ora.get_bias(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(eng.get_bias(), [bias.compute_bias(o) for o in a["occupancy"]], rtol=RTOL, atol=1e-8)
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()
    sampler = moca.Sampler.from_ensemble(ens, temperature=1500.0, step_type="flip", nwalkers=3, seeds=[1, 2, 3],
                                         bias_type="square-hyperplane",
                                         bias_kwargs=dict(hyperplane_normals=A, hyperplane_intercepts=b, penalty=0.05))
    sampler.run(600, occ0[:3], thin_by=200)
    occs = sampler.samples.get_occupancies(flat=False)
    got = sampler.samples.get_trace_value("bias", flat=False)[..., 0]
    np.testing.assert_allclose(got, [[bias.compute_bias(o) for o in row] for row in occs], rtol=RTOL, atol=1e-8)


@pytest.mark.parametrize("step", [capi.STEP_FLIP, capi.STEP_SWAP], ids=["flip", "swap"])
def test_square_hyperplane_bias_two_sublattices(oxyfluoride, step, monkeypatch):
    """Three hyperplanes over the species of two active sublattices on the multi-sublattice lean kernel: the oracle's
    chain, the running bias a recomputation, the rows of the device ring with the bias column."""
    from oracle import oracle as orc

    monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    model, sc, coefs = oxyfluoride
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs, ewald_coefficient=0.15)
    ndim = sum(len(sl.species) for sl in ens.active_sublattices)
    rng = np.random.default_rng(31)
    A = rng.integers(-1, 2, size=(3, ndim)).tolist()
    b = [sc.size // 4, 0, -2]
    bias = moca.SquareHyperplaneBias(ens.sublattices, A, b, penalty=0.02)
    tab = ens.make_tables().set_bias(bias.bias_type, bias._table, bias.penalty, intercepts=bias.intercepts)
    R = 6
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    occ0 = np.zeros((R, sc.num_sites), dtype=np.int32)
    occ0[:, : sc.size] = rng.integers(0, 3, size=(R, sc.size))
    occ0[:, sc.size:] = rng.integers(0, 2, size=(R, sc.num_sites - sc.size))
    seeds = np.arange(1, R + 1, dtype=np.uint64) * np.uint64(613)
    temps = np.linspace(800.0, 5000.0, R)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    assert eng.kernel_info().startswith("lean-multi"), eng.kernel_info()
    eng.set_state(occ0, seeds, temps)
    ora.set_state(occ0, seeds, temps)
    for chunk in (1, 16, 400):
        eng.run(chunk)
        ora.run(chunk)
        a, b_ = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b_["occupancy"])
        assert np.array_equal(a["n_accepted"], b_["n_accepted"])
        np.testing.assert_allclose(eng.get_bias(), ora.get_bias(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(eng.get_bias(), [bias.compute_bias(o) for o in a["occupancy"]], rtol=RTOL, atol=1e-8)
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()
    smp = eng.run_sampled(3, 20, bias=True)
    for j in range(3):
        ora.run(20)
        assert np.array_equal(smp["occupancy"][j], ora.get_state()["occupancy"])
        np.testing.assert_allclose(smp["bias"][j], ora.get_bias(), rtol=RTOL, atol=ATOL)
    eng.close()


@pytest.mark.parametrize("ewald", [False, True], ids=["ce", "ce+ewald"])
@pytest.mark.parametrize("kind", ["fugacity", "square-charge", "square-hyperplane"])
def test_table_flip_with_a_bias_on_the_lean_table_kernel(rocksalt, kind, ewald, monkeypatch):
    """TableFlip composed with an MCBias term (the reference composes any usher with any bias, kernel/base.py:192-239;
    its charge-balanced recipes are the TableFlip usher and the square-charge bias): mc_table_kernel<..., BIAS> since
    round 6 -- the universal kernel until then (SMOLMC_NO_TABLE_BIAS: the A/B switch, same chain).  Same chain as the
    oracle on identical Philox streams; the running trace.bias equals a recomputation; the bias column of the device
    ring; a replayed record of the biased handle (the universal kernel replays it)."""
    from oracle import oracle as orc

    for k in ("SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL", "SMOLMC_NO_TABLE_BIAS"):
        monkeypatch.delenv(k, raising=False)
    model, sc, coefs = rocksalt
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs, ewald_coefficient=0.2 if ewald else None)
    names = ens.active_sublattices[0].species
    if kind == "fugacity":
        bias = moca.FugacityBias(ens.sublattices, [{names[0]: 0.15, names[1]: 0.25, names[2]: 0.6}])
    elif kind == "square-charge":
        bias = moca.SquareChargeBias(ens.sublattices, penalty=0.05)
    else:
        bias = moca.SquareHyperplaneBias(ens.sublattices, [[0, 1, 0, 0], [1, 0, -1, 0]], [sc.size // 3, 1], penalty=0.05)
    tab = ens.make_tables(flip_table=[[1, -3, 2, 0]], swap_weight=0.2)
    tab.set_bias(bias.bias_type, bias._table, bias.penalty, intercepts=getattr(bias, "intercepts", None))
    R = 6
    P = sc.size
    rng = np.random.default_rng(31)
    occ0 = np.zeros((R, sc.num_sites), dtype=np.int32)
    for r in range(R):  # charge neutral: 2 n_Mn + 3 n_Ti = P
        n_ti = 1 + 2 * (r % 3)
        n_mn = (P - 3 * n_ti) // 2
        perm = rng.permutation(P)
        occ0[r, perm[:n_mn]] = 1
        occ0[r, perm[n_mn:n_mn + n_ti]] = 2
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP)
    seeds = np.arange(1, R + 1, dtype=np.uint64) * np.uint64(7919)
    temps = np.linspace(1500.0, 9000.0, R)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    assert eng.kernel_info().startswith("lean "), eng.kernel_info()
    monkeypatch.setenv("SMOLMC_NO_TABLE_BIAS", "1")
    univ = Engine(tab, cfg)
    monkeypatch.delenv("SMOLMC_NO_TABLE_BIAS")
    assert univ.kernel_info().startswith("universal"), univ.kernel_info()
    for e in (eng, ora, univ):
        e.set_state(occ0, seeds, temps)
    np.testing.assert_allclose(eng.get_bias(), ora.get_bias(), rtol=RTOL, atol=ATOL)
    for chunk in (1, 16, 62, 400):
        for e in (eng, ora, univ):
            e.run(chunk)
        a, b, c = eng.get_state(), ora.get_state(), univ.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(eng.get_bias(), ora.get_bias(), rtol=RTOL, atol=ATOL)
        assert np.array_equal(a["occupancy"], c["occupancy"])
    np.testing.assert_allclose(eng.get_bias(), [bias.compute_bias(o) for o in a["occupancy"]], rtol=RTOL, atol=1e-8)
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()
    # the device ring with the bias column (launch + snapshot per sample)
    s = eng.run_sampled(3, 40, occupancy=True, bias=True)
    for j in range(3):
        ora.run(40)
        assert np.array_equal(s["occupancy"][j], ora.get_state()["occupancy"])
        np.testing.assert_allclose(s["bias"][j], ora.get_bias(), rtol=RTOL, atol=ATOL)
    # an empty replayed record: the biased handle has no REPLAY instantiation of its own, the universal kernel takes it
    st = np.full((R, 1, capi.STEP_ROW), -1, dtype=np.int32)
    acc_r, H_r = eng.replay(st, np.full((R, 1), 0.5))
    acc_o, H_o = ora.replay(st, np.full((R, 1), 0.5))  # (a replayed step advances the walker's step counter)
    assert np.array_equal(acc_r, acc_o)
    np.testing.assert_allclose(H_r, H_o, rtol=RTOL, atol=1e-8)
    eng.run(50)
    ora.run(50)
    assert np.array_equal(eng.get_state()["occupancy"], ora.get_state()["occupancy"])
    np.testing.assert_allclose(eng.get_bias(), ora.get_bias(), rtol=RTOL, atol=ATOL)
    eng.close()
    univ.close()


@pytest.mark.parametrize("ewald", [False, True], ids=["ce", "ce+ewald"])
@pytest.mark.parametrize("kind", ["fugacity", "square-charge", "square-hyperplane"])
def test_table_flip_with_a_bias_across_two_sublattices(kind, ewald, monkeypatch):
    """... and with a flip table that spans the cation and the anion sublattice (the shape of the reference's own
    TableFlip tests, tests/test_moca/test_mcushers.py:199-319): mc_table_multi_kernel<..., BIAS> (round 6), one pair
    table per sublattice and bias row.  Same chain as the oracle and as the universal kernel, trace.bias equals a
    recomputation, the bias column of the device ring is recorded in-kernel."""
    from oracle import oracle as orc

    for k in ("SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL", "SMOLMC_NO_TABLE_BIAS"):
        monkeypatch.delenv(k, raising=False)
    model = synth.build_cluster_model(synth.rocksalt_prim(anion_charges=(-2.0, -1.0)), {2: 4.5})
    sc = synth.build_supercell(model, [3, 3, 3])
    ens = moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model, seed=8, scale=0.02),
                                               ewald_coefficient=0.05 if ewald else None)
    cat, ani = (sl.species for sl in ens.active_sublattices)
    if kind == "fugacity":
        bias = moca.FugacityBias(ens.sublattices, [{cat[0]: 0.15, cat[1]: 0.25, cat[2]: 0.6}, {ani[0]: 0.7, ani[1]: 0.3}])
    elif kind == "square-charge":
        bias = moca.SquareChargeBias(ens.sublattices, penalty=0.05)
    else:
        bias = moca.SquareHyperplaneBias(ens.sublattices, [[0, 1, 0, 0, 0], [1, 0, -1, 0, 1]], [sc.size // 3, 1], penalty=0.05)
    table = np.asarray(ens.composition_space(optimize_basis=True, table_ergodic=True).flip_table)
    tab = ens.make_tables(flip_table=table, swap_weight=0.15)
    tab.set_bias(bias.bias_type, bias._table, bias.penalty, intercepts=getattr(bias, "intercepts", None))
    R, P = 6, sc.size
    rng = np.random.default_rng(13)
    occ0 = np.zeros((R, sc.num_sites), dtype=np.int32)
    for r in range(R):  # 17 Li+ + 8 Mn3+ + 2 Ti4+ = +49, 22 O2- + 5 F- = -49
        perm = rng.permutation(P)
        occ0[r, perm[:8]] = 1
        occ0[r, perm[8:10]] = 2
        occ0[r, P + rng.permutation(P)[:5]] = 1
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP)
    seeds = np.arange(1, R + 1, dtype=np.uint64) * np.uint64(7919)
    temps = np.linspace(1500.0, 9000.0, R)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    assert eng.kernel_info().startswith("lean-multi"), eng.kernel_info()
    monkeypatch.setenv("SMOLMC_NO_TABLE_BIAS", "1")
    univ = Engine(tab, cfg)
    monkeypatch.delenv("SMOLMC_NO_TABLE_BIAS")
    assert univ.kernel_info().startswith("universal"), univ.kernel_info()
    for e in (eng, ora, univ):
        e.set_state(occ0, seeds, temps)
    np.testing.assert_allclose(eng.get_bias(), ora.get_bias(), rtol=RTOL, atol=ATOL)
    for chunk in (1, 16, 62, 400, 700):
        for e in (eng, ora, univ):
            e.run(chunk)
        a, b, c = eng.get_state(), ora.get_state(), univ.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(a["features"], b["features"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(eng.get_bias(), ora.get_bias(), rtol=RTOL, atol=ATOL)
        assert np.array_equal(a["occupancy"], c["occupancy"])
    np.testing.assert_allclose(eng.get_bias(), [bias.compute_bias(o) for o in a["occupancy"]], rtol=RTOL, atol=1e-8)
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()
    s = eng.run_sampled(3, 40, occupancy=True, bias=True)
    for j in range(3):
        ora.run(40)
        assert np.array_equal(s["occupancy"][j], ora.get_state()["occupancy"])
        np.testing.assert_allclose(s["bias"][j], ora.get_bias(), rtol=RTOL, atol=ATOL)
    st = np.full((R, 1, capi.STEP_ROW), -1, dtype=np.int32)  # (a replayed record: the universal kernel takes it)
    acc_r, H_r = eng.replay(st, np.full((R, 1), 0.5))
    acc_o, H_o = ora.replay(st, np.full((R, 1), 0.5))
    assert np.array_equal(acc_r, acc_o)
    np.testing.assert_allclose(H_r, H_o, rtol=RTOL, atol=1e-8)
    eng.run(50)
    ora.run(50)
    assert np.array_equal(eng.get_state()["occupancy"], ora.get_state()["occupancy"])
    np.testing.assert_allclose(eng.get_bias(), ora.get_bias(), rtol=RTOL, atol=ATOL)
    eng.close()
    univ.close()
