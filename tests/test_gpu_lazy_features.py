"""Lazy cluster features (engine.hip, build_mc_tables): Metropolis kernels of the lean families take
a correlation-mode model with several functions per orbit -- the reference's default ClusterExpansionProcessor on
any site space of three or more species (evaluator.pyx:211-265) -- as an interaction-mode model of the folded
tensors and evaluate the cluster features from the occupancy where they are read.  Against the oracle on
identical Philox streams: the chains, the feature vectors of smolmc_get_state, every row of the device ring."""

import numpy as np
import pytest

from smol_amd import capi, synth
from tests.cases import load_case, tables_for

pytestmark = pytest.mark.gpu
CORR = capi.FEATURES_CORRELATIONS


def _pair(tab, cfg, occ, seeds, temp):
    from oracle import oracle as orc
    from smol_amd.engine import Engine

    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    eng.set_state(occ, seeds, temp)
    ora.set_state(occ, seeds, temp)
    return eng, ora


def _same_chain(eng, ora, chunks, bias=False):
    for n in chunks:
        eng.run(n)
        ora.run(n)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"]), eng.kernel_info()
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(a["features"], b["features"], rtol=1e-10, atol=1e-8)
        if bias:
            np.testing.assert_allclose(eng.get_bias(), ora.get_bias(), rtol=1e-10, atol=1e-9)
    return a


def _rand_occ(sc, rng, R):
    nsp = np.array([sc.model.prim.nspecies[b] for b in sc.site_b])
    return (rng.random((R, sc.num_sites)) * nsp).astype(np.int32)


def _mu(sc, width, seed):
    mu = np.zeros((sc.num_sites, width))
    mu[:, :] = np.random.default_rng(seed).uniform(-0.3, 0.3, width)[None, :]
    return mu


@pytest.mark.parametrize("step", [capi.STEP_FLIP, capi.STEP_SWAP], ids=["flip", "swap"])
def test_two_sublattices_with_several_functions_per_orbit(step, monkeypatch):
    """Cations of three species and a second active sublattice: K > 1 correlation functions on a multi-class
    model (mc_kernel until round 5) on mc_lean_multi_kernel."""
    monkeypatch.delenv("SMOLMC_NO_LAZY_FEATURES", raising=False)
    c = load_case("rocksalt333_two_sublattices")
    sc = c["sc"]
    mu = _mu(sc, 3, 5) if step == capi.STEP_FLIP else None
    tab = tables_for("rocksalt333_two_sublattices", CORR, mu_table=mu)
    R = 6
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    eng, ora = _pair(tab, cfg, _rand_occ(sc, np.random.default_rng(3), R), np.arange(R, dtype=np.uint64) + np.uint64(70), 3000.0)
    info = eng.kernel_info()
    assert info.startswith("lean-multi") and "lazy-features" in info, info
    a = _same_chain(eng, ora, (1, 9, 400))
    assert a["n_accepted"].sum() > 50
    np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=1e-10, atol=1e-8)
    # the A/B switch puts the model back where it was
    monkeypatch.setenv("SMOLMC_NO_LAZY_FEATURES", "1")
    from smol_amd.engine import Engine

    old = Engine(tab, cfg)
    assert old.kernel_info().startswith("general"), old.kernel_info()
    old.close()
    eng.close()


def test_five_species_more_than_64_functions():
    """65 correlation functions, up to 16 per orbit (beyond SMOLMC_LEAN_MAX_KF and beyond the kernels' 64 feature
    cells): the single-class lean kernel, decisions from the folded tables."""
    model = synth.build_cluster_model(synth.fcc_prim(nspecies=5), {2: 6.0, 3: 3.0})
    assert model.num_corr_functions > 64
    sc = synth.build_supercell(model, [4, 4, 4])
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=6), feature_mode=CORR, mu_table=_mu(sc, 5, 2))
    R = 5
    occ = (np.random.default_rng(5).random((R, sc.num_sites)) * 5).astype(np.int32)
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_FLIP)
    eng, ora = _pair(tab, cfg, occ, np.arange(R, dtype=np.uint64) + np.uint64(11), np.linspace(800.0, 5000.0, R))
    info = eng.kernel_info()
    assert info.startswith("lean ") and "lazy-features" in info, info
    a = _same_chain(eng, ora, (1, 30, 500))
    assert a["n_accepted"].sum() > 50
    eng.close()


def _triclinic_prim(nspecies=2):
    """One site in a triclinic cell: inversion is the only point symmetry, so every +-vector pair and every
    triangle is an orbit of its own -- many orbits with few clusters each."""
    return synth.PrimCell(np.array([[3.1, 0.0, 0.0], [0.7, 3.4, 0.0], [0.5, 0.9, 3.7]]), [[0, 0, 0]], [nspecies])


def test_interaction_mode_with_more_than_64_orbits():
    """Interaction features of a model with more than 64 orbits: the same mechanism without folding (the kernels'
    feature cells are one per lane).  74 orbits, 221 clusters per site on a low-symmetry lattice -- round 5's
    conventional-cell quadruplet model had 67 orbits but 4687 clusters per site (beyond the lean families' 512) and
    the test skipped itself."""
    model = synth.build_cluster_model(_triclinic_prim(), {2: 10.0, 3: 6.0})
    assert model.num_orbits > 64
    sc = synth.build_supercell(model, [7, 7, 7])
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=9), feature_mode=capi.FEATURES_INTERACTIONS)
    assert tab.num_features > 64
    R = 4
    occ = (np.random.default_rng(1).random((R, sc.num_sites)) < 0.5).astype(np.int32)
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
    eng, ora = _pair(tab, cfg, occ, np.arange(R, dtype=np.uint64) + np.uint64(3), 2000.0)
    info = eng.kernel_info()
    assert info.startswith("lean") and "lazy-features" in info, info
    a = _same_chain(eng, ora, (1, 40, 300))
    assert a["n_accepted"].sum() > 50
    np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=1e-10, atol=1e-8)
    eng.close()


@pytest.mark.parametrize("kind", ["fugacity", "square-charge"])
def test_biased_kernels_with_several_functions_per_orbit(kind):
    """K = 3 / 4 / 6 functions per orbit with an MCBias term (the KF instantiations are unbiased): the biased lean
    kernel; bias column and feature rows of the device ring through the snapshot path."""
    from smol_amd import moca

    model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 6.0, 3: 4.5})
    sc = synth.build_supercell(model, [3, 3, 3])
    ens = moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model, seed=4), processor_type="expansion",
                                               ewald_coefficient=0.2)
    names = ens.active_sublattices[0].species
    bias = (moca.FugacityBias(ens.sublattices, [{names[0]: 0.15, names[1]: 0.25, names[2]: 0.6}])
            if kind == "fugacity" else moca.SquareChargeBias(ens.sublattices, penalty=0.05))
    tab = ens.make_tables().set_bias(bias.bias_type, bias._table, bias.penalty)
    R = 5
    rng = np.random.default_rng(4)
    occ = np.zeros((R, sc.num_sites), dtype=np.int32)
    occ[:, : sc.size] = rng.integers(0, 3, size=(R, sc.size))
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_FLIP)
    eng, ora = _pair(tab, cfg, occ, np.arange(R, dtype=np.uint64) + np.uint64(21), np.linspace(1500.0, 6000.0, R))
    info = eng.kernel_info()
    assert info.startswith("lean") and "lazy-features" in info, info
    _same_chain(eng, ora, (1, 25, 300), bias=True)
    s = eng.run_sampled(4, 50, occupancy=True, bias=True)  # (one launch + snapshot per sample)
    for j in range(4):
        ora.run(50)
        b = ora.get_state()
        assert np.array_equal(s["occupancy"][j], b["occupancy"])
        np.testing.assert_allclose(s["features"][j], b["features"], rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(s["bias"][j], ora.get_bias(), rtol=1e-10, atol=1e-9)
    eng.close()


@pytest.mark.parametrize("with_occupancy", [True, False], ids=["occ", "no-occ"])
def test_device_ring_rows_of_a_lazy_handle(with_occupancy):
    """Rows recorded in-kernel: scalar features from the kernel, cluster features evaluated from the recorded
    occupancies when the launch is through -- also when the caller did not ask for occupancies -- over two
    pipelined blocks; then the chain continues from the right scalars."""
    c = load_case("rocksalt333_two_sublattices")
    sc = c["sc"]
    tab = tables_for("rocksalt333_two_sublattices", CORR, mu_table=_mu(sc, 3, 6))
    R = 7
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_FLIP)
    eng, ora = _pair(tab, cfg, _rand_occ(sc, np.random.default_rng(9), R), np.arange(R, dtype=np.uint64) + np.uint64(90), 2500.0)
    assert "lazy-features" in eng.kernel_info(), eng.kernel_info()
    eng.run_sampled_async(3, 40, occupancy=with_occupancy)
    eng.run_sampled_async(2, 25, occupancy=with_occupancy)
    for ns, thin in ((3, 40), (2, 25)):
        s = eng.fetch_samples()
        for j in range(ns):
            ora.run(thin)
            b = ora.get_state()
            if with_occupancy:
                assert np.array_equal(s["occupancy"][j], b["occupancy"])
            np.testing.assert_allclose(s["enthalpy"][j], b["enthalpy"], rtol=1e-10, atol=1e-8)
            np.testing.assert_allclose(s["features"][j], b["features"], rtol=1e-10, atol=1e-8)
            assert np.array_equal(s["accepted"][j].astype(bool), b["accepted"].astype(bool))
    _same_chain(eng, ora, (17, 200))
    eng.close()


def test_replayed_records_on_a_lazy_handle():
    """Records the lean replay kernels take and records they do not (three flips: the universal kernel, which
    updates the features incrementally) interleaved on one handle: the features stay those of the oracle."""
    c = load_case("rocksalt333_two_sublattices")
    sc = c["sc"]
    tab = tables_for("rocksalt333_two_sublattices", CORR, mu_table=_mu(sc, 3, 6))
    R = 3
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_FLIP)
    rng = np.random.default_rng(12)
    eng, ora = _pair(tab, cfg, _rand_occ(sc, rng, R), np.arange(R, dtype=np.uint64) + np.uint64(5), 3500.0)
    assert "lazy-features" in eng.kernel_info(), eng.kernel_info()
    nsp = np.array([sc.model.prim.nspecies[b] for b in sc.site_b])
    active = np.flatnonzero(nsp > 1)
    for nflips in (1, 3, 1, 2):
        nst = 12
        steps = -np.ones((R, nst, capi.STEP_ROW), dtype=np.int32)
        for r in range(R):
            for k in range(nst):
                sites = rng.choice(active, nflips, replace=False)
                for f, site in enumerate(sites):
                    steps[r, k, 2 * f] = site
                    steps[r, k, 2 * f + 1] = rng.integers(nsp[site])
        u = rng.random((R, nst))
        x = eng.replay(steps, u)
        y = ora.replay(steps, u)
        assert np.array_equal(np.asarray(x[0], dtype=bool), np.asarray(y[0], dtype=bool))
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        np.testing.assert_allclose(a["features"], b["features"], rtol=1e-10, atol=1e-8)
        eng.run(30)
        ora.run(30)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        np.testing.assert_allclose(a["features"], b["features"], rtol=1e-10, atol=1e-8)
    eng.close()
