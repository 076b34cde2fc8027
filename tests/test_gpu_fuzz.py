"""Randomised shape sweep: odd supercells (N not a multiple of 16 / 64), walker counts that do
not fill a workgroup, single-step and uneven launches, every kernel / step type -- GPU vs the
CPU oracle on identical Philox streams (bit-exact occupancies, 1e-10 enthalpies)."""

import numpy as np
import pytest

from smol_amd import capi, ewald, synth
from smol_amd.engine import Engine

pytestmark = pytest.mark.gpu

SHAPES = [
    (synth.fcc_prim, {2: 4.5}, [3, 3, 3], 1),
    (synth.fcc_prim, {2: 6.0, 3: 5.0}, [5, 3, 4], 5),
    (synth.fcc_prim, {2: 6.0, 3: 5.0}, [[3, 1, 0], [0, 4, 1], [1, 0, 5]], 7),
    (lambda: synth.fcc_prim(nspecies=3), {2: 5.0, 3: 3.0}, [4, 5, 3], 3),
    (lambda: synth.fcc_prim(nspecies=4), {2: 4.5}, [3, 4, 5], 9),
    (synth.rocksalt_prim, {2: 4.5, 3: 3.2}, [3, 2, 5], 6),
]


@pytest.mark.parametrize("shape", range(len(SHAPES)))
@pytest.mark.parametrize("step", [capi.STEP_FLIP, capi.STEP_SWAP], ids=["flip", "swap"])
@pytest.mark.parametrize("kernel", ["metropolis", "wang-landau"])
def test_random_shapes_match_oracle(shape, step, kernel):
    from oracle import oracle as orc

    prim_fn, cutoffs, scm, R = SHAPES[shape]
    prim = prim_fn()
    model = synth.build_cluster_model(prim, cutoffs)
    sc = synth.build_supercell(model, scm)
    rng = np.random.default_rng(100 + shape)
    is_ionic = prim.nb > 1
    ew = ewald.supercell_ewald(sc) if is_ionic else None  # (Wang-Landau too: mc_wl_kernel with the field in LDS)
    mu = None
    if step == capi.STEP_FLIP:  # semigrand, Metropolis and Wang-Landau
        nsp_max = max(prim.nspecies)
        mu = np.zeros((sc.num_sites, nsp_max))
        act = np.array([prim.nspecies[b] for b in sc.site_b]) > 1
        mu[act] = rng.uniform(-0.3, 0.3, nsp_max)[None, :]
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=shape), ewald=ew, ewald_coef=0.2,
                                   mu_table=mu)
    nsp = np.array([prim.nspecies[b] for b in sc.site_b])
    occ0 = (rng.random((R, sc.num_sites)) * nsp).astype(np.int32)
    if kernel == "metropolis":
        cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    else:
        probe = orc.OracleEvaluator(tab)
        h = np.array([probe.natural_parameters() @ probe.feature_vector(o) for o in occ0])
        # window edges NOT commensurate with the starting enthalpies: a walker that returns to
        # its initial state would otherwise sit exactly on a bin edge, where the last bit of the
        # accumulated enthalpy (summation order) decides the bin -- in the reference too
        cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, step, min_enthalpy=float(h.min()) - 3.0371,
                               max_enthalpy=float(h.max()) + 3.0113, bin_size=0.25, check_period=50)
    seeds = rng.integers(1, 2**62, size=R).astype(np.uint64)
    temps = rng.uniform(500.0, 4000.0, size=R)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    eng.set_state(occ0, seeds, temps)
    ora.set_state(occ0, seeds, temps)
    for chunk in (1, 1, 15, 17, 130):
        eng.run(chunk)
        ora.run(chunk)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(a["features"], b["features"], rtol=1e-10, atol=1e-8)
    if kernel == "wang-landau":
        wa, wb = eng.get_wl(), ora.get_wl()
        assert np.array_equal(wa["histogram"], wb["histogram"])
        assert np.array_equal(wa["occurrences"], wb["occurrences"])
        np.testing.assert_allclose(wa["entropy"], wb["entropy"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(wa["mean_features"], wb["mean_features"], rtol=1e-10, atol=1e-8)
        if shape != 2:  # (the skew cell of shape 2 is aliased: mc_kernel)
            assert eng.kernel_info().startswith("lean"), eng.kernel_info()


@pytest.mark.parametrize("scm", [[9, 9, 9], [10, 10, 10], [8, 8, 12], [6, 6, 5]],
                         ids=["729", "1000", "768", "180"])
@pytest.mark.parametrize("step", [capi.STEP_FLIP, capi.STEP_SWAP], ids=["flip", "swap"])
def test_ewald_field_sweep_shapes_match_oracle(scm, step):
    """Potential-field sweep after accepted flips (mc_lean.h field_sweep): batches of 9 / 14 groups
    of 64 entries with the last batch shifted back, batches of 4 / 1 for short rows and the ragged
    tail -- cation counts 729 (11 groups + 25), 1000 (15 + 40), 768 (12, no tail), 180 (2 + 52) -- at
    a temperature where most steps are accepted, GPU vs oracle on the same streams."""
    from oracle import oracle as orc

    prim = synth.rocksalt_prim()
    model = synth.build_cluster_model(prim, {2: 4.5})
    sc = synth.build_supercell(model, scm)
    rng = np.random.default_rng(sum(scm))
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=3), ewald=ewald.supercell_ewald(sc),
                                   ewald_coef=0.2)
    R = 3
    nsp = np.array([prim.nspecies[b] for b in sc.site_b])
    occ0 = (rng.random((R, sc.num_sites)) * nsp).astype(np.int32)
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    seeds = rng.integers(1, 2**62, size=R).astype(np.uint64)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    assert eng.kernel_info().startswith("lean ") and "field=1" in eng.kernel_info()
    eng.set_state(occ0, seeds, 2.0e5)
    ora.set_state(occ0, seeds, 2.0e5)
    for chunk in (1, 40, 160):
        eng.run(chunk)
        ora.run(chunk)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(a["features"], b["features"], rtol=1e-10, atol=1e-8)
    assert a["n_accepted"].min() > 60
    # the field itself has not drifted: running Ewald term == from-scratch evaluation
    full = eng.eval_full(a["occupancy"])
    np.testing.assert_allclose(a["features"], full, rtol=1e-9, atol=1e-8)


@pytest.mark.parametrize("scm,field", [([8, 8, 8], "hbm"), ([5, 5, 5], "hbm"), ([4, 4, 4], "lds"), ([8, 8, 8], "general")],
                         ids=["1024-hbm", "250-hbm", "128-lds", "1024-general"])
@pytest.mark.parametrize("step", [capi.STEP_FLIP, capi.STEP_SWAP], ids=["flip", "swap"])
def test_ewald_field_sweep_two_sublattices(scm, field, step, monkeypatch):
    """The same sweep with the potential field of ALL changeable sites (cations + anions) in HBM
    (multi-sublattice lean kernel, forced or chosen by size), in LDS, and in the general kernel:
    1024 entries = 16 groups (batches of 14 with the shifted last batch, of 9 for single flips),
    250 = 3 groups + 58, 128 = 2 groups (the LDS copy is only chosen for small cells)."""
    from oracle import oracle as orc

    monkeypatch.delenv("SMOLMC_FORCE_GENERAL", raising=False)
    monkeypatch.delenv("SMOLMC_MULTI_PHI_HBM", raising=False)
    if field == "hbm":
        monkeypatch.setenv("SMOLMC_MULTI_PHI_HBM", "1")
    if field == "general":
        monkeypatch.setenv("SMOLMC_FORCE_GENERAL", "1")
    prim = synth.rocksalt_prim(anion_charges=(-2.0, -1.0))
    model = synth.build_cluster_model(prim, {2: 4.5})
    sc = synth.build_supercell(model, scm)
    rng = np.random.default_rng(sum(scm) + 1)
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=5), ewald=ewald.supercell_ewald(sc),
                                   ewald_coef=0.2)
    R = 3
    nsp = np.array([prim.nspecies[b] for b in sc.site_b])
    occ0 = (rng.random((R, sc.num_sites)) * nsp).astype(np.int32)
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    seeds = rng.integers(1, 2**62, size=R).astype(np.uint64)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    info = eng.kernel_info()
    if field == "general":
        assert info.startswith("general")
    else:
        assert info.startswith("lean-multi") and ("field=2" if field == "hbm" else "field=1") in info
    eng.set_state(occ0, seeds, 2.0e5)
    ora.set_state(occ0, seeds, 2.0e5)
    for chunk in (1, 40, 160):
        eng.run(chunk)
        ora.run(chunk)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"])
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(a["features"], b["features"], rtol=1e-10, atol=1e-8)
    assert a["n_accepted"].min() > 60
    np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=1e-9, atol=1e-8)
