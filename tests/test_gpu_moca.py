"""GPU tests of the smol.moca mirror (Sampler / Ensemble / Processors on the engine),
restating the reference's own invariants:
  tests/test_moca/test_processor.py:170-231  delta == difference, reversibility, drift
  tests/test_moca/test_kernel.py:109-170     accepted => occupancy changed, trace deltas
  tests/test_moca/test_sampler.py:59-129     trace rows == recomputed features, anneal
"""

import numpy as np
import pytest

from smol_amd import capi, moca, synth

pytestmark = pytest.mark.gpu

RTOL = 1e-12
ATOL = 2e4 * np.finfo(float).eps  # tests/test_moca/test_processor.py:27-29
DRIFT_TOL = 1e-9


@pytest.fixture(scope="module")
def fcc():
    model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [5, 5, 5])
    return model, sc, synth.random_coefs(model, seed=3)


@pytest.fixture(scope="module")
def rocksalt():
    model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 6.0, 3: 4.5})
    sc = synth.build_supercell(model, [3, 3, 3])
    return model, sc, synth.random_coefs(model, seed=4)


def _processors(fcc, rocksalt):
    model, sc, coefs = fcc
    yield moca.ClusterExpansionProcessor(sc, coefs)
    yield moca.ClusterDecompositionProcessor(sc, model.cluster_interaction_tensors(coefs))
    model, sc, coefs = rocksalt
    yield moca.EwaldProcessor(sc, coefficient=0.3)
    comp = moca.CompositeProcessor(sc)
    comp.add_processor(moca.ClusterExpansionProcessor(sc, coefs))
    comp.add_processor(moca.EwaldProcessor(sc, coefficient=0.3))
    yield comp


def _rand_occ(rng, sc):
    nsp = np.array([sc.model.prim.nspecies[b] for b in sc.site_b])
    return (rng.random(sc.num_sites) * nsp).astype(np.int32), nsp


def test_processor_delta_is_difference_and_reversible(fcc, rocksalt):
    rng = np.random.default_rng(0)
    for proc in _processors(fcc, rocksalt):
        occ, nsp = _rand_occ(rng, proc.supercell)
        active = np.flatnonzero(nsp > 1)
        prop_f = proc.compute_property(occ)
        for _ in range(15):
            s1, s2 = rng.choice(active, 2, replace=False)
            flips = [(int(s1), int((occ[s1] + 1) % nsp[s1])), (int(s2), int((occ[s2] + 1) % nsp[s2]))]
            new = occ.copy()
            for s, c in flips:
                new[s] = c
            d = proc.compute_feature_vector_change(occ, flips)
            f0, f1 = proc.compute_feature_vector(occ), proc.compute_feature_vector(new)
            np.testing.assert_allclose(d, f1 - f0, rtol=1e-8, atol=1e-8)
            rev = [(s, int(occ[s])) for s, _ in flips][::-1]
            np.testing.assert_allclose(d, -1 * np.asarray(proc.compute_feature_vector_change(new, rev)),
                                       rtol=RTOL, atol=1e-9)
            dprop = proc.compute_property_change(occ, flips)
            np.testing.assert_allclose(proc.compute_property(new) - prop_f, dprop, rtol=1e-8, atol=1e-8)
            occ, prop_f = new, proc.compute_property(new)


def test_processor_average_drift(fcc):
    model, sc, coefs = fcc
    for proc in (moca.ClusterExpansionProcessor(sc, coefs),
                 moca.ClusterDecompositionProcessor(sc, model.cluster_interaction_tensors(coefs))):
        fwd, rev = proc.compute_average_drift(iterations=60, rng=1)
        assert fwd <= DRIFT_TOL and rev <= DRIFT_TOL


def test_decomposition_and_expansion_give_same_energy(fcc):
    """ClusterDecompositionProcessor samples the same ensemble (expansion.py:249-252)."""
    model, sc, coefs = fcc
    pe = moca.ClusterExpansionProcessor(sc, coefs)
    pd = moca.ClusterDecompositionProcessor(sc, model.cluster_interaction_tensors(coefs))
    rng = np.random.default_rng(2)
    for _ in range(5):
        occ, _ = _rand_occ(rng, sc)
        np.testing.assert_allclose(pe.compute_property(occ), pd.compute_property(occ), rtol=1e-10)


@pytest.mark.parametrize("kind", ["canonical", "semigrand"])
def test_sampler_trace_rows_match_recomputed_features(fcc, kind):
    """tests/test_moca/test_sampler.py:59-84 (ATOL 5e-13 there; rel 1e-10 here)."""
    model, sc, coefs = fcc
    mu = {"A0": 0.3, "A1": -0.1} if kind == "semigrand" else None
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs, chemical_potentials=mu)
    nw = 5
    sampler = moca.Sampler.from_ensemble(ens, temperature=1500, nwalkers=nw, seeds=list(range(1, nw + 1)))
    rng = np.random.default_rng(3)
    occu = np.vstack([_rand_occ(rng, sc)[0] for _ in range(nw)])
    sampler.run(2000, occu, thin_by=50)
    c = sampler.samples
    assert c.num_samples == 40 and c.total_mc_steps == 2000
    occs = c.get_occupancies(flat=False)
    feats = c.get_feature_vectors(flat=False)
    enth = c.get_enthalpies(flat=False)
    for i in (0, 17, 39):
        for w in range(nw):
            f = ens.compute_feature_vector(occs[i, w])
            np.testing.assert_allclose(feats[i, w], f, rtol=1e-10, atol=1e-9)
            np.testing.assert_allclose(enth[i, w, 0], ens.natural_parameters @ f, rtol=1e-10, atol=1e-9)
    if kind == "canonical":  # composition conserved by swaps
        assert np.all(occs.sum(axis=-1) == occu.sum(axis=-1)[None, :])
        np.testing.assert_allclose(c.get_energies(), c.get_enthalpies())
    else:
        assert not np.all(occs.sum(axis=-1) == occu.sum(axis=-1)[None, :])
        assert np.abs(c.get_energies() - c.get_enthalpies()).max() > 0
    assert 0 < sampler.efficiency() <= 1
    np.testing.assert_allclose(c.get_temperatures(), 1500.0)
    # continuing a run takes the last sample as the start and keeps the random streams going
    sampler.run(500, thin_by=50)
    assert c.num_samples == 50 and c.total_mc_steps == 2500


def test_sampler_matches_oracle_stream(fcc):
    """Same seeds -> same Philox streams as the CPU oracle: identical sampled occupancies."""
    from oracle import oracle as orc

    model, sc, coefs = fcc
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs)
    nw = 3
    seeds = [11, 12, 13]
    sampler = moca.Sampler.from_ensemble(ens, temperature=1200, nwalkers=nw, seeds=seeds)
    rng = np.random.default_rng(8)
    occu = np.vstack([_rand_occ(rng, sc)[0] for _ in range(nw)])
    sampler.run(300, occu, thin_by=100)
    sampler.run(200, thin_by=100)  # continuation keeps the stream position
    tab = ens.make_tables()
    ora = orc.OracleMC(tab, capi.make_config(nw))
    ora.set_state(occu, np.array(seeds, dtype=np.uint64), 1200.0)
    occs = sampler.samples.get_occupancies(flat=False)
    for i in range(5):
        ora.run(100)
        st = ora.get_state()
        assert np.array_equal(occs[i], st["occupancy"])
        np.testing.assert_allclose(sampler.samples.get_enthalpies(flat=False)[i, :, 0], st["enthalpy"],
                                   rtol=1e-10, atol=1e-9)


def test_continuing_on_the_device_equals_reloading_the_last_sample(fcc):
    """run(n) + run(n) without occupancies continues from the device state (no upload); forcing
    the re-upload of the last recorded sample instead gives the same chain, and so does clearing
    the bookkeeping by running through sample()."""
    model, sc, coefs = fcc
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs)
    occu = np.vstack([_rand_occ(np.random.default_rng(21), sc)[0] for _ in range(4)])
    out = []
    for reload_each_time in (False, True):
        sampler = moca.Sampler.from_ensemble(ens, temperature=1500, nwalkers=4, seeds=[1, 2, 3, 4])
        sampler.run(300, occu, thin_by=50)
        for _ in range(3):
            if reload_each_time:
                sampler._resume_at = None
            else:
                assert sampler._resume_at == (id(sampler.samples), sampler.samples.num_samples)
            sampler.run(200, thin_by=50)
        out.append((sampler.samples.get_occupancies(flat=False), sampler.samples.get_enthalpies(flat=False),
                    sampler.samples.get_trace_value("accepted", flat=False)))
    assert out[0][0].shape == (18, 4, sc.num_sites)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][2], out[1][2])
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-11, atol=1e-9)


def test_streamed_run_equals_in_memory_run(fcc, tmp_path):
    """Sampler.run(stream_chunk, stream_file) (sampler.py:264-301): chunks go to the streaming
    directory, none stays in memory (one sample with keep_last_chunk), the next run continues
    the chain, and the directory holds exactly the in-memory run's samples."""
    model, sc, coefs = fcc
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs)
    occu = np.vstack([_rand_occ(np.random.default_rng(31), sc)[0] for _ in range(3)])
    mem = moca.Sampler.from_ensemble(ens, temperature=1400, nwalkers=3, seeds=[7, 8, 9])
    mem.run(1100, occu, thin_by=100)
    mem.run(500, thin_by=100)
    st = moca.Sampler.from_ensemble(ens, temperature=1400, nwalkers=3, seeds=[7, 8, 9])
    path = str(tmp_path / "samples")
    st.run(1100, occu, thin_by=100, stream_chunk=4, stream_file=path, keep_last_chunk=True)
    assert st.samples.num_samples == 1  # only the last sample is kept
    st.run(500, thin_by=100, stream_chunk=4, stream_file=path)
    assert st.samples.num_samples == 0
    with pytest.raises(RuntimeError):
        st.run(100, thin_by=100)  # nothing in memory to start from (as in the reference)
    back = moca.SampleContainer.from_stream(path, ens)
    assert back.num_samples == 16 and back.total_mc_steps == 1600
    assert np.array_equal(back.get_occupancies(flat=False), mem.samples.get_occupancies(flat=False))
    np.testing.assert_allclose(back.get_enthalpies(flat=False), mem.samples.get_enthalpies(flat=False),
                               rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(back.get_feature_vectors(flat=False), mem.samples.get_feature_vectors(flat=False),
                               rtol=1e-12, atol=1e-10)
    assert back.sampling_efficiency() == mem.samples.sampling_efficiency()


def test_anneal(fcc):
    """tests/test_moca/test_sampler.py:89-113."""
    model, sc, coefs = fcc
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs)
    sampler = moca.Sampler.from_ensemble(ens, temperature=5000, nwalkers=2, seeds=[5, 6])
    occu = np.vstack([_rand_occ(np.random.default_rng(4), sc)[0] for _ in range(2)])
    temps = np.linspace(2000, 500, 4)
    with pytest.raises(ValueError):
        sampler.anneal(temps[::-1], 100, occu)
    sampler.anneal(temps, 400, occu, thin_by=100)
    c = sampler.samples
    assert c.num_samples == 16
    np.testing.assert_allclose(c.get_temperatures(flat=False) if False else c.get_trace_value(
        "temperature", flat=False)[:, 0, 0], np.repeat(temps, 4))
    # cooling lowers the energy on average
    e = c.get_enthalpies(flat=False)[:, :, 0].mean(axis=1)
    assert e[-1] < e[0]


def test_wang_landau_sampler(fcc):
    model, sc, coefs = fcc
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs)
    occu = _rand_occ(np.random.default_rng(5), sc)[0]
    h0 = float(ens.natural_parameters @ ens.compute_feature_vector(occu))
    sampler = moca.Sampler.from_ensemble(
        ens, h0 - 8.0, h0 + 8.0, 0.5, kernel_type="Wang-Landau", nwalkers=2, seeds=[1, 2],
        check_period=200, flatness=0.2,
    )
    assert "entropy" in sampler.samples.traced_values
    sampler.run(4000, np.vstack([occu, occu]), thin_by=1000)
    c = sampler.samples
    ent = c.get_trace_value("entropy", flat=False)
    hist = c.get_trace_value("histogram", flat=False)
    assert ent.shape == (4, 2, 32) and (ent[-1] > 0).sum() >= 2
    assert np.all(np.diff(ent.sum(axis=-1), axis=0) > 0)  # entropy only grows
    assert np.all(hist >= 0)
    enth = c.get_enthalpies(flat=False)
    assert np.all((enth >= h0 - 8.0) & (enth < h0 + 8.0))  # never leaves the window
    feats = c.get_feature_vectors(flat=False)
    occs = c.get_occupancies(flat=False)
    np.testing.assert_allclose(feats[-1, 0], ens.compute_feature_vector(occs[-1, 0]), rtol=1e-10, atol=1e-9)


def test_replica_exchange_on_engine(fcc):
    """Config-5 style ladder on one rank: temperatures stay a permutation of the ladder, the
    coldest rungs end up lowest in enthalpy, traces stay consistent."""
    import torch

    from smol_amd import parallel
    from smol_amd.engine import Engine

    model, sc, coefs = fcc
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs)
    tab = ens.make_tables()
    R = 16
    eng = Engine(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP))
    rng = np.random.default_rng(12)
    occ = np.vstack([_rand_occ(rng, sc)[0] for _ in range(R)])
    ladder = parallel.geometric_ladder(300.0, 6000.0, R)
    eng.set_state(occ, np.arange(R, dtype=np.uint64) + np.uint64(1), ladder)
    rex = parallel.ReplicaExchange(ladder, R, seed=3)
    parallel.run_replica_exchange(eng, rex, 60, 250)
    assert sorted(rex.rung_of) == list(range(R))
    assert rex.accepted.sum() > 0
    st = eng.get_state()
    np.testing.assert_allclose(st["features"], eng.eval_full(st["occupancy"]), rtol=1e-10, atol=1e-8)
    H_by_rung = st["enthalpy"][np.argsort(rex.rung_of)]
    assert H_by_rung[:4].mean() < H_by_rung[-4:].mean()  # cold rungs sit lower


def test_accumulated_drift_stays_at_rounding_level(fcc):
    """SURVEY 8f rank 3 (drift audit): after 2e5 swap steps per walker the running features /
    enthalpy differ from a from-scratch evaluation by ~1e-11 absolute (extensive values ~1e2)."""
    from smol_amd.engine import Engine

    model, sc, coefs = fcc
    tab = moca.Ensemble.from_cluster_expansion(sc, coefs).make_tables()
    R = 32
    eng = Engine(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP))
    rng = np.random.default_rng(1)
    occ = np.vstack([_rand_occ(rng, sc)[0] for _ in range(R)])
    eng.set_state(occ, np.arange(R, dtype=np.uint64) + np.uint64(5), 2000.0)
    assert max(eng.audit_drift()) < 1e-9
    eng.run(200_000)
    df, dh = eng.audit_drift()
    assert df < 1e-8 and dh < 1e-8


def test_uniformly_random_kernel(fcc):
    """UniformlyRandom (smol/moca/kernel/random.py:16-38): every proposed step is accepted -- the
    engine's Metropolis kernel at beta = 0 -- and the trace stays consistent."""
    model, sc, coefs = fcc
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs)
    sampler = moca.Sampler.from_ensemble(ens, kernel_type="UniformlyRandom", nwalkers=3, seeds=[4, 5, 6])
    occu = np.vstack([_rand_occ(np.random.default_rng(8), sc)[0] for _ in range(3)])
    sampler.run(600, occu, thin_by=100)
    c = sampler.samples
    assert "temperature" not in c.traced_values
    st = sampler.engine.get_state()
    assert np.array_equal(st["n_accepted"], st["n_steps"]) and st["n_steps"][0] == 600
    assert c.sampling_efficiency() == 1.0
    occs, feats = c.get_occupancies(flat=False), c.get_feature_vectors(flat=False)
    assert np.array_equal(occs.sum(axis=-1), np.broadcast_to(occu.sum(axis=-1), occs.shape[:2]))  # swaps conserve
    np.testing.assert_allclose(feats[-1, 1], ens.compute_feature_vector(occs[-1, 1]), rtol=1e-10, atol=1e-9)


def _wl_sampler(ens, h0, **kw):
    return moca.Sampler.from_ensemble(ens, h0 - 8.0, h0 + 8.0, 0.5, kernel_type="Wang-Landau", nwalkers=3,
                                      seeds=[1, 2, 3], check_period=150, flatness=0.2, **kw)


def test_wang_landau_callable_mod_update(fcc):
    """mod_update may be any callable (wanglandau.py:100-105): it is applied by the host at every
    flatness check; with m -> m / 2 the run must equal the device's own divide-by-2 exactly."""
    model, sc, coefs = fcc
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs)
    occu = _rand_occ(np.random.default_rng(5), sc)[0]
    h0 = float(ens.natural_parameters @ ens.compute_feature_vector(occu))
    start = np.vstack([occu] * 3)
    dev = _wl_sampler(ens, h0)
    dev.run(3000, start, thin_by=500)
    host = _wl_sampler(ens, h0, mod_update=lambda m: m / 2.0)
    host.run(3000, start, thin_by=500)
    m = dev.samples.get_trace_value("mod_factor", flat=False)
    assert (m[-1] < 1.0).all()  # the histograms did get flat
    for name in ("entropy", "histogram", "occurrences", "mod_factor", "occupancy"):
        assert np.array_equal(dev.samples.get_trace_value(name, flat=False), host.samples.get_trace_value(name, flat=False)), name
    # a different schedule: m -> m ** 0.5 wherever a check passes
    root = _wl_sampler(ens, h0, mod_factor=0.5, mod_update=lambda m: m ** 0.5)
    root.run(1500, start, thin_by=500)
    mr = root.samples.get_trace_value("mod_factor", flat=False)[-1, :, 0]
    assert all(any(np.isclose(x, 0.5 ** (0.5 ** k)) for k in range(12)) for x in mr) and (mr > 0.5).any()


def test_wang_landau_resumes_in_a_fresh_engine(fcc):
    """aux_checkpoint / restore_aux (smolmc_set_wl + smolmc_set_counters): a Wang-Landau run split
    over two engines (as over two processes) equals the uninterrupted run bit for bit."""
    model, sc, coefs = fcc
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs)
    occu = _rand_occ(np.random.default_rng(5), sc)[0]
    h0 = float(ens.natural_parameters @ ens.compute_feature_vector(occu))
    start = np.vstack([occu] * 3)
    whole = _wl_sampler(ens, h0)
    whole.run(2400, start, thin_by=1200)
    first = _wl_sampler(ens, h0)
    first.run(1200, start, thin_by=1200)
    ck = first.aux_checkpoint()
    first.engine.close()
    second = _wl_sampler(ens, h0)
    second._engine = None  # a fresh handle, as a new process would have
    second.restore_aux(ck)
    second.run(1200, ck["occupancy"], thin_by=1200)
    a, b = whole.engine.get_state(), second.engine.get_state()
    assert np.array_equal(a["occupancy"], b["occupancy"]) and np.array_equal(a["n_accepted"], b["n_accepted"])
    assert np.array_equal(a["n_steps"], b["n_steps"])
    wa, wb = whole.engine.get_wl(), second.engine.get_wl()
    for k in ("entropy", "histogram", "occurrences", "mod_factor"):
        assert np.array_equal(wa[k], wb[k]), k
    np.testing.assert_allclose(wa["mean_features"], wb["mean_features"], rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=1e-12, atol=1e-9)


def _neutral(sc, n_ti, rng):
    P = sc.size
    n_mn = (P - 3 * n_ti) // 2
    occ = np.zeros(sc.num_sites, dtype=np.int32)
    perm = rng.permutation(P)
    occ[perm[:n_mn]] = 1
    occ[perm[n_mn:n_mn + n_ti]] = 2
    return occ


def test_any_usher_with_any_kernel_through_the_sampler(rocksalt):
    """kernel/base.py:192-239 composes any usher with any kernel and bias: TableFlip under
    Wang-Landau and TableFlip with a FugacityBias through the smol-shaped Sampler (both refused
    before round 4; the universal kernel runs them).  Trace rows == recomputed features, the
    composition stays on the charge-neutral line, Wang-Landau never leaves its window."""
    model, sc, coefs = rocksalt
    comp = moca.CompositeProcessor(sc)
    comp.add_processor(moca.ClusterExpansionProcessor(sc, coefs))
    comp.add_processor(moca.EwaldProcessor(sc, coefficient=0.1))
    ens = moca.Ensemble(comp, chemical_potentials={"Li+": 0.1, "Mn3+": -0.2, "Ti4+": 0.05})
    rng = np.random.default_rng(3)
    occ = np.array([_neutral(sc, 3, rng), _neutral(sc, 5, rng)])
    table = [[1, -3, 2, 0]]  # 3 Mn3+ <-> Li+ + 2 Ti4+ (the O2- column is always zero)
    h0 = np.array([ens.natural_parameters @ ens.compute_feature_vector(o) for o in occ])
    wl = moca.Sampler.from_ensemble(ens, float(h0.min()) - 30.3, float(h0.max()) + 30.1, 0.5, kernel_type="Wang-Landau",
                                    step_type="table-flip", nwalkers=2, seeds=[5, 6], flip_table=table,
                                    swap_weight=0.2, check_period=100)
    wl.run(3000, occ, thin_by=500)
    assert wl._engine.kernel_info().startswith("universal")
    c = wl.samples
    occs, feats, enth = c.get_occupancies(flat=False), c.get_feature_vectors(flat=False), c.get_enthalpies(flat=False)
    assert np.all((enth >= h0.min() - 30.3) & (enth < h0.max() + 30.1))
    for w in range(2):
        np.testing.assert_allclose(feats[-1, w], ens.compute_feature_vector(occs[-1, w]), rtol=1e-10, atol=1e-8)
        n = np.bincount(occs[-1, w][: sc.size], minlength=3)
        assert n[0] + 3 * n[1] + 4 * n[2] == 2 * sc.size  # charge neutral against the O2- sublattice
    assert (c.get_trace_value("entropy", flat=False)[-1] > 0).sum() >= 2
    assert len(np.unique(occs[:, 0], axis=0)) > 1

    fug = moca.Sampler.from_ensemble(ens, temperature=4000.0, step_type="table-flip", nwalkers=2, seeds=[7, 8],
                                     flip_table=table, swap_weight=0.2, bias_type="fugacity",
                                     bias_kwargs={"fugacity_fractions": [{"Li+": 0.2, "Mn3+": 0.3, "Ti4+": 0.5}]})
    fug.run(2000, occ, thin_by=500)
    assert fug._engine.kernel_info().startswith("lean ")  # (TableFlip + bias: mc_table_kernel<..., BIAS> since round 6)
    c = fug.samples
    b = c.get_trace_value("bias", flat=False)
    bias = fug.mckernels[0].bias
    occs = c.get_occupancies(flat=False)
    for i in range(len(occs)):
        np.testing.assert_allclose(b[i, :, 0], [bias.compute_bias(o) for o in occs[i]], rtol=1e-10, atol=1e-9)
    st = fug._engine.get_state()
    assert 0 < st["n_accepted"].sum() < st["n_steps"].sum()


def test_processor_change_of_a_whole_table_step(rocksalt):
    """Processor.compute_feature_vector_change with more than two flips: one smolmc_eval_delta
    record (ABI v6) instead of a chain of pairs -- equal to the difference of full vectors."""
    model, sc, coefs = rocksalt
    comp = moca.CompositeProcessor(sc)
    comp.add_processor(moca.ClusterExpansionProcessor(sc, coefs))
    comp.add_processor(moca.EwaldProcessor(sc, coefficient=0.3))
    rng = np.random.default_rng(9)
    occ, nsp = _rand_occ(rng, sc)
    for nflips in (3, 5, 8, 11):
        sites = rng.choice(sc.size, nflips, replace=False)
        flips = [(int(s), int((occ[s] + 1) % 3)) for s in sites]
        new = occ.copy()
        for s, cde in flips:
            new[s] = cde
        d = comp.compute_feature_vector_change(occ, flips)
        np.testing.assert_allclose(d, comp.compute_feature_vector(new) - comp.compute_feature_vector(occ),
                                   rtol=1e-9, atol=1e-8)


def test_eval_delta_spellings_are_explicit(rocksalt):
    """Engine.eval_delta: a list of (site, code) tuples is ONE step (the ushers' form, mcusher.py:104-116), rows of
    records are one step each, and the (n, 2) ndarray that can be read both ways is refused until `single_step` says
    which (round 4 read it as one step, round 5 as n steps: same call, other numbers, no error)."""
    from smol_amd.engine import Engine

    model, sc, coefs = rocksalt
    tab = capi.TableSet.from_synth(sc, coefs)
    eng = Engine(tab, capi.make_config(1))
    rng = np.random.default_rng(4)
    occ = np.zeros(sc.num_sites, dtype=np.int32)
    occ[: sc.size] = rng.integers(0, 3, sc.size)
    s0, s1 = (int(x) for x in rng.choice(sc.size, 2, replace=False))
    pairs = [(s0, int((occ[s0] + 1) % 3)), (s1, int((occ[s1] + 2) % 3))]
    arr = np.array(pairs, dtype=np.int32)
    new = occ.copy()
    for s, c in pairs:
        new[s] = c
    step = eng.eval_full(new[None])[0] - eng.eval_full(occ[None])[0]
    one = eng.eval_delta(occ, pairs)  # list of tuples: one two-flip step
    assert one.shape == (1, eng.F)
    np.testing.assert_allclose(one[0], step, rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(eng.eval_delta(occ, arr.ravel())[0], one[0], rtol=0, atol=0)  # a flat record: the same step
    np.testing.assert_allclose(eng.eval_delta(occ, arr, single_step=True)[0], one[0], rtol=0, atol=0)
    with pytest.raises(ValueError, match="ambiguous"):
        eng.eval_delta(occ, arr)
    two = eng.eval_delta(occ, arr, single_step=False)  # two single-flip steps, each against `occ`
    assert two.shape == (2, eng.F)
    for k, (s, c) in enumerate(pairs):
        x = occ.copy()
        x[s] = c
        np.testing.assert_allclose(two[k], eng.eval_full(x[None])[0] - eng.eval_full(occ[None])[0], rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(eng.eval_delta(occ, pairs, single_step=False), two, rtol=0, atol=0)
    eng.close()


def test_kernel_single_step_walks_the_sampler_chain(fcc):
    """MCKernel.single_step / compute_initial_trace / set_aux_state / trace (kernel/base.py:145-166,
    287-289,345-366; tests/test_moca/test_kernel.py test_single_step): a kernel stepped one step at
    a time is walker 0 of a Sampler with the same seed -- same occupancies after every step -- the
    occupancy is updated in place exactly when the step is accepted, and delta_trace holds the
    change of the traced values."""
    model, sc, coefs = fcc
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs)
    ens.chemical_potentials = {"A0": 0.05, "A1": -0.05}
    occu0 = _rand_occ(np.random.default_rng(21), sc)[0]
    kernel = moca.Metropolis(ens, "flip", 900.0, seed=77)
    tr0 = kernel.compute_initial_trace(occu0)
    assert tr0.accepted.tolist() == [True] and tr0.enthalpy.shape == (1,)
    np.testing.assert_allclose(tr0.features, ens.compute_feature_vector(occu0), rtol=1e-12, atol=1e-10)
    sampler = moca.Sampler.from_ensemble(ens, temperature=900.0, step_type="flip", nwalkers=1, seeds=[77])
    sampler.run(60, occu0[None], thin_by=1)
    ref = sampler.samples.get_occupancies(flat=False)[:, 0]
    acc = sampler.samples.get_trace_value("accepted", flat=False)[:, 0, 0]
    occu = occu0.copy()
    n_acc = 0
    for i in range(60):
        prev = occu.copy()
        feats = ens.compute_feature_vector(prev)
        tr = kernel.single_step(occu)
        assert tr.occupancy is occu and np.array_equal(occu, ref[i])
        assert bool(tr.accepted[0]) == bool(acc[i]) == (not np.array_equal(prev, occu))
        n_acc += bool(tr.accepted[0])
        np.testing.assert_allclose(tr.delta_trace.features, ens.compute_feature_vector(occu) - feats,
                                   rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(tr.delta_trace.enthalpy, ens.natural_parameters @ tr.delta_trace.features,
                                   rtol=1e-9, atol=1e-9)
        assert tr.temperature.tolist() == [900.0]
    assert 0 < n_acc < 60 and kernel.trace is tr
    assert "delta_trace" not in tr.names
    # a kernel owned by a sampler reports its walker's trace
    k = sampler.mckernels[0]
    np.testing.assert_array_equal(k.trace.occupancy, ref[-1])


def test_wang_landau_kernel_accessors(fcc):
    """WangLandau.levels / entropy / dos / histogram / mod_factor (wanglandau.py:150-173) of a
    kernel inside a Sampler are its walker's rows of the device state, on visited levels only."""
    model, sc, coefs = fcc
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs)
    occu = _rand_occ(np.random.default_rng(5), sc)[0]
    h0 = float(ens.natural_parameters @ ens.compute_feature_vector(occu))
    sampler = moca.Sampler.from_ensemble(ens, h0 - 8.0, h0 + 8.0, 0.5, kernel_type="Wang-Landau", nwalkers=2,
                                         seeds=[1, 2], check_period=200, flatness=0.2)
    sampler.run(3000, np.vstack([occu, occu]), thin_by=1500)
    c = sampler.samples
    ent = c.get_trace_value("entropy", flat=False)[-1]
    hist = c.get_trace_value("histogram", flat=False)[-1]
    m = c.get_trace_value("mod_factor", flat=False)[-1]
    all_levels = np.arange(h0 - 8.0, h0 + 8.0, 0.5)
    for w, k in enumerate(sampler.mckernels):
        seen = ent[w] > 0
        assert seen.sum() >= 2
        np.testing.assert_array_equal(k.entropy, ent[w][seen])
        np.testing.assert_array_equal(k.histogram, hist[w][seen])
        np.testing.assert_allclose(k.levels, all_levels[seen])
        np.testing.assert_allclose(k.dos, np.exp(ent[w][seen] - ent[w][seen].min()))
        assert k.mod_factor == m[w, 0] and k.dos.min() == 1.0
    # a lone kernel stepped by hand keeps its own Wang-Landau state
    k = moca.WangLandau(ens, "flip", h0 - 8.0, h0 + 8.0, 0.5, seed=9, check_period=50)
    o = occu.copy()
    for _ in range(40):
        tr = k.single_step(o)
    assert tr.histogram.sum() == 40 and k.histogram.sum() == 40 and len(k.levels) == len(k.dos) >= 1
    np.testing.assert_allclose(tr.features, ens.compute_feature_vector(o), rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("step", ["swap", "flip"])
def test_split_sublattice_sampling_matches_the_oracle(rocksalt, step):
    """Ensemble.split_sublattice_by_species (ensemble.py:288-321): after splitting the cation
    sublattice into {Li+, Ti4+} and {Mn3+} by occupancy, steps only mix Li+ and Ti4+ (codes 0 and 2:
    an encoding with a gap), the Mn3+ sites never change, and the GPU chain equals the oracle's."""
    from oracle import oracle as orc

    model, sc, coefs = rocksalt
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs, processor_type="expansion")
    cation = next(i for i, s in enumerate(ens.sublattices) if len(s.species) == 3)
    sub = ens.sublattices[cation]
    rng = np.random.default_rng(31)
    nw = 3
    occ = np.zeros((nw, sc.num_sites), dtype=np.int32)
    occ[:, sub.sites] = rng.integers(0, 3, len(sub.sites))[None]  # same partition for every walker
    for w in range(1, nw):  # ... but the Li/Ti arrangement differs
        mix = sub.sites[occ[0, sub.sites] != 1]
        occ[w, mix] = rng.permutation(occ[0, mix])
    if step == "flip":
        ens.chemical_potentials = {sp: 0.02 * i for i, sp in enumerate(ens.species)}
    ens.split_sublattice_by_species(cation, occ[0], [[0, 2], [1]])
    assert [s.encoding.tolist() for s in ens.sublattices[cation:cation + 2]] == [[0, 2], [1]]
    seeds = [3, 4, 5]
    sampler = moca.Sampler.from_ensemble(ens, temperature=2500.0, step_type=step, nwalkers=nw, seeds=seeds)
    sampler.run(400, occ, thin_by=100)
    # canonical swaps never draw a species code, so the lean families take ANY code list under swaps (round 6; the
    # scattered sites of the part are renumbered behind the C-ABI); flips over a code list with a gap keep mc_kernel
    info = sampler.engine.kernel_info()
    assert info.startswith("lean" if step == "swap" else "general"), info
    assert "relabelled=1" in info
    occs = sampler.samples.get_occupancies(flat=False)
    mn = sub.sites[occ[0, sub.sites] == 1]
    assert np.all(occs[:, :, mn] == 1) and np.all(np.isin(occs[:, :, ens.sublattices[cation].sites], [0, 2]))
    assert len(np.unique(occs[:, 0], axis=0)) > 1
    ora = orc.OracleMC(ens.make_tables(), capi.make_config(nw, step_type=moca.STEP_TYPES[step]))
    ora.set_state(occ, np.array(seeds, dtype=np.uint64), 2500.0)
    for i in range(4):
        ora.run(100)
        st = ora.get_state()
        assert np.array_equal(occs[i], st["occupancy"])
        np.testing.assert_allclose(sampler.samples.get_enthalpies(flat=False)[i, :, 0], st["enthalpy"],
                                   rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("step", ["swap", "flip", "table-flip"])
def test_restricted_sites_stay_on_the_specialised_kernels(rocksalt, step):
    """Ensemble.restrict_sites (sublattice.py:84-107) scatters the active sites; smolmc_create renumbers the
    sites behind the C-ABI (ABI 8) so that the lean kernels still take the model.  Checked against the oracle on
    the same tables in the caller's numbering: the restricted sites never change, the chain is the same."""
    from oracle import oracle as orc

    model, sc, coefs = rocksalt
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs, ewald_coefficient=0.15)
    rng = np.random.default_rng(41)
    cations = ens.sublattices[0]
    frozen = rng.choice(cations.sites, 6, replace=False)
    ens.restrict_sites(frozen)
    if step == "flip":
        ens.chemical_potentials = {sp: 0.04 * i for i, sp in enumerate(ens.species)}
    nw = 4
    occ = np.array([_neutral(sc, 2 + w, rng) for w in range(nw)])
    kw = dict(flip_table=[[1, -3, 2, 0]], swap_weight=0.2) if step == "table-flip" else {}
    seeds = [21, 22, 23, 24]
    sampler = moca.Sampler.from_ensemble(ens, temperature=4000.0, step_type=step, nwalkers=nw, seeds=seeds, **kw)
    sampler.run(600, occ, thin_by=150)
    assert sampler.engine.kernel_info().startswith("lean"), sampler.engine.kernel_info()
    assert "relabelled=1" in sampler.engine.kernel_info()
    occs = sampler.samples.get_occupancies(flat=False)
    assert np.all(occs[:, :, frozen] == occ[None, :, frozen]) and len(np.unique(occs[:, 0], axis=0)) > 1
    ukw = {k: v for k, v in kw.items()}
    ora = orc.OracleMC(ens.make_tables(**ukw), capi.make_config(nw, step_type=moca.STEP_TYPES[step]))
    ora.set_state(occ, np.array(seeds, dtype=np.uint64), 4000.0)
    for i in range(4):
        ora.run(150)
        st = ora.get_state()
        assert np.array_equal(occs[i], st["occupancy"])
        np.testing.assert_allclose(sampler.samples.get_enthalpies(flat=False)[i, :, 0], st["enthalpy"], rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(sampler.samples.get_feature_vectors(flat=False)[i], st["features"], rtol=1e-10, atol=1e-8)
    # evaluator calls keep the caller's numbering too
    np.testing.assert_allclose(ens.compute_feature_vector(occs[-1, 0]), st["features"][0], rtol=1e-10, atol=1e-8)
