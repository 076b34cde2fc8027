"""The reference-side half of the boundary (smol_amd.bridge / tools/export_smol_model.py) run
against stand-ins for smol's Python objects that hold the reference's REAL compiled classes.

smol.moca cannot be imported here (pymatgen / monty absent), but everything the exporter reads
from a processor is either a plain attribute or lives inside a Cython ``ClusterSpaceEvaluator``.
The evaluators below are instances of the reference's own extension type, built out of tree from
/root/reference exactly as tests/golden/make_golden.py does, so attribute visibility (the orbit
tuples are a private ``cdef tuple``) is the reference's.  Runs only where /root/reference exists
(the build container); the GPU box never needs it."""

import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest

from smol_amd import bridge, capi, ewald, io, synth

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "smol", "utils", "cluster")),
                                reason="needs the reference checkout (build container only)")


@pytest.fixture(scope="module")
def core():
    pytest.importorskip("Cython")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden

    return make_golden.build_reference_core()


def _smol_like_ensemble(core, decomposition, with_ewald, with_mu):
    """Objects shaped like smol's, assembled the way its constructors do
    (processor/expansion.py:104-156, :318-389; processor/ewald.py:76-101; composite.py)."""
    ev, ct = core[0], core[1]
    model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 6.0, 3: 4.5})
    sc = synth.build_supercell(model, [3, 3, 3])
    coefs = synth.random_coefs(model, seed=9)
    od = model.orbit_data()
    itens = model.cluster_interaction_tensors(coefs)
    flat_it = tuple(np.ravel(t) for t in itens[1:])
    subspace = SimpleNamespace(orbits=model.orbits, num_orbits=model.num_orbits,
                               num_corr_functions=model.num_corr_functions)
    by_site = {}
    for site, data in sc.local_tables().items():
        local_od = tuple(od[pos] for pos, _, _ in data)
        args = (local_od, model.num_orbits, model.num_corr_functions)
        if decomposition:
            args += (1, itens[0], tuple(flat_it[pos] for pos, _, _ in data))
        rows = tuple(r for _, r, _ in data)
        by_site[site] = SimpleNamespace(
            site_index=site, evaluator=ev.ClusterSpaceEvaluator(*args),
            indices=SimpleNamespace(arrays=rows, container=ct.IntArray2DContainer(rows)),
            cluster_ratio=np.array([r for _, _, r in data]))
    full = tuple(sc.full_indices)
    ce = SimpleNamespace(cluster_subspace=subspace, num_sites=sc.num_sites, size=sc.size,
                         _indices=SimpleNamespace(arrays=full, container=ct.IntArray2DContainer(full)),
                         _eval_data_by_sites=by_site,
                         coefs=model.orbit_multiplicities.astype(float) if decomposition else coefs)
    if decomposition:
        ce._interaction_tensors = itens
    processor, ew_tab = ce, None
    if with_ewald:
        ew_tab = ewald.supercell_ewald(sc)
        charges = capi.TableSet._synth_charges(sc, ew_tab, "auto")
        ewp = SimpleNamespace(_ewald_inds=ew_tab[0], ewald_matrix=ew_tab[1], coefs=np.array(0.25),
                              _ewald_structure=[SimpleNamespace(specie=SimpleNamespace(oxi_state=q)) for q in charges])
        processor = SimpleNamespace(processors=[ce, ewp])
    mu = None
    if with_mu:
        mu = np.zeros((sc.num_sites, 3))
        mu[: sc.size] = [0.1, -0.2, 0.4]
    subl = SimpleNamespace(active_sites=np.arange(sc.size), encoding=np.arange(3, dtype=np.int32))
    ensemble = SimpleNamespace(processor=processor, active_sublattices=[subl],
                               _chemical_potentials=None if mu is None else {"table": mu})
    want = capi.TableSet.from_synth(
        sc, coefs, feature_mode=capi.FEATURES_INTERACTIONS if decomposition else capi.FEATURES_CORRELATIONS,
        ewald=ew_tab, ewald_coef=0.25, mu_table=mu)
    return ensemble, want, by_site


@pytest.mark.parametrize("decomposition,with_ewald,with_mu", [(True, True, True), (False, False, False),
                                                               (True, False, True), (False, True, False)])
def test_bridge_flattens_reference_objects_into_the_same_tables(core, decomposition, with_ewald, with_mu, tmp_path):
    ensemble, want, by_site = _smol_like_ensemble(core, decomposition, with_ewald, with_mu)
    some = next(iter(by_site.values())).evaluator
    # the reference keeps the evaluator's orbit tuples private (container.pxd:22): the exporter
    # of round 1 read `_orbit_data` and could never have run
    assert not hasattr(some, "_orbit_data")
    assert type(some).__module__ == "smol.utils.cluster.evaluator"
    got = bridge.tables_from_ensemble(ensemble)
    assert set(got._keep) == set(want._keep)
    # a ClusterExpansionProcessor carries no interaction tensors (the evaluator's default, the sum
    # over bit combos, evaluator.pyx:58-59, is what the bridge passes on); they are not used in
    # correlation mode
    skip = set() if decomposition else {"interaction_tensors"}
    for k, v in want._keep.items():
        if k not in skip:
            np.testing.assert_array_equal(got._keep[k], v, err_msg=k)
    for f, _ in capi.smolmc_tables._fields_:
        a, b = getattr(want.struct, f), getattr(got.struct, f)
        # (from_synth also counts the species of inactive sites; offset belongs to the tensors)
        if isinstance(a, (int, float)) and f != "max_species" and (decomposition or f != "offset"):
            assert a == b, f
    np.testing.assert_array_equal(got.natural_parameters, want.natural_parameters)
    # and the file route: tools/export_smol_model.py -> smol_amd.io.load_tables
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import export_smol_model

    path = str(tmp_path / "model.npz")
    export_smol_model.export_ensemble(ensemble, path)
    back = io.load_tables(path)
    for k, v in want._keep.items():
        if k not in skip:
            np.testing.assert_array_equal(back._keep[k], v, err_msg=k)


def test_unsupported_compositions_are_refused(core):
    ensemble, _, _ = _smol_like_ensemble(core, True, True, False)
    ensemble.processor.processors.append(SimpleNamespace(some_other_processor=True))
    with pytest.raises(NotImplementedError):
        bridge.tables_from_ensemble(ensemble)
