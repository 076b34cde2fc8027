"""The site renumbering smolmc_create applies behind the C-ABI (engine.hip: plan_relabelling / relabel_tables, ABI 8),
checked WITHOUT a GPU through the library's host-only test hooks: with restricted sites
(smol/moca/sublattice.py:84-107) or a sublattice split by species (:109-186) the active sites of a sublattice are
scattered; the renumbered tables make every active sublattice one ascending site range in list order and are the SAME
model -- the CPU oracle walks the same chain on both table sets (occupancies mapped through the permutation), with an
Ewald term, chemical potentials and a bias.  (The engine itself is compared with the oracle on the caller's tables in
tests/test_gpu_moca.py and tests/test_gpu_capi_relabel.py.)"""

import ctypes as C
import types

import numpy as np
import pytest

from smol_amd import capi, engine, moca, synth


@pytest.fixture(scope="module")
def lib():
    import os

    import __graft_entry__ as g

    if not os.path.exists(engine.LIB_PATH):
        g.build()
    L = engine.load_library()
    L.smolmc_debug_relabel.restype = C.c_void_p
    L.smolmc_debug_relabel.argtypes = [C.POINTER(capi.smolmc_tables), C.POINTER(C.c_int32)]
    L.smolmc_debug_relabel_free.restype = None
    L.smolmc_debug_relabel_free.argtypes = [C.c_void_p]
    return L


def _relabel(lib, tab):
    new_of = np.full(tab.num_sites, -1, dtype=np.int32)
    ptr = lib.smolmc_debug_relabel(C.byref(tab.struct), new_of.ctypes.data_as(C.POINTER(C.c_int32)))
    if not ptr:
        return None, None, None
    view = types.SimpleNamespace(struct=capi.smolmc_tables.from_address(ptr), _keep_alive=tab)
    return ptr, view, new_of


def _ensemble(seed=2):
    model = synth.build_cluster_model(synth.rocksalt_prim(anion_charges=(-2.0, -1.0)), {2: 5.0, 3: 3.5})
    sc = synth.build_supercell(model, [3, 3, 2])
    ens = moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model, seed=seed), ewald_coefficient=0.2)
    ens.chemical_potentials = {sp: 0.05 * i for i, sp in enumerate(ens.species)}
    return sc, ens


def test_contiguous_sublattices_need_no_renumbering(lib):
    sc, ens = _ensemble()
    ptr, view, new_of = _relabel(lib, ens.make_tables())
    assert ptr is None


@pytest.mark.parametrize("how", ["restricted", "split"])
@pytest.mark.parametrize("step", ["flip", "swap"])
def test_renumbered_tables_are_the_same_model(lib, how, step):
    from oracle import oracle as orc

    sc, ens = _ensemble()
    rng = np.random.default_rng(0)
    nsp = np.array([sc.model.prim.nspecies[b] for b in sc.site_b])
    R = 3
    occ = (rng.random((R, sc.num_sites)) * nsp).astype(np.int32)
    if how == "restricted":
        act = np.concatenate([s.active_sites for s in ens.active_sublattices])
        ens.restrict_sites(rng.choice(act, 7, replace=False))
    else:
        ens.split_sublattice_by_species(0, occ[0], [[ens.sublattices[0].species[0]], list(ens.sublattices[0].species[1:])])
        occ[:] = occ[0]  # (a split follows ONE occupancy: every walker starts from it)
    a = ens.make_tables()
    bias = moca.SquareChargeBias(ens.sublattices, penalty=0.05)
    a.set_bias(bias.bias_type, bias._table, bias.penalty)
    ptr, b, new_of = _relabel(lib, a)
    assert ptr is not None, "scattered active sites must be renumbered"
    try:
        N = sc.num_sites
        assert sorted(new_of) == list(range(N))
        old_of = np.argsort(new_of)
        ta, tb = a.struct, b.struct
        ns = ta.n_sublattices
        ptrs = np.ctypeslib.as_array(ta.sub_site_ptr, (ns + 1,))
        sa = np.ctypeslib.as_array(ta.sub_active_sites, (ptrs[-1],))
        sb = np.ctypeslib.as_array(tb.sub_active_sites, (ptrs[-1],))
        # every active sublattice: one ascending range, list order kept, sublattices in order from site 0
        assert np.array_equal(sb, np.arange(ptrs[-1]))
        assert np.array_equal(sb, new_of[sa])
        # the other changeable sites (restricted ones) directly behind: the Ewald field's block is contiguous
        ew = np.ctypeslib.as_array(tb.ewald_inds, (N, tb.ewald_width))
        changeable = np.flatnonzero(((ew >= 0).sum(axis=1) != 1) | (ew[:, 0] < 0) | np.isin(np.arange(N), sb))
        assert np.array_equal(changeable, np.arange(len(changeable)))
        cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, moca.STEP_TYPES[step])
        oa, ob = orc.OracleMC(a, cfg), orc.OracleMC(b, cfg)
        seeds = np.arange(R, dtype=np.uint64) + 5
        oa.set_state(occ, seeds, 3000.0)
        ob.set_state(np.ascontiguousarray(occ[:, old_of]), seeds, 3000.0)
        for n in (1, 50, 300):
            oa.run(n)
            ob.run(n)
            xa, xb = oa.get_state(), ob.get_state()
            assert np.array_equal(xa["occupancy"], xb["occupancy"][:, new_of])
            np.testing.assert_allclose(xa["enthalpy"], xb["enthalpy"], rtol=1e-10, atol=1e-9)
            np.testing.assert_allclose(xa["features"], xb["features"], rtol=1e-10, atol=1e-9)
            np.testing.assert_allclose(oa.get_bias(), ob.get_bias(), rtol=1e-10, atol=1e-9)
        assert 0 < xa["n_accepted"].sum()
        # from-scratch evaluation too (full tables renamed consistently with the local ones)
        ea, eb = orc.OracleEvaluator(a), orc.OracleEvaluator(b)
        np.testing.assert_allclose(ea.feature_vector(xa["occupancy"][0]), eb.feature_vector(xb["occupancy"][0]), rtol=1e-12, atol=1e-9)
        del oa, ob, ea, eb
    finally:
        lib.smolmc_debug_relabel_free(ptr)
