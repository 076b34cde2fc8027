"""The reference's serialized LiNiO2 models on the engine: tables read
with smol_amd.mson, evaluated through the C-ABI, checked against (i) the feature matrix smol
stored with the model -- correlation vectors and pymatgen's Ewald energies -- and (ii) the oracle
on the same imported tables, for the evaluator entry points and for Monte-Carlo runs on a larger
supercell whose cluster indices are regenerated from the stored orbits."""

import os

import numpy as np
import pytest

from smol_amd import capi, moca, mson
from smol_amd.engine import Engine

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CE_EWALD = os.path.join(GOLD, "lno_ce_ewald.mson.json.gz")
RTOL, ATOL = 1e-10, 1e-9


@pytest.fixture(scope="module")
def lno():
    return mson.load_mson(CE_EWALD), mson.wrangler_entries(CE_EWALD)


def test_eval_full_reproduces_the_reference_feature_matrix(lno):
    ce, entries = lno
    nc = ce.subspace.num_corr_functions
    for i, e in enumerate(entries):
        tab = ce.tables(e["supercell_matrix"], feature_mode=capi.FEATURES_CORRELATIONS)
        cell = tab.supercell
        occ = cell.occupancy_from_sites(e["species"], e["site_mapping"])
        eng = Engine(tab, capi.make_config(1))
        feats = eng.eval_full(occ[None])[0] / cell.size
        np.testing.assert_allclose(feats[:nc], ce.feature_matrix[i, :nc], rtol=0, atol=1e-10)
        assert abs(feats[nc] - ce.feature_matrix[i, nc]) < 5e-9  # pymatgen's Ewald energy per prim
        # predicted energy of the fitted expansion (what smol's ClusterExpansion.predict returns)
        np.testing.assert_allclose(feats @ eng.natural_parameters, ce.feature_matrix[i] @ ce.coefs,
                                   rtol=1e-12, atol=1e-10)
        eng.close()


@pytest.mark.parametrize("mode", ["corr", "int"])
def test_eval_delta_on_the_imported_model_equals_oracle_and_difference(lno, mode):
    from oracle import oracle as orc

    ce, entries = lno
    fmode = capi.FEATURES_CORRELATIONS if mode == "corr" else capi.FEATURES_INTERACTIONS
    e = entries[7]
    tab = ce.tables(e["supercell_matrix"], feature_mode=fmode)
    cell = tab.supercell
    occ = cell.occupancy_from_sites(e["species"], e["site_mapping"])
    eng, ora = Engine(tab, capi.make_config(1)), orc.OracleEvaluator(tab)
    rng = np.random.default_rng(1)
    P = cell.size
    for _ in range(25):
        s1, s2 = rng.choice(2 * P, 2, replace=False)
        flips = [(int(s1), int(1 - occ[s1])), (int(s2), int(1 - occ[s2]))]
        d = eng.eval_delta(occ, [flips[0] + flips[1]])[0]
        np.testing.assert_allclose(d, ora.feature_vector_change(occ, flips), rtol=RTOL, atol=ATOL)
        new = occ.copy()
        for s, c in flips:
            new[s] = c
        np.testing.assert_allclose(d, eng.eval_full(new[None])[0] - eng.eval_full(occ[None])[0],
                                   rtol=1e-8, atol=1e-8)
        occ = new


def _big_model(ce, dim=4, **kw):
    tab = ce.tables(np.diag([dim] * 3), **kw)  # 64 prims, 256 sites: indices regenerated
    cell = tab.supercell
    rng = np.random.default_rng(8)
    return tab, cell, rng


def _neutral_occupancies(cell, R, rng, n_li):
    """Li_x Ni3+_x Ni4+_(1-x) O2: as many Li+ as Ni3+ (codes: Li+ 0 / vacancy 1, Ni3+ 0 / Ni4+ 1)."""
    P = cell.size
    occ = np.zeros((R, cell.num_sites), dtype=np.int32)
    for r in range(R):
        li = rng.permutation(P)[: n_li]
        ni3 = P + rng.permutation(P)[: n_li]
        occ[r, :P] = 1
        occ[r, li] = 0
        occ[r, P:2 * P] = 1
        occ[r, ni3] = 0
    return occ


@pytest.mark.parametrize("mode", ["int", "corr"])
def test_canonical_swaps_with_ewald_match_oracle(lno, mode):
    """Two active sublattices (Li+/vacancy and Ni3+/Ni4+) + Ewald term with a vacancy species,
    canonical swaps, engine stream == oracle stream -> identical trajectories."""
    from oracle import oracle as orc

    ce, _ = lno
    fmode = capi.FEATURES_CORRELATIONS if mode == "corr" else capi.FEATURES_INTERACTIONS
    tab, cell, rng = _big_model(ce, feature_mode=fmode)
    R = 6
    occ = _neutral_occupancies(cell, R, rng, n_li=cell.size // 2)
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(50)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    for e in (eng, ora):
        e.set_state(occ, seeds, 900.0)
    for chunk in (3, 97, 400):
        eng.run(chunk)
        ora.run(chunk)
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"]), eng.kernel_info()
        assert np.array_equal(a["n_accepted"], b["n_accepted"])
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(a["features"], b["features"], rtol=RTOL, atol=1e-8)
    assert 0 < a["n_accepted"].sum() < a["n_steps"].sum()
    np.testing.assert_allclose(a["features"], eng.eval_full(a["occupancy"]), rtol=RTOL, atol=1e-8)
    # composition of each sublattice conserved
    P = cell.size
    assert np.all((a["occupancy"][:, :P] == 0).sum(axis=1) == P // 2)
    assert np.all((a["occupancy"][:, P:2 * P] == 0).sum(axis=1) == P // 2)


def test_smol_shaped_api_on_the_imported_model(lno):
    """Ensemble.from_mson -> Sampler with the charge-neutral TableFlip step whose flip table comes
    from the CompositionSpace of the model's own sublattices (Li+ + Ni3+ <-> vacancy + Ni4+):
    every sample stays charge neutral and its trace rows equal a from-scratch evaluation."""
    ce, _ = lno
    ens = moca.Ensemble.from_mson(ce, np.diag([4, 4, 4]))
    # the reference's call shape, Ensemble.from_cluster_expansion(expansion, supercell_matrix) (ensemble.py:133-217)
    same = moca.Ensemble.from_cluster_expansion(ce, np.diag([4, 4, 4]))
    np.testing.assert_array_equal(same.natural_parameters, ens.natural_parameters)
    assert [s.species for s in same.sublattices] == [s.species for s in ens.sublattices]
    assert [s.species for s in ens.sublattices] == [("Li+", "Vacancy"), ("Ni3+", "Ni4+"), ("O2-",)]
    assert len(ens.natural_parameters) == 12  # 11 orbit interactions + Ewald
    nw = 8
    sampler = moca.Sampler.from_ensemble(ens, temperature=1200.0, nwalkers=nw, step_type="table-flip",
                                         seeds=list(range(nw)))
    ft = np.asarray(sampler.mckernels[0].usher_kwargs["flip_table"])
    assert ft.shape == (1, 5) and abs(ft[0]).tolist() == [1, 1, 1, 1, 0]
    cell = ens.processor.supercell
    occ = _neutral_occupancies(cell, nw, np.random.default_rng(2), n_li=cell.size // 2)
    sampler.run(400, occ, thin_by=20)  # (without a chemical potential the cell fills up with Li within ~1e3 steps)
    # the Ensemble path hands the Ewald charges to the engine like MsonClusterExpansion.tables does:
    # same (lean, compact-Ewald) kernel, not the general one
    info = sampler._get_engine().kernel_info()
    direct = Engine(ce.tables(np.diag([4, 4, 4]), flip_table=ft[:, :4]),
                    capi.make_config(nw, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP))
    assert info.startswith("lean") and info == direct.kernel_info(), (info, direct.kernel_info())
    direct.close()
    c = sampler.samples
    occs = c.get_occupancies(flat=False)
    P = cell.size
    n_li = (occs[..., :P] == 0).sum(axis=-1)
    n_ni3 = (occs[..., P:2 * P] == 0).sum(axis=-1)
    assert np.array_equal(n_li, n_ni3)  # neutrality: every Li+ is compensated by a Ni3+
    assert len(np.unique(n_li)) > 1  # and the composition does move
    feats = c.get_feature_vectors(flat=False)
    for i in (0, 19):
        for w in (0, nw - 1):
            np.testing.assert_allclose(feats[i, w], ens.compute_feature_vector(occs[i, w]), rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(c.get_enthalpies(flat=False)[..., 0], feats @ ens.natural_parameters,
                               rtol=1e-10, atol=1e-8)


def test_ewald_field_placement_follows_residency(lno, monkeypatch):
    """Multi-sublattice kernel: for flips the potential field of the 8^3 LiNiO2 cell (8 KiB per
    walker) lives in LDS while all walkers stay resident in one round with it there (16 x CU-count
    / 2 here), in HBM beyond; canonical swaps keep it in LDS whenever it fits.  Both placements run
    the same chain."""
    import torch

    monkeypatch.delenv("SMOLMC_MULTI_PHI_HBM", raising=False)
    monkeypatch.delenv("SMOLMC_MULTI_PHI_LDS", raising=False)
    ce, _ = lno
    tab = ce.tables(np.diag([8, 8, 8]))
    cell = tab.supercell
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    small, big = 64, 16 * cus
    out = {}
    for R in (small, big):
        rng = np.random.default_rng(5)
        occ = _neutral_occupancies(cell, small, rng, n_li=cell.size // 2)
        occ = np.tile(occ, (R // small, 1))
        if R == big:
            swp = Engine(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP))
            assert "field=1" in swp.kernel_info(), swp.kernel_info()
            swp.close()
        eng = Engine(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_FLIP))
        info = eng.kernel_info()
        assert info.startswith("lean-multi") and ("field=1" if R == small else "field=2") in info, info
        seeds = np.tile(np.arange(small, dtype=np.uint64) + np.uint64(9), R // small)
        eng.set_state(occ, seeds, 1100.0)
        eng.run(300)
        st = eng.get_state()
        out[R] = (st["occupancy"][:small], st["enthalpy"][:small], st["n_accepted"][:small])
        eng.close()
    assert np.array_equal(out[small][0], out[big][0]) and np.array_equal(out[small][2], out[big][2])
    np.testing.assert_allclose(out[small][1], out[big][1], rtol=RTOL, atol=1e-8)
    assert out[small][2].sum() > 0
