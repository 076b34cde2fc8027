"""Pin the CPU oracle against outputs of the reference's compiled Cython core.

Fixtures: tests/golden/*.npz, produced by tests/golden/make_golden.py which drove
smol/utils/cluster/{evaluator,ewald,correlations}.pyx (built out of tree).
Tolerances: the reference's own (tests/test_moca/test_processor.py:27-29):
rtol 1e-12, atol 2e4*eps -- the reference is built with -ffast-math so float
bit-exactness is not a property it has; occupancies / accept masks are bit-exact.
"""

import numpy as np
import pytest

from oracle import oracle as orc
from smol_amd import capi
from tests.cases import CASES, flips_of, load_case, tables_for

RTOL = 1e-12
ATOL = 2e4 * np.finfo(float).eps


@pytest.mark.parametrize("name", list(CASES))
def test_full_vectors_match_reference_core(name):
    c = load_case(name)
    g = c["gold"]
    tc = tables_for(name, capi.FEATURES_CORRELATIONS)
    ti = tables_for(name, capi.FEATURES_INTERACTIONS)
    ec, ei = orc.OracleEvaluator(tc), orc.OracleEvaluator(ti)
    nc, no = c["model"].num_corr_functions, c["model"].num_orbits
    for k, occ in enumerate(g["occ"]):
        fc = ec.feature_vector(occ)
        fi = ei.feature_vector(occ)
        np.testing.assert_allclose(fc[:nc], g["full_corr"][k], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(fc[:nc], g["full_corr_legacy"][k], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(fi[:no], g["full_int"][k], rtol=RTOL, atol=ATOL)
        # evaluator level is intensive (evaluator.pyx:165)
        np.testing.assert_allclose(ec.correlations(occ) * c["sc"].size, g["full_corr"][k],
                                   rtol=RTOL, atol=ATOL)
        if c["ewald"] is not None:
            np.testing.assert_allclose(fc[nc], g["full_ewald"][k], rtol=1e-11)
            np.testing.assert_allclose(fi[no], g["full_ewald"][k], rtol=1e-11)


@pytest.mark.parametrize("name", list(CASES))
def test_deltas_match_reference_core(name):
    c = load_case(name)
    g = c["gold"]
    tc = tables_for(name, capi.FEATURES_CORRELATIONS)
    ti = tables_for(name, capi.FEATURES_INTERACTIONS)
    ec, ei = orc.OracleEvaluator(tc), orc.OracleEvaluator(ti)
    nc, no = c["model"].num_corr_functions, c["model"].num_orbits
    nper = len(g["flips"]) // len(g["occ"])
    for k, row in enumerate(g["flips"]):
        occ = g["occ"][k // nper]
        fl = flips_of(row)
        dc = ec.feature_vector_change(occ, fl)
        di = ei.feature_vector_change(occ, fl)
        np.testing.assert_allclose(dc[:nc], g["delta_corr"][k], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(dc[:nc], g["delta_corr_legacy"][k], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(di[:no], g["delta_int"][k], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(di[:no], g["delta_int_legacy"][k], rtol=RTOL, atol=ATOL)
        if c["ewald"] is not None:
            np.testing.assert_allclose(dc[nc], g["delta_ewald"][k], rtol=1e-12, atol=1e-10)
            np.testing.assert_allclose(di[no], g["delta_ewald_legacy"][k], rtol=1e-12, atol=1e-10)


@pytest.mark.parametrize("name", ["fcc_prim666_triplets", "rocksalt444_ewald", "fcc3_indicator_skew"])
@pytest.mark.parametrize("mode", [capi.FEATURES_CORRELATIONS, capi.FEATURES_INTERACTIONS])
def test_delta_is_difference_and_reversible(name, mode):
    """tests/test_moca/test_processor.py:175-231 restated for the oracle."""
    c = load_case(name)
    g = c["gold"]
    e = orc.OracleEvaluator(tables_for(name, mode))
    occ = g["occ"][0].copy()
    f0 = e.feature_vector(occ)
    for row in g["flips"][:60]:
        fl = flips_of(row)
        new = occ.copy()
        for s, code in fl:
            new[s] = code
        d = e.feature_vector_change(occ, fl)
        f1 = e.feature_vector(new)
        np.testing.assert_allclose(d, f1 - f0, rtol=1e-9, atol=1e-9)
        rev = [(s, int(occ[s])) for s, _ in fl][::-1]
        dr = e.feature_vector_change(new, rev)
        np.testing.assert_allclose(d, -dr, rtol=RTOL, atol=1e-9)


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    assert orc.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    f = 0xFFFFFFFF
    assert orc.philox([f, f, f, f], [f, f]) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert orc.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_wrong_dtype_raises_valueerror():
    """The reference raises ValueError('Buffer dtype mismatch...') for int64 occupancies."""
    e = orc.OracleEvaluator(tables_for("fcc_prim222_aliased", capi.FEATURES_CORRELATIONS))
    with pytest.raises(ValueError):
        e.feature_vector(np.zeros(8, dtype=np.int64))
