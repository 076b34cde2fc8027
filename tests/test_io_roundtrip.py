"""The table wire format (smol_amd.io): a model written to .npz and read back is the same
model -- every array of smolmc_tables identical."""

import numpy as np
import pytest

from smol_amd import capi, io
from tests.cases import tables_for


@pytest.mark.parametrize("name,mode,mu", [
    ("fcc_conv444_pairs", capi.FEATURES_INTERACTIONS, False),
    ("rocksalt444_ewald", capi.FEATURES_CORRELATIONS, True),
    ("fcc_prim222_aliased", capi.FEATURES_INTERACTIONS, False),
])
def test_save_load_roundtrip(name, mode, mu, tmp_path):
    mu_table = None
    tab = tables_for(name, mode)
    if mu:
        mu_table = np.random.default_rng(0).normal(size=(tab.num_sites, 3))
        tab = tables_for(name, mode, mu_table=mu_table)
    path = str(tmp_path / "model.npz")
    io.save_tables(path, tab)
    back = io.load_tables(path)
    assert set(back._keep) == set(tab._keep)
    for k, v in tab._keep.items():
        assert back._keep[k].dtype == v.dtype, k
        np.testing.assert_array_equal(back._keep[k], v, err_msg=k)
    for f, _ in capi.smolmc_tables._fields_:
        a, b = getattr(tab.struct, f), getattr(back.struct, f)
        if isinstance(a, (int, float)):
            assert a == b, f
    np.testing.assert_array_equal(back.natural_parameters, tab.natural_parameters)


def test_version_check(tmp_path):
    path = str(tmp_path / "bad.npz")
    np.savez(path, format_version=np.array(99))
    with pytest.raises(ValueError):
        io.load_tables(path)
