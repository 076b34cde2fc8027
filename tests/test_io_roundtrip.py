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


def test_table_flip_and_bias_survive_the_roundtrip(tmp_path):
    """A TableFlip / biased model must not reload as a plain one."""
    from tests.cases import load_case

    c = load_case("rocksalt444_ewald")
    tab = capi.TableSet.from_synth(c["sc"], c["coefs"], ewald=c["ewald"], ewald_coef=0.1,
                                   flip_table=[[1, -3, 2]], flip_weights=[0.7, 0.3], swap_weight=0.25)
    charges = np.zeros((tab.num_sites, 3))
    charges[: c["sc"].size] = [1.0, 3.0, 4.0]
    tab.set_bias(capi.BIAS_SQUARE_CHARGE, charges, penalty=0.75)
    path = str(tmp_path / "tf.npz")
    io.save_tables(path, tab)
    back = io.load_tables(path)
    assert back.struct.n_flip_vectors == 1 and back.struct.swap_weight == 0.25
    np.testing.assert_array_equal(back._keep["flip_table"], [[1, -3, 2]])
    np.testing.assert_array_equal(back._keep["flip_weights"], [0.7, 0.3])
    assert back.struct.bias_type == capi.BIAS_SQUARE_CHARGE and back.struct.bias_penalty == 0.75
    np.testing.assert_array_equal(back._keep["bias_table"], charges)


def test_unknown_arrays_are_refused(tmp_path):
    tab = tables_for("fcc_conv444_pairs", capi.FEATURES_INTERACTIONS)
    path = str(tmp_path / "m.npz")
    io.save_tables(path, tab)
    d = dict(np.load(path))
    d["arr_future_table"] = np.zeros(3)
    np.savez(path, **d)
    with pytest.raises(ValueError, match="future_table"):
        io.load_tables(path)


def test_files_with_python_side_relabelling_are_refused(tmp_path):
    """Rounds 4-5 could write tables the Python binding had renumbered (`site_new_of`); that binding-side translation
    is gone (smolmc_create renumbers internally, ABI 8), so such a file is refused instead of silently running with
    the wrong numbering at the boundary."""
    tab = tables_for("fcc_conv444_pairs", capi.FEATURES_INTERACTIONS)
    path = str(tmp_path / "m.npz")
    io.save_tables(path, tab)
    d = dict(np.load(path))
    d["site_new_of"] = np.arange(tab.num_sites)
    np.savez(path, **d)
    with pytest.raises(ValueError, match="site-relabelled"):
        io.load_tables(path)
