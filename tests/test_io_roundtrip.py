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


def test_relabelled_tables_survive_the_roundtrip(tmp_path):
    """TableSet.permute_sites: the site map travels with the file, every array as relabelled."""
    from tests.cases import load_case

    c = load_case("rocksalt444_ewald")
    tab = capi.TableSet.from_synth(c["sc"], c["coefs"], ewald=c["ewald"], ewald_coef=0.1)
    N = tab.num_sites
    charges = np.zeros((N, 3))
    charges[: c["sc"].size] = [1.0, 3.0, 4.0]
    tab.set_bias(capi.BIAS_SQUARE_CHARGE, charges, penalty=0.75)
    new_of = np.random.default_rng(0).permutation(N)
    before = tab._keep["sub_active_sites"].copy()
    tab.permute_sites(new_of)
    np.testing.assert_array_equal(tab._keep["sub_active_sites"], new_of[before])
    np.testing.assert_array_equal(tab._keep["bias_table"][new_of], charges)
    path = str(tmp_path / "perm.npz")
    io.save_tables(path, tab)
    back = io.load_tables(path)
    np.testing.assert_array_equal(back.site_perm[0], new_of)
    np.testing.assert_array_equal(back.site_perm[1], np.argsort(new_of))
    for name in ("full_idx", "site_ptr", "loc_orbit", "loc_nrows", "ewald_inds", "bias_table", "sub_active_sites"):
        np.testing.assert_array_equal(back._keep[name], tab._keep[name])

    def rows(t, r):  # (the loader packs the local rows in site order: same rows, other offsets)
        n = int(t._keep["loc_nrows"][r]) * int(t._keep["orb_nsites"][t._keep["loc_orbit"][r]])
        return t._keep["loc_idx"][int(t._keep["loc_off"][r]):int(t._keep["loc_off"][r]) + n]

    for r in range(len(tab._keep["loc_orbit"])):
        np.testing.assert_array_equal(rows(back, r), rows(tab, r))
    with pytest.raises(ValueError, match="permutation"):
        tab.permute_sites(np.zeros(N, dtype=int))
