"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/smolmc.h declares (no compute calls without a GPU),
and the product path fails loudly when no device is present."""

import ctypes
import os
import re

import numpy as np
import pytest

from smol_amd import capi, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g

    if not os.path.exists(engine.LIB_PATH):
        g.build()
    return engine.load_library()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "smolmc.h")).read()
    declared = set(re.findall(r"\b(smolmc_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(engine.SYMBOLS), declared ^ set(engine.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.smolmc_abi_version() == capi.ABI_VERSION == 8


def test_struct_layout_matches_header():
    """ctypes mirrors must agree with the C structs (compiled probe)."""
    import subprocess
    import tempfile

    src = r"""
    #include <stdio.h>
    #include <stddef.h>
    #include "smolmc.h"
    int main(){printf("%zu %zu %zu %zu %zu %zu\n", sizeof(smolmc_tables), sizeof(smolmc_config),
      offsetof(smolmc_tables, offset), offsetof(smolmc_tables, ewald_coef),
      offsetof(smolmc_tables, sub_probs), offsetof(smolmc_config, wl_check_period));return 0;}
    """
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "p"),
                               os.path.join(d, "p.c")])
        out = subprocess.check_output([os.path.join(d, "p")]).split()
    vals = [int(x) for x in out]
    T, Cf = capi.smolmc_tables, capi.smolmc_config
    assert vals == [ctypes.sizeof(T), ctypes.sizeof(Cf), T.offset.offset, T.ewald_coef.offset,
                    T.sub_probs.offset, Cf.wl_check_period.offset]


def test_no_cpu_fallback_without_gpu(lib):
    """Without a HIP device the engine must refuse to run rather than fall back."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tests.cases import tables_for

    tab = tables_for("fcc_prim222_aliased", capi.FEATURES_INTERACTIONS)
    with pytest.raises(RuntimeError, match="HIP"):
        engine.Engine(tab, capi.make_config(1))


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.load_library(str(tmp_path / "libsmolmc_hip.so"))


def test_tableset_validation_errors():
    """Error conventions of container.pyx:52-64 / expansion.py:97-103."""
    from smol_amd import synth

    m = synth.build_cluster_model(synth.fcc_prim(), {2: 3.0})
    sc = synth.build_supercell(m, [3, 3, 3])
    coefs = synth.random_coefs(m)
    with pytest.raises(ValueError):
        capi.TableSet.from_synth(sc, coefs[:-1], feature_mode=capi.FEATURES_CORRELATIONS)
    od = list(m.orbit_data())
    bad = [(1.5,) + od[0][1:]] + od[1:]
    subs = [dict(active_sites=np.arange(sc.num_sites), codes=np.arange(2))]
    with pytest.raises(TypeError):
        capi.TableSet(sc.num_sites, sc.size, m.num_orbits, m.num_corr_functions, tuple(bad),
                      tuple(sc.full_indices), sc.local_tables(), None, coefs,
                      capi.FEATURES_CORRELATIONS, subs)
    bad2 = [(od[0][0], od[0][1], od[0][2].ravel(), od[0][3])] + od[1:]
    with pytest.raises(ValueError):
        capi.TableSet(sc.num_sites, sc.size, m.num_orbits, m.num_corr_functions, tuple(bad2),
                      tuple(sc.full_indices), sc.local_tables(), None, coefs,
                      capi.FEATURES_CORRELATIONS, subs)


def test_step_records_pad_to_the_abi_row():
    """capi.step_rows: the ushers' lists of (site, code) tuples / narrow int arrays -> SMOLMC_STEP_ROW records."""
    assert capi.STEP_ROW == 16 and capi.MAX_STEP_FLIPS == 8
    r = capi.step_rows([[(1, 2), (3, 4), (5, 6)], []])
    assert r.shape == (2, 16) and r.dtype == np.int32
    assert r[0, :6].tolist() == [1, 2, 3, 4, 5, 6] and (r[0, 6:] == -1).all() and (r[1] == -1).all()
    narrow = np.array([[[7, 1, -1, -1]], [[8, 0, 9, 1]]], dtype=np.int32)  # (R, n, 4): the pre-v6 layout
    w = capi.step_rows(narrow, 2, 1)
    assert w.shape == (2, 1, 16) and w[1, 0, :4].tolist() == [8, 0, 9, 1] and (w[..., 4:] == -1).all()
    with pytest.raises(ValueError):
        capi.step_rows([[(0, 0)] * 9])
    with pytest.raises(ValueError):
        capi.step_rows(np.zeros((3, 5), dtype=np.int32))
