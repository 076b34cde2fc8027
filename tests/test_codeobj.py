"""smol_amd/codeobj.py: per-kernel machine-code digests of libsmolmc_hip.so -- the stamp that ties a PMC entry of
profiles/pmc_constants.json to the kernel it was collected on.  (CPU only: the library is cross-compiled here.)"""
import json
import os

import pytest

from smol_amd import codeobj, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.exists(engine.LIB_PATH), reason="libsmolmc_hip.so not built")


def test_every_kernel_family_has_digests():
    d = codeobj.kernel_digests()
    assert len(d) > 500
    for family in ("mc_lean_kernel<", "mc_lean_multi_kernel<", "mc_wl_kernel<", "mc_table_kernel<", "mc_table_multi_kernel<",
                   "mc_kernel<", "mc_univ_kernel<", "eval_full_kernel"):
        assert any(family in n for n in d), family
    assert all(len(v) == 64 for v in d.values())
    # the headline instantiation, spelled as rocprofv3 spells it and cut short as a summary may cut it
    name = "void mc_lean_kernel<2, 2, 1, false, 0, false, false, true, 0, 0, false>(LeanParams)"
    assert codeobj.find_kernel(name) == (name, d[name])
    assert codeobj.find_kernel("mc_lean_kernel<2,2,1,false,0,false,false,true,0,0,false>")[0] == name
    assert codeobj.find_kernel("mc_lean_kernel<") is None  # ambiguous fragments do not match


def test_pmc_entries_name_kernels_of_this_library():
    data = json.load(open(os.path.join(ROOT, "profiles", "pmc_constants.json")))
    assert data
    for key, e in data.items():
        assert e.get("kernel_symbol") and e.get("isa_sha256"), key
        assert codeobj.find_kernel(e["kernel_symbol"]) is not None, key
        assert e["kernel"] in e["kernel_symbol"], key


def test_isa_stale_follows_the_machine_code_not_the_comments():
    name, dig = codeobj.find_kernel("void mc_table_kernel<2, 2, 2, false, false, false>(LeanParams)")
    fresh = dict(kernel_symbol=name, isa_sha256=dig, csrc_sha256="0" * 64)  # (another tree, same kernel bytes)
    assert not codeobj.isa_stale(fresh)
    assert codeobj.isa_stale(dict(fresh, isa_sha256="1" * 64))
    assert codeobj.isa_stale(dict(fresh, kernel_symbol="void no_such_kernel<1>(LeanParams)"))
    assert not codeobj.isa_stale({})  # no PMC pass for the shape: nothing to be stale
    # entries from before the per-kernel stamp: the digest over the source tree decides
    assert not codeobj.isa_stale(dict(csrc_sha256=engine.source_digest()))
    assert codeobj.isa_stale(dict(csrc_sha256="0" * 64))


def test_a_name_emitted_by_several_translation_units_changes_when_any_copy_does():
    """Static __global__ helpers of a shared header appear once per translation unit under one mangled name: equal
    copies keep the copy's digest, differing copies get one over all of them -- not "whichever bundle came last"."""
    a, b, c = "a" * 64, "b" * 64, "c" * 64
    m = codeobj._merge_copies({"k1": {a}, "k2": {a, b}, "k3": {b, a}})
    assert m["k1"] == a
    assert m["k2"] == m["k3"] and m["k2"] not in (a, b) and len(m["k2"]) == 64
    assert codeobj._merge_copies({"k2": {a, c}})["k2"] != m["k2"]
