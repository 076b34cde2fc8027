"""Wire format for model tables ("general table import").

A model is exchanged as ONE ``.npz`` holding exactly the flattened arrays and scalars of
``smolmc_tables`` (include/smolmc.h) -- the input side of the hot path.  ``save_tables``
writes it from a TableSet; ``load_tables`` rebuilds a TableSet without smol_amd.synth, so a
model exported on a machine that has smol + pymatgen (tools/export_smol_model.py) runs
here unchanged: arbitrary lattices, multi-sublattice systems, real ECIs.
"""

from __future__ import annotations

import numpy as np

from . import capi

FORMAT_VERSION = 1
_SCALARS = ("num_sites", "size", "num_orbits", "num_corr", "max_species", "n_orb", "feature_mode",
            "has_ewald", "ewald_dim", "ewald_width", "has_mu", "mu_width", "n_sublattices",
            "n_flip_vectors", "bias_type", "bias_width", "bias_rows")
_FLOATS = ("offset", "ewald_coef", "swap_weight", "bias_penalty")
# every array a TableSet keeps alive; a file holding anything else was written by a newer
# exporter and must not be loaded as if the extra tables were absent
_ARRAYS = frozenset((
    "orb_id", "orb_bit_id", "orb_nsites", "orb_nfunc", "orb_tensor_len", "orb_stride_off",
    "tensor_indices", "orb_ctensor_off", "corr_tensors", "orb_itensor_off", "interaction_tensors",
    "full_off", "full_idx", "site_ptr", "loc_orbit", "loc_ratio", "loc_nrows", "loc_off", "loc_idx",
    "ce_coefs", "ewald_inds", "ewald_matrix", "ewald_charges", "mu_table", "sub_site_ptr",
    "sub_active_sites", "sub_code_ptr", "sub_codes", "sub_probs", "flip_table", "flip_weights",
    "bias_table", "bias_intercepts",
))


def save_tables(path, tab: capi.TableSet):
    """Write a TableSet to ``path`` (.npz)."""
    t = tab.struct
    out = {"format_version": np.array(FORMAT_VERSION)}
    for name in _SCALARS:
        out[name] = np.array(getattr(t, name), dtype=np.int64)
    for name in _FLOATS:
        out[name] = np.array(getattr(t, name), dtype=np.float64)
    for name, arr in tab._keep.items():
        out["arr_" + name] = arr
    np.savez_compressed(path, **out)


def load_tables(path) -> capi.TableSet:
    """Rebuild a TableSet from a file written by save_tables / export_smol_model.py.

    Goes through the normal TableSet constructor (all its validation applies)."""
    d = np.load(path)
    if int(d["format_version"]) != FORMAT_VERSION:
        raise ValueError("unsupported table file version")
    A = {k[4:]: d[k] for k in d.files if k.startswith("arr_")}
    unknown = sorted(set(A) - _ARRAYS)
    if unknown:
        raise ValueError(f"table file holds arrays this loader does not know: {unknown}")
    n_orb = int(d["n_orb"])
    N = int(d["num_sites"])
    orbit_data, full, its = [], [], [float(d["offset"])]
    for o in range(n_orb):
        I, K, Nt = int(A["orb_nsites"][o]), int(A["orb_nfunc"][o]), int(A["orb_tensor_len"][o])
        so, co, io = int(A["orb_stride_off"][o]), int(A["orb_ctensor_off"][o]), int(A["orb_itensor_off"][o])
        orbit_data.append((int(A["orb_id"][o]), int(A["orb_bit_id"][o]),
                           np.ascontiguousarray(A["corr_tensors"][co:co + K * Nt].reshape(K, Nt)),
                           np.ascontiguousarray(A["tensor_indices"][so:so + I])))
        a, b = int(A["full_off"][o]), int(A["full_off"][o + 1])
        full.append(np.ascontiguousarray(A["full_idx"][a:b].reshape(-1, I)))
        its.append(A["interaction_tensors"][io:io + Nt])
    local = {}
    for s in range(N):
        recs = []
        for r in range(int(A["site_ptr"][s]), int(A["site_ptr"][s + 1])):
            o = int(A["loc_orbit"][r])
            I, J, off = int(A["orb_nsites"][o]), int(A["loc_nrows"][r]), int(A["loc_off"][r])
            recs.append((o, np.ascontiguousarray(A["loc_idx"][off:off + J * I].reshape(J, I)),
                         float(A["loc_ratio"][r])))
        if recs:
            local[s] = recs
    subs = []
    for k in range(int(d["n_sublattices"])):
        subs.append(dict(
            active_sites=A["sub_active_sites"][int(A["sub_site_ptr"][k]):int(A["sub_site_ptr"][k + 1])],
            codes=A["sub_codes"][int(A["sub_code_ptr"][k]):int(A["sub_code_ptr"][k + 1])],
        ))
    has_ew, has_mu = int(d["has_ewald"]), int(d["has_mu"])
    tab = capi.TableSet(
        N, int(d["size"]), int(d["num_orbits"]), int(d["num_corr"]), tuple(orbit_data), tuple(full),
        local, its, A["ce_coefs"], int(d["feature_mode"]), subs,
        sublattice_probabilities=A["sub_probs"],
        ewald_inds=A["ewald_inds"] if has_ew else None,
        ewald_matrix=A["ewald_matrix"] if has_ew else None,
        ewald_coef=float(d["ewald_coef"]),
        mu_table=A["mu_table"] if has_mu else None,
        ewald_charges=A.get("ewald_charges"),
        flip_table=A.get("flip_table"),
        flip_weights=A.get("flip_weights"),
        swap_weight=float(d["swap_weight"]) if "swap_weight" in d.files else 0.1,
    )
    tab.struct.max_species = int(d["max_species"])
    if "bias_table" in A:  # MCBias term (files written before it was persisted have none)
        tab.set_bias(int(d["bias_type"]), A["bias_table"], float(d["bias_penalty"]),
                     intercepts=A.get("bias_intercepts"))
    if "site_new_of" in d.files:
        # (files of rounds 4-5 could hold tables the Python binding had renumbered; the map was only ever applied by
        # that binding, which no longer exists -- smolmc_create renumbers internally)
        raise ValueError("this file holds site-relabelled tables (site_new_of) of an older smol_amd: export it again")
    return tab
