"""Flattening of smol's OWN objects into the engine tables: the reference-side half of the
boundary.  Runs where smol + pymatgen are installed; imports nothing from smol --
it only reads attributes of the objects it is handed, so the same code is what a maintainer would
put next to smol's Sampler and what tools/export_smol_model.py uses to write the .npz wire format.

Attributes read (reference checkout):
    ensemble.processor, .active_sublattices, ._chemical_potentials["table"]   ensemble.py:102-217,66-70
    CompositeProcessor.processors                                                composite.py:49-52
    processor.cluster_subspace (.orbits, .num_orbits, .num_corr_functions)       clusterspace.py
    orbit.id / .bit_id / .flat_correlation_tensors / .flat_tensor_indices        orbit.py:251-275,478-498
    processor._indices.arrays            OrbitIndices                            clusterspace.py:59,1329-1366
    processor._eval_data_by_sites        site -> LocalEvalData(site_index, evaluator, indices,
                                         cluster_ratio)                          processor/expansion.py:24-36,142-156
    LocalEvalData.evaluator.__reduce__() the evaluator's orbit tuples: its `_orbit_data` is a plain
                                         `cdef tuple` (container.pxd:22), invisible from Python;
                                         the pickling protocol (evaluator.pyx:65-77) is the public
                                         way to read it back
    ClusterDecompositionProcessor._interaction_tensors                           processor/expansion.py:324
    EwaldProcessor._ewald_inds / .ewald_matrix / ._ewald_structure / .coefs      processor/ewald.py:76-101
    Sublattice.active_sites / .encoding                                          sublattice.py:52-64
"""

from __future__ import annotations

import numpy as np

from . import capi


def orbit_data_of(orbits):
    """smol/utils/cluster/__init__.py:4-15 (get_orbit_data)."""
    return tuple((int(o.id), int(o.bit_id), o.flat_correlation_tensors, o.flat_tensor_indices) for o in orbits)


def split_processor(processor):
    """(cluster processor, Ewald processor or None) of a plain or composite processor."""
    parts = list(getattr(processor, "processors", None) or [processor])
    ewald = [p for p in parts if hasattr(p, "_ewald_inds")]
    cluster = [p for p in parts if hasattr(p, "_eval_data_by_sites")]
    if len(cluster) != 1 or len(ewald) > 1 or len(cluster) + len(ewald) != len(parts):
        raise NotImplementedError(
            "supported: one ClusterExpansionProcessor / ClusterDecompositionProcessor, optionally "
            "composed with one EwaldProcessor")
    return cluster[0], (ewald[0] if ewald else None)


def local_tables_of(cluster_processor):
    """site -> [(orbit position, rows int32[J, I], ratio)] from the processor's LocalEvalData."""
    sub = cluster_processor.cluster_subspace
    position = {int(orbit.id): i for i, orbit in enumerate(sub.orbits)}
    out = {}
    for site, data in cluster_processor._eval_data_by_sites.items():
        orbit_tuples = data.evaluator.__reduce__()[1][0]  # ((id, bit_id, tensors, strides), ...)
        out[int(site)] = [
            (position[int(od[0])], np.ascontiguousarray(rows, dtype=np.int32), float(ratio))
            for od, rows, ratio in zip(orbit_tuples, data.indices.arrays, data.cluster_ratio)
        ]
    return out


def tables_from_ensemble(ensemble, flip_table=None, flip_weights=None, swap_weight=0.1):
    """capi.TableSet (= smolmc_tables + the arrays it points at) of a smol.moca.Ensemble, in smol's own site
    numbering.  Restricted sites and sublattices split by species (sublattice.py:84-186) leave the active sites of a
    sublattice scattered; ``smolmc_create`` renumbers the sites internally then (ABI 8) and every entry point keeps
    speaking the caller's numbering, so neither this function nor any C-ABI client has anything to translate."""
    ce, ew = split_processor(ensemble.processor)
    sub = ce.cluster_subspace
    decomposition = hasattr(ce, "_interaction_tensors")
    kwargs = {}
    if ew is not None:
        charges = [getattr(site.specie, "oxi_state", 0) or 0 for site in ew._ewald_structure]
        kwargs.update(ewald_inds=ew._ewald_inds, ewald_matrix=ew.ewald_matrix,
                      ewald_coef=float(np.asarray(ew.coefs)), ewald_charges=np.array(charges, float))
    chem = getattr(ensemble, "_chemical_potentials", None)
    if chem is not None:
        kwargs["mu_table"] = chem["table"]
    if flip_table is not None:
        kwargs.update(flip_table=flip_table, flip_weights=flip_weights, swap_weight=swap_weight)
    tab = capi.TableSet(
        ce.num_sites, ce.size, sub.num_orbits, sub.num_corr_functions,
        orbit_data_of(sub.orbits), tuple(ce._indices.arrays), local_tables_of(ce),
        ce._interaction_tensors if decomposition else None, ce.coefs,
        capi.FEATURES_INTERACTIONS if decomposition else capi.FEATURES_CORRELATIONS,
        [dict(active_sites=s.active_sites, codes=s.encoding) for s in ensemble.active_sublattices],
        **kwargs,
    )
    return tab


def engine_from_sampler_arguments(ensemble, nwalkers, kernel_type="metropolis", step_type="swap",
                                  device=0, **wl):
    """What a ``Sampler.from_ensemble`` in smol would call once to get its engine handle."""
    from .engine import Engine

    kernels = {"metropolis": capi.KERNEL_METROPOLIS, "wanglandau": capi.KERNEL_WANGLANDAU}
    steps = {"flip": capi.STEP_FLIP, "swap": capi.STEP_SWAP, "table-flip": capi.STEP_TABLE_FLIP}
    tables = tables_from_ensemble(ensemble)
    cfg = capi.make_config(nwalkers, kernels[kernel_type.lower().replace("-", "")], steps[step_type],
                           device, **wl)
    return Engine(tables, cfg)
