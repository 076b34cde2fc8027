#!/usr/bin/env python3
"""Export a smol Ensemble to the smol_amd table file (run where smol + pymatgen are installed).

    from export_smol_model import export_ensemble
    export_ensemble(ensemble, "model.npz")          # smol.moca.Ensemble

The file holds the flattened arrays of include/smolmc.h:smolmc_tables; smol_amd.io.load_tables
reads it on the GPU box (no smol / pymatgen needed there).  Attribute names follow the reference
checkout: processor._indices.arrays (clusterspace.py:59,1329-1366), processor._eval_data_by_sites
-> LocalEvalData(site_index, evaluator, indices, cluster_ratio) (processor/expansion.py:24-36,
142-156), get_orbit_data (smol/utils/cluster/__init__.py:4-15), EwaldProcessor._ewald_inds /
.ewald_matrix / ._ewald_structure (processor/ewald.py:76-101), Ensemble._chemical_potentials
["table"] (ensemble.py:66-70), Sublattice.active_sites / .encoding (sublattice.py:52-64).

NOTE: this script cannot be exercised in the build container (pymatgen is absent); the file
FORMAT is covered by tests/test_io_roundtrip.py through smol_amd.io.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def export_ensemble(ensemble, path):
    from smol.moca.processor import (ClusterDecompositionProcessor, ClusterExpansionProcessor,
                                     CompositeProcessor, EwaldProcessor)
    from smol.utils.cluster import get_orbit_data

    from smol_amd import capi, io

    proc = ensemble.processor
    ew = None
    if isinstance(proc, CompositeProcessor):
        parts = proc.processors
        ce = next(p for p in parts if isinstance(p, (ClusterExpansionProcessor, ClusterDecompositionProcessor)))
        ew = next((p for p in parts if isinstance(p, EwaldProcessor)), None)
    else:
        ce = proc
    sub = ce.cluster_subspace
    orbit_pos = {orbit.id: i for i, orbit in enumerate(sub.orbits)}
    local = {}
    for site, data in ce._eval_data_by_sites.items():
        evaluator_orbits = data.evaluator._orbit_data  # tuple of (id, bit_id, tensors, strides)
        local[int(site)] = [
            (orbit_pos[od[0]], np.ascontiguousarray(rows, dtype=np.int32), float(ratio))
            for od, rows, ratio in zip(evaluator_orbits, data.indices.arrays, data.cluster_ratio)
        ]
    decomposition = isinstance(ce, ClusterDecompositionProcessor)
    kwargs = {}
    if ew is not None:
        charges = [getattr(site.specie, "oxi_state", 0) or 0 for site in ew._ewald_structure]
        kwargs.update(ewald_inds=ew._ewald_inds, ewald_matrix=ew.ewald_matrix,
                      ewald_coef=float(np.asarray(ew.coefs)), ewald_charges=np.array(charges, float))
    chem = getattr(ensemble, "_chemical_potentials", None)
    if chem is not None:
        kwargs["mu_table"] = chem["table"]
    tab = capi.TableSet(
        ce.num_sites, ce.size, sub.num_orbits, sub.num_corr_functions,
        get_orbit_data(sub.orbits), tuple(ce._indices.arrays), local,
        ce._interaction_tensors if decomposition else None, ce.coefs,
        capi.FEATURES_INTERACTIONS if decomposition else capi.FEATURES_CORRELATIONS,
        [dict(active_sites=s.active_sites, codes=s.encoding) for s in ensemble.active_sublattices],
        **kwargs,
    )
    io.save_tables(path, tab)
    return tab
