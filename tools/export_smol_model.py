#!/usr/bin/env python3
"""Export a smol Ensemble to the smol_amd table file (run where smol + pymatgen are installed).

    from export_smol_model import export_ensemble
    export_ensemble(ensemble, "model.npz")          # smol.moca.Ensemble

The file holds the flattened arrays of include/smolmc.h:smolmc_tables; smol_amd.io.load_tables
reads it on the GPU box (no smol / pymatgen needed there).  The flattening itself is
smol_amd.bridge.tables_from_ensemble, which only reads attributes of smol's objects (listed
there with file:line); tests/test_bridge_reference_core.py runs it against stand-ins built from
the reference's compiled Cython classes, so attribute visibility is the reference's.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def export_ensemble(ensemble, path, **usher):
    from smol_amd import bridge, io

    tab = bridge.tables_from_ensemble(ensemble, **usher)
    io.save_tables(path, tab)
    return tab


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit("usage: export_smol_model.py <ensemble.json (Ensemble.as_dict)> <out.npz>")
    import json

    from smol.moca import Ensemble

    export_ensemble(Ensemble.from_dict(json.load(open(sys.argv[1]))), sys.argv[2])
