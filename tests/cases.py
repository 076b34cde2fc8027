"""Shared model builders for the golden cases (same recipes as tests/golden/make_golden.py)."""

import functools
import os

import numpy as np

from smol_amd import capi, ewald, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    "fcc_conv444_pairs": dict(prim=lambda: synth.fcc_conventional_prim(), cutoffs={2: 6.0},
                              sc=[4, 4, 4], basis="sinusoid", ewald=False, seed=1),
    "fcc_prim666_triplets": dict(prim=lambda: synth.fcc_prim(), cutoffs={2: 6.0, 3: 5.0},
                                 sc=[6, 6, 6], basis="sinusoid", ewald=False, seed=2),
    "rocksalt444_ewald": dict(prim=lambda: synth.rocksalt_prim(), cutoffs={2: 6.0, 3: 5.0},
                              sc=[4, 4, 4], basis="sinusoid", ewald=True, seed=3),
    "fcc3_indicator_skew": dict(prim=lambda: synth.fcc_prim(nspecies=3), cutoffs={2: 5.0, 3: 3.0},
                                sc=[[3, 0, 0], [1, 4, 0], [0, 1, 5]], basis="indicator",
                                ewald=False, seed=4),
    "rocksalt333_vacancy_ewald": dict(prim=lambda: synth.rocksalt_prim(cation_charges=(1.0, 3.0, None)),
                                      cutoffs={2: 6.0, 3: 4.5}, sc=[3, 3, 3], basis="sinusoid",
                                      ewald=True, seed=6),
    "rocksalt333_two_sublattices": dict(prim=lambda: synth.rocksalt_prim(anion_charges=(-2.0, -1.0)),
                                        cutoffs={2: 4.5, 3: 3.2}, sc=[3, 3, 3], basis="sinusoid",
                                        ewald=True, seed=7),
    "fcc_prim222_aliased": dict(prim=lambda: synth.fcc_prim(), cutoffs={2: 6.0, 3: 5.0},
                                sc=[2, 2, 2], basis="sinusoid", ewald=False, seed=5),
}


@functools.lru_cache(maxsize=None)
def load_case(name):
    """Returns dict(gold, model, sc, coefs, ewald) with tables checked against the fixture."""
    spec = CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    model = synth.build_cluster_model(spec["prim"](), spec["cutoffs"], basis=spec["basis"])
    sc = synth.build_supercell(model, spec["sc"])
    coefs = gold["coefs"]
    # the generated tables must be the ones the reference core was driven with
    for o in model.orbits:
        np.testing.assert_allclose(o.flat_correlation_tensors, gold[f"ct_{o.id}"], rtol=0, atol=1e-14)
    ew = ewald.supercell_ewald(sc) if spec["ewald"] else None
    if ew is not None:
        np.testing.assert_allclose(np.diag(ew[1]), gold["ewald_diag"], rtol=1e-12)
        np.testing.assert_allclose(ew[1][0], gold["ewald_row0"], rtol=1e-10, atol=1e-12)
    return dict(gold=gold, model=model, sc=sc, coefs=coefs, ewald=ew)


def tables_for(name, mode, mu_table=None, ewald_coef=0.1):
    c = load_case(name)
    return capi.TableSet.from_synth(
        c["sc"], c["coefs"], feature_mode=mode, ewald=c["ewald"], ewald_coef=ewald_coef,
        mu_table=mu_table,
    )


def flips_of(row):
    return [(int(row[2 * j]), int(row[2 * j + 1])) for j in range(2) if row[2 * j] >= 0]
