#!/usr/bin/env python3
"""Kernel time of a model with RESTRICTED sites (Ensemble.restrict_sites: 10 % of the cations of the
config-3 lattice frozen) through the Sampler, with and without the site relabelling that keeps such
models on the specialised kernels (SMOLMC_NO_SITE_RELABEL=1: the general / universal kernels).
    python tools/bench_restricted.py [--step swap|flip|table-flip] [--walkers 2048] [--steps 2000]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from smol_amd import moca, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--step", default="swap")
    ap.add_argument("--walkers", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--dim", type=int, default=12)
    a = ap.parse_args()
    model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 6.0, 3: 4.5})
    sc = synth.build_supercell(model, [a.dim] * 3)
    ens = moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model), ewald_coefficient=0.1)
    rng = np.random.default_rng(0)
    cations = ens.sublattices[0]
    ens.restrict_sites(rng.choice(cations.sites, len(cations.sites) // 10, replace=False))
    if a.step == "flip":
        ens.chemical_potentials = {sp: 0.0 for sp in ens.species}
    P = sc.size
    occ = np.zeros((a.walkers, sc.num_sites), dtype=np.int32)
    base = np.zeros(P, dtype=np.int32)  # charge neutral: n_Li + 3 n_Mn + 4 n_Ti = 2 P
    n_ti = P // 6
    n_mn = (P - 3 * n_ti) // 2
    base[:n_mn] = 1
    base[n_mn:n_mn + n_ti] = 2
    for w in range(a.walkers):
        occ[w, :P] = rng.permutation(base)
    kw = dict(flip_table=[[1, -3, 2, 0]], swap_weight=0.1) if a.step == "table-flip" else {}
    sampler = moca.Sampler.from_ensemble(ens, temperature=4000.0, step_type=a.step, nwalkers=a.walkers,
                                         seeds=list(range(a.walkers)), **kw)
    sampler.run(a.steps, occ, thin_by=a.steps)
    ms = []
    for _ in range(3):
        sampler.run(a.steps, thin_by=a.steps)
        ms.append(sampler.engine.last_kernel_ms())
    acc = float(sampler.samples.get_trace_value("accepted", flat=False)[1:].mean())
    print(json.dumps(dict(step=a.step, walkers=a.walkers, relabel=os.environ.get("SMOLMC_NO_SITE_RELABEL") is None,
                          kernel=sampler.engine.kernel_info(), kernel_ms=min(ms),
                          steps_per_s=a.walkers * a.steps / (min(ms) * 1e-3), last_step_accepted=acc)))


if __name__ == "__main__":
    main()
