#!/usr/bin/env python3
"""Bounded experiments on the headline kernel:

  --plateau      config 2 at R = 1024 .. 16384 walkers per GPU: where does throughput saturate?
  --two-walker   upper bound for a two-walkers-per-wave kernel: a wave that evaluates TWICE the
                 clusters per flip (NSLOT = 4, a 237-clusters-per-site model on the same lattice)
                 is what each wave of such a kernel would at least have to do for its two chains
                 (before vectorised proposals, divergent accept paths and per-half reductions);
                 bound = 2 * t_step(NSLOT=2) / t_step(NSLOT=4).
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from smol_amd import capi, synth, workloads  # noqa: E402
from smol_amd.engine import Engine  # noqa: E402


def time_model(cutoffs, R, mc=5000, launches=4):
    model = synth.build_cluster_model(synth.fcc_prim(a=4.09), cutoffs)
    sc = synth.build_supercell(model, [16] * 3)
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=20260928, scale=0.02))
    eng = Engine(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP))
    eng.set_state(workloads.balanced_binary(sc, 0, R), np.arange(R, dtype=np.uint64) + np.uint64(12345), 2500.0)
    eng.run(mc, sync=True)
    s0 = eng.get_state(occupancy=False)
    ms = []
    for _ in range(launches):
        eng.run(mc, sync=True)
        ms.append(eng.last_kernel_ms())
    s1 = eng.get_state(occupancy=False)
    k = float(np.mean(ms))
    ncl = sum(o.multiplicity * o.size for o in model.orbits)
    return dict(cutoffs={str(a): b for a, b in cutoffs.items()}, clusters_per_site=int(ncl), kernel=eng.kernel_info(),
                replicas=R, kernel_ms=k, ns_per_wave_step=k * 1e6 / mc,
                flips_per_s=2.0 * R * mc / (k * 1e-3),
                acceptance=float((s1["n_accepted"] - s0["n_accepted"]).sum()) / (launches * R * mc))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plateau", action="store_true")
    ap.add_argument("--two-walker", action="store_true")
    a = ap.parse_args()
    if a.plateau:
        for R in (512, 1024, 2048, 4096, 6144, 8192, 12288, 16384):
            print(json.dumps(time_model({2: 6.0, 3: 5.0}, R)), flush=True)
    if a.two_walker:
        t2 = time_model({2: 6.0, 3: 5.0}, 4096)
        t4 = time_model({2: 9.0, 3: 5.0}, 4096)
        print(json.dumps(t2))
        print(json.dumps(t4))
        print(json.dumps(dict(two_walkers_per_wave_speedup_upper_bound=2.0 * t2["kernel_ms"] / t4["kernel_ms"],
                              needed_for_1p8e10=1.8e10 / t2["flips_per_s"])))


if __name__ == "__main__":
    main()
