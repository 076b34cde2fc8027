#!/usr/bin/env python3
"""Throughput of one BASELINE.json configuration (or one of the shapes outside it) on one GPU:
kernel time from HIP events on the launch stream.  The configurations are defined once, in
smol_amd/workloads.py; bench.py reports configs 1, 3, 4, 5 in its `other_configs` array with the
same builders.

    python tools/bench_configs.py --config 4 [--replicas 1024] [--mc 5000] [--launches 3]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from smol_amd import capi, parallel, workloads  # noqa: E402
from smol_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, required=True, choices=sorted(workloads.BUILDERS))
    ap.add_argument("--replicas", type=int, default=0)
    ap.add_argument("--mc", type=int, default=0)
    ap.add_argument("--launches", type=int, default=3)
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--temperature", type=float, default=0.0)
    ap.add_argument("--ladder", default="", help="config 5: lowest,highest temperature of the exchange ladder")
    ap.add_argument("--sampled", type=int, default=0, help="also time blocks of the device ring with this thin_by (wall clock, "
                                                            "features recorded, no occupancies): what lazy cluster features cost")
    a = ap.parse_args()
    kw = {}
    if a.ladder:
        kw["t_lo"], kw["t_hi"] = (float(x) for x in a.ladder.split(","))
    if a.replicas:
        kw["count"] = a.replicas
    if a.dim:
        kw["dim"] = a.dim
    if a.mc:
        kw["mc"] = a.mc
    wl = workloads.BUILDERS[a.config](**kw)
    if a.config in (4, 10, 11):  # the window is centred on the starting enthalpy, evaluated on the engine
        probe = Engine(wl.tables, capi.make_config(1))
        h0 = float(probe.natural_parameters @ probe.eval_full(wl.occupancy[:1])[0])
        if a.config == 11:  # random starts with a long upper tail: the window's upper edge 30 eV above the highest
            h0 = float((probe.eval_full(wl.occupancy) @ probe.natural_parameters).max()) + 30.0 - 95.63
        probe.close()
        wl = workloads.BUILDERS[a.config](h0=h0, **kw)
    eng = Engine(wl.tables, wl.make_config())
    T = a.temperature if a.temperature > 0 else wl.temperature
    eng.set_state(wl.occupancy, wl.seeds, T)
    R, mc = wl.n_walkers, wl.mc_per_launch
    rex = None
    if a.config == 5:
        rex = parallel.ReplicaExchange(wl.extras["ladder"], R, seed=11)

    def launch():
        if rex is None:
            eng.run(mc, sync=True)
        else:
            parallel.run_replica_exchange(eng, rex, 1, mc)
            eng.sync()

    launch()
    s0 = eng.get_state(occupancy=False)
    ms = []
    for _ in range(a.launches):
        launch()
        ms.append(eng.last_kernel_ms())
    s1 = eng.get_state(occupancy=False)
    k_ms = float(np.mean(ms))
    steps = R * mc
    sampled = None
    if a.sampled:
        import time

        ns = max(1, mc // a.sampled)
        eng.run_sampled(ns, a.sampled, occupancy=False)
        t0 = time.perf_counter()
        for _ in range(a.launches):
            eng.run_sampled(ns, a.sampled, occupancy=False)
        dt = (time.perf_counter() - t0) / a.launches
        sampled = dict(thin_by=a.sampled, samples_per_block=ns, wall_ms=dt * 1e3, mc_steps_per_s=R * ns * a.sampled / dt)
    print(json.dumps(dict(
        config=wl.name, kernel=eng.kernel_info(), replicas=R, mc_steps_per_launch=mc, kernel_ms=k_ms,
        mc_steps_per_s=steps / (k_ms * 1e-3), flips_per_s=wl.flips_per_step * steps / (k_ms * 1e-3),
        us_per_step_per_walker=k_ms * 1e3 / mc,
        acceptance=float((s1["n_accepted"] - s0["n_accepted"]).sum()) / (a.launches * steps),
        exchange_acceptance_mean=None if rex is None else float(rex.acceptance.mean()),
        sampled=sampled,
    )))


if __name__ == "__main__":
    main()
