#!/usr/bin/env python3
"""What recording biased rows in-kernel buys (round 6): config 9 (config 3 + SquareChargeBias, 2048 walkers) sampled
through the device ring at several thinning periods, in-kernel rows (default) against the launch + snapshot pairs
(SMOLMC_NO_INKERNEL_BIAS=1), wall clock per block.   python tools/bias_ring_timing.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from smol_amd import workloads  # noqa: E402
from smol_amd.engine import Engine  # noqa: E402

wl = workloads.config9()
for thin in (3456, 256, 32, 4):
    row = dict(config="config9", walkers=wl.n_walkers, thin_by=thin)
    for name, env in (("in_kernel", None), ("snapshot", "1")):
        if env:
            os.environ["SMOLMC_NO_INKERNEL_BIAS"] = env
        else:
            os.environ.pop("SMOLMC_NO_INKERNEL_BIAS", None)
        eng = Engine(wl.tables, wl.make_config())
        eng.set_state(wl.occupancy, wl.seeds, wl.temperature)
        ns = max(4, min(256, 40000 // thin))
        eng.run_sampled(ns, thin, occupancy=False, bias=True)
        t0 = time.perf_counter()
        for _ in range(3):
            s = eng.run_sampled(ns, thin, occupancy=False, bias=True)
        dt = (time.perf_counter() - t0) / 3
        row[name + "_steps_per_s"] = wl.n_walkers * ns * thin / dt
        row[name + "_bias_checksum"] = float(s["bias"].sum())
        eng.close()
    os.environ.pop("SMOLMC_NO_INKERNEL_BIAS", None)
    row["speedup"] = row["in_kernel_steps_per_s"] / row["snapshot_steps_per_s"]
    print(json.dumps(row), flush=True)
