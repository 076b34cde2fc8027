"""Wang-Landau with TableFlip proposals on BASELINE config 5's model (12^3 ternary rocksalt + Ewald, 2048 walkers): the lean
table kernel (mc_table_kernel<..., WLT>, round 6) against the universal kernel it ran on before (SMOLMC_NO_TABLE_WL), one
sweep of 3456 steps per launch, kernel time from the HIP events around the launch.
    python tools/wl_table_timing.py [--walkers 2048] [--update-period 1] > gpurun_out/wl_table_timing.jsonl"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smol_amd import capi, workloads  # noqa: E402
from smol_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--walkers", type=int, default=2048)
    ap.add_argument("--update-period", type=int, default=1)
    args = ap.parse_args()
    wl = workloads.config5()
    R = args.walkers
    occ, seeds = wl.occupancy[:R], wl.seeds[:R]
    probe = Engine(wl.tables, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP))
    probe.set_state(occ, seeds, 2000.0)
    h0 = probe.get_state()["enthalpy"]
    probe.close()
    lo, hi = h0.min() - 40.0, h0.max() + 40.0
    cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, capi.STEP_TABLE_FLIP, min_enthalpy=lo, max_enthalpy=hi,
                           bin_size=(hi - lo) / 200.0, check_period=1000, update_period=args.update_period)
    for env in (None, "SMOLMC_NO_TABLE_WL"):
        if env:
            os.environ[env] = "1"
        eng = Engine(wl.tables, cfg)
        os.environ.pop("SMOLMC_NO_TABLE_WL", None)
        eng.set_state(occ, seeds, 0.0)
        eng.run(wl.mc_per_launch)
        ms = []
        for _ in range(3):
            eng.run(wl.mc_per_launch)
            ms.append(eng.last_kernel_ms())
        st = eng.get_state()
        print(json.dumps(dict(kernel=eng.kernel_info()[:80], walkers=R, update_period=args.update_period, steps_per_launch=wl.mc_per_launch,
                              kernel_ms=min(ms), steps_per_s=R * wl.mc_per_launch / (min(ms) * 1e-3),
                              acceptance=float(st["n_accepted"].sum() / st["n_steps"].sum()))), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
