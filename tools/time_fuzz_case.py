"""Steps per second of a fuzz-campaign model (tests/fuzz_campaign.py, by case seed) at a production walker count, on
the kernel the handle picks -- for A/B runs of dispatch thresholds (environment switches are read at smolmc_create).
    python tools/time_fuzz_case.py --seed 62000430 --profile any [--walkers 2048] [--steps 4000]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smol_amd import capi  # noqa: E402
from smol_amd.engine import Engine  # noqa: E402
from tests import fuzz_campaign as fc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, required=True)
    ap.add_argument("--profile", default="any")
    ap.add_argument("--walkers", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=4000)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    case = fc.build_case(rng, args.profile)
    if case is None:
        print(json.dumps(dict(seed=args.seed, status="void")))
        return
    cfg0, R = case["cfg"], args.walkers
    cfg = capi.make_config(R, cfg0.kernel_type, cfg0.step_type, min_enthalpy=cfg0.wl_min_enthalpy, max_enthalpy=cfg0.wl_max_enthalpy,
                           bin_size=cfg0.wl_bin_size, flatness=cfg0.wl_flatness, check_period=int(cfg0.wl_check_period),
                           update_period=int(cfg0.wl_update_period))
    eng = Engine(case["tab_engine"], cfg)
    reps = -(-R // len(case["occ"]))
    occ = np.tile(case["occ"], (reps, 1))[:R]
    temps = np.tile(np.atleast_1d(case["temps"]), reps)[:R] if np.ndim(case["temps"]) else case["temps"]
    eng.set_state(occ, np.arange(R, dtype=np.uint64) + np.uint64(5), temps)
    eng.run(args.steps // 4)
    best = None
    for _ in range(3):
        eng.run(args.steps)
        dt = eng.last_kernel_ms() * 1e-3  # (HIP events around the launch)
        best = dt if best is None else min(best, dt)
    st = eng.get_state()
    print(json.dumps(dict(seed=args.seed, kernel_info=eng.kernel_info()[:70], walkers=R, steps=args.steps,
                          steps_per_s=R * args.steps / best, acceptance=float(st["n_accepted"].sum() / st["n_steps"].sum()),
                          desc={k: case["desc"].get(k) for k in ("nspecies", "cutoffs", "supercell", "kernel", "step", "sites")})))


if __name__ == "__main__":
    main()
