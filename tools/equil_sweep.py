#!/usr/bin/env python3
"""Steady-state acceptance of configs 3 and 5 as a function of temperature and chemical-potential
scale (the bench timed these two in the equilibration transient).  For every
parameter set: `equil` steps per walker untimed, then `launches` timed launches; prints one JSON
line per set with the transient (first launches) and the steady-state rate / acceptance.

    python tools/equil_sweep.py --config 3 --T 3000 6000 12000 --mu 0.5 0.1 0.02
    python tools/equil_sweep.py --config 5 --T 400:2000 1500:6000 --mu 0.1
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from smol_amd import parallel, workloads  # noqa: E402
from smol_amd.engine import Engine  # noqa: E402


def measure(eng, wl, launches, rex=None):
    s0 = eng.get_state(occupancy=False)
    ms = []
    for _ in range(launches):
        if rex is None:
            eng.run(wl.mc_per_launch, sync=True)
        else:
            parallel.run_replica_exchange(eng, rex, 1, wl.mc_per_launch)
            eng.sync()
        ms.append(eng.last_kernel_ms())
    s1 = eng.get_state(occupancy=False)
    steps = wl.n_walkers * wl.mc_per_launch
    acc = float((s1["n_accepted"] - s0["n_accepted"]).sum()) / (launches * steps)
    k_ms = float(np.mean(ms))
    return dict(kernel_ms=k_ms, mc_steps_per_s=steps / (k_ms * 1e-3), acceptance=acc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, required=True, choices=(3, 5, 9))
    ap.add_argument("--T", nargs="+", required=True, help="config 3: temperatures; config 5: lo:hi ladders")
    ap.add_argument("--mu", nargs="+", type=float, default=[0.5])
    ap.add_argument("--equil", type=int, default=400_000, help="untimed steps per walker before the steady-state figure")
    ap.add_argument("--launches", type=int, default=5)
    ap.add_argument("--replicas", type=int, default=2048)
    ap.add_argument("--mc", type=int, default=0)
    ap.add_argument("--penalty", nargs="+", type=float, default=[None], help="config 9: SquareChargeBias penalties")
    ap.add_argument("--ewald-coef", nargs="+", type=float, default=[0.1], help="config 3: 1 / epsilon of the Ewald term")
    ap.add_argument("--mu-values", nargs="+", default=[None],
                    help="config 3: 'a,b,c' chemical potentials of Li+ / Mn3+ / Ti4+ instead of the seeded draw")
    a = ap.parse_args()
    for mu, pen, ewc, muv in ((m, p, e, v) for m in a.mu for p in a.penalty for e in a.ewald_coef for v in a.mu_values):
        for T in a.T:
            kw = dict(count=a.replicas, mu_scale=mu)
            if a.mc:
                kw["mc"] = a.mc
            if a.config == 3:
                mv = None if muv in (None, "none") else [float(x) for x in muv.split(",")]
                wl = workloads.config3(temperature=float(T), ewald_coef=ewc, mu_values=mv, **kw)
            elif a.config == 9:
                wl = workloads.config9(temperature=float(T), penalty=pen, **kw)
            else:
                lo, hi = (float(x) for x in T.split(":"))
                wl = workloads.config5(t_lo=lo, t_hi=hi, **kw)
            eng = Engine(wl.tables, wl.make_config())
            eng.set_state(wl.occupancy, wl.seeds, wl.temperature)
            rex = parallel.ReplicaExchange(wl.extras["ladder"], wl.n_walkers, seed=11) if a.config == 5 else None
            first = measure(eng, wl, a.launches, rex)
            done = a.launches * wl.mc_per_launch
            while done < a.equil:
                if rex is None:
                    eng.run(wl.mc_per_launch * 20)
                    done += wl.mc_per_launch * 20
                else:
                    parallel.run_replica_exchange(eng, rex, 5, wl.mc_per_launch)
                    done += wl.mc_per_launch * 5
            eng.sync()
            steady = measure(eng, wl, a.launches, rex)
            st = eng.get_state(occupancy=True)
            nact = wl.sc.size
            comp = [float((st["occupancy"][:, :nact] == c).mean()) for c in range(3)]
            print(json.dumps(dict(config=a.config, T=T, mu_scale=mu, penalty=pen, ewald_coef=ewc, mu_values=muv, kernel=eng.kernel_info(), equil_steps=done,
                                  transient=first, steady=steady, composition=comp,
                                  exchange_acceptance=None if rex is None else float(rex.acceptance.mean()))),
                  flush=True)
            eng.close()


if __name__ == "__main__":
    main()
