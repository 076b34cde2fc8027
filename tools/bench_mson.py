"""Throughput of the reference's own LiNiO2 model (tests/golden/lno_ce_ewald.mson.json.gz, read with
smol_amd.mson) on the engine: two active sublattices (Li+/vacancy, Ni3+/Ni4+), fixed O2-, Ewald
term with a vacancy species -- the shape real smol models have, next to the synthetic BASELINE
configurations.  One JSON line per case: kernel chosen, steps/s, acceptance.

  python tools/bench_mson.py [--dim 8] [--walkers 4096] [--mc 2000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smol_amd import capi, mson  # noqa: E402
from smol_amd.engine import Engine  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                    "lno_ce_ewald.mson.json.gz")


def neutral(cell, R, rng, n_li):
    P = cell.size
    occ = np.ones((R, cell.num_sites), dtype=np.int32)
    occ[:, 2 * P:] = 0
    for r in range(R):
        occ[r, rng.permutation(P)[:n_li]] = 0
        occ[r, P + rng.permutation(P)[:n_li]] = 0
    return occ


def wang_landau(a):
    """The reference's own model under Wang-Landau (kernel/wanglandau.py): two active sublattices, Ewald
    term; the window is centred on the starting enthalpy evaluated on the engine (as for config 4)."""
    ce = mson.load_mson(GOLD)
    rng = np.random.default_rng(3)
    for mode in ("int", "corr"):
        fmode = capi.FEATURES_CORRELATIONS if mode == "corr" else capi.FEATURES_INTERACTIONS
        tab = ce.tables(np.diag([a.dim] * 3), feature_mode=fmode)
        cell = tab.supercell
        occ = neutral(cell, a.walkers, rng, cell.size // 2)
        probe = Engine(tab, capi.make_config(1))
        h0 = float(probe.natural_parameters @ probe.eval_full(occ[:1])[0])
        probe.close()
        for step, name in ((capi.STEP_SWAP, "swap"), (capi.STEP_FLIP, "flip")):
            for upd in (1, 3):
                cfg = capi.make_config(a.walkers, capi.KERNEL_WANGLANDAU, step, min_enthalpy=h0 - 160.37,
                                       max_enthalpy=h0 + 95.63, bin_size=0.5, flatness=0.8, check_period=1000,
                                       update_period=upd)
                eng = Engine(tab, cfg)
                eng.set_state(occ, np.arange(a.walkers, dtype=np.uint64) + np.uint64(11), 0.0)
                eng.run(a.mc)
                eng.sync()
                ms = []
                for _ in range(3):
                    eng.run(a.mc)
                    ms.append(eng.last_kernel_ms())
                st = eng.get_state(occupancy=False)
                print(json.dumps({"model": "LiNiO2 + Ewald (reference .mson), Wang-Landau", "sites": int(cell.num_sites),
                                  "walkers": a.walkers, "features": mode, "step": name, "update_period": upd,
                                  "kernel": eng.kernel_info(), "kernel_ms": float(np.mean(ms)),
                                  "steps_per_s": a.walkers * a.mc / (np.mean(ms) * 1e-3),
                                  "acceptance": float(st["n_accepted"].sum() / st["n_steps"].sum())}), flush=True)
                eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=8)
    ap.add_argument("--walkers", type=int, default=4096)
    ap.add_argument("--mc", type=int, default=2000)
    ap.add_argument("--temperature", type=float, default=1200.0)
    ap.add_argument("--kernel", default="metropolis", choices=("metropolis", "wang-landau"),
                    help="wang-landau: canonical swaps / semigrand-free flips under the Wang-Landau kernel "
                         "(512 bins of 0.5 eV around the starting enthalpy), interaction and correlation features")
    a = ap.parse_args()
    if a.kernel == "wang-landau":
        return wang_landau(a)
    ce = mson.load_mson(GOLD)
    rng = np.random.default_rng(3)
    for mode in ("int", "corr"):
        fmode = capi.FEATURES_CORRELATIONS if mode == "corr" else capi.FEATURES_INTERACTIONS
        tab = ce.tables(np.diag([a.dim] * 3), feature_mode=fmode)
        cell = tab.supercell
        occ = neutral(cell, a.walkers, rng, cell.size // 2)
        for step, name in ((capi.STEP_SWAP, "swap"), (capi.STEP_FLIP, "flip")):
            cfg = capi.make_config(a.walkers, capi.KERNEL_METROPOLIS, step)
            eng = Engine(tab, cfg)
            eng.set_state(occ, np.arange(a.walkers, dtype=np.uint64) + np.uint64(11), a.temperature)
            eng.run(a.mc)
            eng.sync()
            ms = []
            for _ in range(3):
                eng.run(a.mc)
                ms.append(eng.last_kernel_ms())
            st = eng.get_state()
            print(json.dumps({"model": "LiNiO2 + Ewald (reference .mson)", "sites": int(cell.num_sites),
                              "walkers": a.walkers, "features": mode, "step": name,
                              "kernel": eng.kernel_info(), "kernel_ms": float(np.mean(ms)),
                              "steps_per_s": a.walkers * a.mc / (np.mean(ms) * 1e-3),
                              "acceptance": float(st["n_accepted"].sum() / st["n_steps"].sum())}), flush=True)
            eng.close()
    # through the smol-shaped API: charge-neutral TableFlip (flip table from the model's own
    # CompositionSpace) and semigrand flips under a SquareChargeBias
    from smol_amd import moca

    ens = moca.Ensemble.from_mson(ce, np.diag([a.dim] * 3))
    cell = ens.processor.supercell
    occ = neutral(cell, a.walkers, rng, cell.size // 2)
    for name, kw in (("table-flip", dict(step_type="table-flip")),
                     ("flip + square-charge bias", dict(step_type="flip", bias_type="square-charge",
                                                        bias_kwargs={"penalty": 0.5}))):
        sampler = moca.Sampler.from_ensemble(ens, temperature=a.temperature, nwalkers=a.walkers,
                                             seeds=list(range(a.walkers)), **kw)
        sampler.setup_sample(occ)
        eng = sampler.engine
        eng.run(a.mc)
        eng.sync()
        ms = []
        for _ in range(3):
            eng.run(a.mc)
            ms.append(eng.last_kernel_ms())
        st = eng.get_state(occupancy=False)
        print(json.dumps({"model": "LiNiO2 + Ewald (reference .mson)", "sites": int(cell.num_sites),
                          "walkers": a.walkers, "features": "int", "step": name,
                          "kernel": eng.kernel_info(), "kernel_ms": float(np.mean(ms)),
                          "steps_per_s": a.walkers * a.mc / (np.mean(ms) * 1e-3),
                          "acceptance": float(st["n_accepted"].sum() / st["n_steps"].sum())}), flush=True)


if __name__ == "__main__":
    main()
