#!/usr/bin/env python3
"""Throughput of the non-headline BASELINE.json configs on one GPU (kernel time from HIP
events on the launch stream).  bench.py stays the contract benchmark (config 2); this
script measures configs 1, 3 and 4 so DESIGN.md can quote them.

    python tools/bench_configs.py --config 3 [--replicas 2048] [--mc 200] [--launches 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from smol_amd import capi, ewald, synth  # noqa: E402
from smol_amd.engine import Engine  # noqa: E402


def rand_occ(sc, R, seed, balanced=False):
    prim = sc.model.prim
    nsp = np.array([prim.nspecies[b] for b in sc.site_b])
    rng = np.random.default_rng(seed)
    if balanced:
        occ = np.zeros((R, sc.num_sites), np.int32)
        act = np.flatnonzero(nsp > 1)
        for r in range(R):
            occ[r, rng.permutation(act)[: len(act) // 2]] = 1
        return occ
    return (rng.random((R, sc.num_sites)) * nsp).astype(np.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, required=True)
    ap.add_argument("--replicas", type=int, default=0)
    ap.add_argument("--mc", type=int, default=0)
    ap.add_argument("--launches", type=int, default=3)
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--temperature", type=float, default=0.0, help="override the configuration's temperature")
    a = ap.parse_args()
    t0 = time.time()
    if a.config == 1:
        # binary FCC conventional 4x4x4 (256 sites), pairs only, canonical swap
        model = synth.build_cluster_model(synth.fcc_conventional_prim(), {2: 6.0})
        sc = synth.build_supercell(model, [a.dim or 4] * 3)
        tab = capi.TableSet.from_synth(sc, synth.random_coefs(model))
        R, mc = a.replicas or 4096, a.mc or 5000
        cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
        occ, T, flips_per_step = rand_occ(sc, R, 1, True), 2500.0, 2
        name = "config1: binary FCC conventional 4x4x4, pairs, canonical swap"
    elif a.config == 8:
        # (not in BASELINE.json) a larger expansion on the config-2 lattice: pairs to 6.5 A and
        # triplets to 5.2 A = 451 clusters per site -> mc_lean_multi_kernel with 8 slots per lane
        model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.5, 3: 5.2})
        sc = synth.build_supercell(model, [a.dim or 16] * 3)
        tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=20260928))
        R, mc = a.replicas or 4096, a.mc or 2000
        cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
        occ, T, flips_per_step = rand_occ(sc, R, 1, True), 2500.0, 2
        name = "config8: binary FCC 16^3 (4096 sites), pairs <= 6.5 A + triplets <= 5.2 A (451 clusters/site), canonical swap"
    elif a.config == 3:
        # ternary rocksalt 12^3 + Ewald, semigrand flip with mu table
        d = a.dim or 12
        model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 6.0, 3: 5.0})
        sc = synth.build_supercell(model, [d] * 3)
        ew = ewald.supercell_ewald(sc)
        mu = np.zeros((sc.num_sites, 3))
        mu[: sc.size] = np.random.default_rng(7).uniform(-0.5, 0.5, 3)[None, :]
        tab = capi.TableSet.from_synth(sc, synth.random_coefs(model), ewald=ew, ewald_coef=0.1,
                                       mu_table=mu)
        R, mc = a.replicas or 2048, a.mc or 2000
        cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_FLIP)
        occ, T, flips_per_step = rand_occ(sc, R, 3), 3000.0, 1
        name = f"config3: ternary rocksalt {d}^3 ({sc.num_sites} sites), triplet CE + Ewald, semigrand flip"
    elif a.config == 6:
        # (not in BASELINE.json) config-3 lattice, canonical swap with the Ewald term: the common
        # production case for ionic systems at fixed composition
        d = a.dim or 12
        model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 6.0, 3: 5.0})
        sc = synth.build_supercell(model, [d] * 3)
        ew = ewald.supercell_ewald(sc)
        tab = capi.TableSet.from_synth(sc, synth.random_coefs(model), ewald=ew, ewald_coef=0.1)
        R, mc = a.replicas or 2048, a.mc or 2000
        cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
        occ, T, flips_per_step = rand_occ(sc, R, 3), 3000.0, 2
        name = f"config6: ternary rocksalt {d}^3 ({sc.num_sites} sites), triplet CE + Ewald, canonical swap"
    elif a.config == 7:
        # (not in BASELINE.json) two ACTIVE sublattices: Li+/Mn3+/Ti4+ cations and O2-/F- anions
        # on rocksalt (the disordered-rocksalt oxyfluoride shape), CE (+ Ewald), canonical swap
        d = a.dim or 12
        model = synth.build_cluster_model(synth.rocksalt_prim(anion_charges=(-2.0, -1.0)), {2: 6.0, 3: 4.5})
        sc = synth.build_supercell(model, [d] * 3)
        ew = ewald.supercell_ewald(sc) if not os.environ.get("CONFIG7_NO_EWALD") else None
        tab = capi.TableSet.from_synth(sc, synth.random_coefs(model), ewald=ew, ewald_coef=0.1)
        R, mc = a.replicas or 2048, a.mc or 1000
        cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
        occ, T, flips_per_step = rand_occ(sc, R, 3), 3000.0, 2
        name = (f"config7: rocksalt {d}^3 ({sc.num_sites} sites), ternary cations + binary anions, CE"
                f"{' + Ewald' if ew is not None else ''}, canonical swap")
    elif a.config == 4:
        # config-2 Hamiltonian, Wang-Landau, 1024 walkers
        model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.0, 3: 5.0})
        sc = synth.build_supercell(model, [a.dim or 16] * 3)
        tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=20260928))
        R, mc = a.replicas or 1024, a.mc or 5000
        occ = rand_occ(sc, R, 4, True)
        probe = Engine(tab, capi.make_config(1))
        h0 = float(probe.natural_parameters @ probe.eval_full(occ[:1])[0])
        probe.close()
        cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, capi.STEP_SWAP, min_enthalpy=h0 - 160.0,
                               max_enthalpy=h0 + 96.0, bin_size=0.5, flatness=0.8, check_period=1000)
        T, flips_per_step = 0.0, 2
        name = "config4: binary FCC 16^3 pair+triplet, Wang-Landau swap, 512 bins"
    elif a.config == 5:
        # config-3 lattice, charge-neutral TableFlip (3 Mn3+ <-> Li+ + 2 Ti4+) + replica-exchange
        # ladder (one rank here; bench.py-style multi-rank launch shards the ladder)
        import torch

        from smol_amd import parallel

        d = a.dim or 12
        model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 6.0, 3: 5.0})
        sc = synth.build_supercell(model, [d] * 3)
        ew = ewald.supercell_ewald(sc)
        mu = np.zeros((sc.num_sites, 3))
        mu[: sc.size] = np.random.default_rng(7).uniform(-0.5, 0.5, 3)[None, :]
        tab = capi.TableSet.from_synth(sc, synth.random_coefs(model), ewald=ew, ewald_coef=0.1,
                                       mu_table=mu, flip_table=[[1, -3, 2]], swap_weight=0.1)
        R, mc = a.replicas or 2048, a.mc or sc.num_sites
        cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP)
        P = sc.size
        n_ti = 2 * (P // 12)  # neutral: 2 n_Mn + 3 n_Ti = P
        n_mn = (P - 3 * n_ti) // 2
        rng = np.random.default_rng(5)
        occ = np.zeros((R, sc.num_sites), np.int32)
        for r in range(R):
            perm = rng.permutation(P)
            occ[r, perm[:n_mn]] = 1
            occ[r, perm[n_mn:n_mn + n_ti]] = 2
        ladder = parallel.geometric_ladder(400.0, 2000.0, R)
        eng = Engine(tab, cfg)
        eng.set_state(occ, np.arange(R, dtype=np.uint64) + np.uint64(777), ladder)
        rex = parallel.ReplicaExchange(ladder, R, seed=11)
        parallel.run_replica_exchange(eng, rex, 1, mc)
        s0 = eng.get_state(occupancy=False)
        t1 = time.time()
        parallel.run_replica_exchange(eng, rex, a.launches, mc)
        eng.sync()
        wall = time.time() - t1
        s1 = eng.get_state(occupancy=False)
        steps = R * mc * a.launches
        print(json.dumps(dict(
            config=f"config5: ternary rocksalt {d}^3 + Ewald, charge-neutral TableFlip, replica-exchange "
                   f"ladder 400-2000 K over {R} walkers, exchange every {mc} steps",
            replicas=R, mc_steps_between_exchanges=mc, exchanges=a.launches, wall_s=wall,
            mc_steps_per_s=steps / wall, kernel_ms_last=eng.last_kernel_ms(),
            acceptance=float((s1["n_accepted"] - s0["n_accepted"]).sum()) / steps,
            exchange_acceptance_mean=float(rex.acceptance.mean()),
        )))
        return
    else:
        raise SystemExit("config must be 1, 3, 4, 5, 6 or 7")
    setup_s = time.time() - t0
    eng = Engine(tab, cfg)
    if a.temperature > 0.0:
        T = a.temperature
    eng.set_state(occ, np.arange(R, dtype=np.uint64) + np.uint64(777), T)
    eng.run(mc, sync=True)
    ms = []
    s0 = eng.get_state(occupancy=False)
    for _ in range(a.launches):
        eng.run(mc, sync=True)
        ms.append(eng.last_kernel_ms())
    s1 = eng.get_state(occupancy=False)
    k_ms = float(np.mean(ms))
    steps = R * mc
    out = dict(
        config=name, replicas=R, mc_steps_per_launch=mc, kernel_ms=k_ms, setup_s=setup_s,
        mc_steps_per_s=steps / (k_ms * 1e-3), flips_per_s=flips_per_step * steps / (k_ms * 1e-3),
        us_per_step_per_walker=k_ms * 1e3 / mc,
        acceptance=float((s1["n_accepted"] - s0["n_accepted"]).sum()) / (a.launches * steps),
        lean_kernel=bool(os.environ.get("SMOLMC_FORCE_GENERAL") is None),
    )
    if a.config == 3:
        N = sc.num_sites
        bytes_per_flip = 2 * N * 8 + N  # two matrix rows gathered at N indices + occupancy
        out["ewald_algorithmic_GBs"] = out["flips_per_s"] * bytes_per_flip / 1e9
        out["ewald_hbm_frac"] = out["ewald_algorithmic_GBs"] / 8000.0
    print(json.dumps(out))


if __name__ == "__main__":
    main()
