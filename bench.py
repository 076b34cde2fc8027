#!/usr/bin/env python3
"""Benchmark of the MI355X ensemble Monte-Carlo hot path (BASELINE.json metric).

Workload (BASELINE.json configs[1], SURVEY.md §8d "Config 2"): binary FCC 16x16x16
primitive supercell (4096 sites), point + 4 pair + 2 triplet orbits (115 clusters per
site), canonical swap Metropolis, 4096 independent replica walkers per GPU, 50/50
composition, ECI U(-0.02, 0.02) eV (seed 20260928), T = 2500 K (acceptance ~0.38,
tuned once and frozen).  Feature trace = cluster-interaction vector (the reference's
default ClusterDecompositionProcessor), tracked every step.

A bench "step" = one launch of the engine advancing every walker MC_PER_STEP Metropolis
steps (one swap step = 2 attempted flips).  value = attempted flips/s over the whole job
(all ranks), inputs resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_REPLICAS = 4096  # per GPU (weak scaling: independent shards, no data-path collective)
MC_PER_STEP = 10000  # Metropolis steps per walker per bench step (launch)
TEMPERATURE = 2500.0
ALGO_BYTES_PER_FLIP = 56.0  # D*s_occ + p_acc*s_occ, D=55 distinct sites, int8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def build_workload():
    from smol_amd import capi, synth

    model = synth.build_cluster_model(synth.fcc_prim(a=4.09), {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [16, 16, 16])
    coefs = synth.random_coefs(model, seed=20260928, scale=0.02)
    tab = capi.TableSet.from_synth(sc, coefs, feature_mode=capi.FEATURES_INTERACTIONS)
    return model, sc, tab


def initial_occupancies(sc, first, count):
    occ = np.zeros((count, sc.num_sites), dtype=np.int32)
    for i in range(count):
        perm = np.random.default_rng(1000 + first + i).permutation(sc.num_sites)
        occ[i, perm[: sc.num_sites // 2]] = 1
    return occ


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box shows 256 hardware threads but grants a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


# HBM bytes per launch of the default configuration from the two separate PMC passes committed
# under profiles/ (r01_final_pmc.txt): FETCH_SIZE 50.4 MB + WRITE_SIZE 17.6 MB (KiB counters,
# summed over 4 dispatches / 4).  The occupancies go in and out once per launch (16.78 MB each
# way: the 4 B/lane occupancy stream reads back exactly in the undoubled counter, which is how
# MI355X_MICROARCH.md's x2 FETCH_SIZE correction for 16 B/lane streams was calibrated away for
# it); the remaining 33 MB of fetches are L2 misses of the 4 MB table of 32-bit index rows (read
# with 16 B/lane loads, so that part may be under-counted by up to 2x: <= 1.0e8 bytes in all).
# Everything else is LDS/L2 resident.
MEASURED_TRAFFIC_BYTES = {(4096, 10000): 6.81e7}


def cpu_baseline_child(seconds=12.0):
    """Runs inside a fresh interpreter (see cpu_baseline): the CPU oracle timed on the host."""
    cores = int(os.environ["OMP_NUM_THREADS"])
    from oracle import oracle as orc
    from smol_amd import capi

    model, sc, tab = build_workload()
    R = max(cores, 1) * 4
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
    mc = orc.OracleMC(tab, cfg)
    mc.set_state(initial_occupancies(sc, 0, R), np.arange(R, dtype=np.uint64) + np.uint64(12345),
                 TEMPERATURE)
    mc.run(2000)
    chunk, done, t0 = 5000, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        mc.run(chunk)
        done += chunk
    dt = time.perf_counter() - t0
    # (i) of SURVEY 8d: one walker on one thread, a few seconds
    one = orc.OracleMC(tab, capi.make_config(1, capi.KERNEL_METROPOLIS, capi.STEP_SWAP))
    one.set_state(initial_occupancies(sc, 0, 1), np.array([12345], dtype=np.uint64), TEMPERATURE)
    one.run(2000)
    n1, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < 3.0:
        one.run(20000)
        n1 += 20000
    single = 2.0 * n1 / (time.perf_counter() - t1)
    print(json.dumps({
        "value": 2.0 * R * done / dt,
        "single_thread_value": single,
        "unit": "attempted flips/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{R} walkers x {done} swap steps of the same 4096-site workload, "
                  f"OpenMP over walkers ({cores} bound threads), {dt:.1f} s",
    }))


def cpu_baseline():
    """The CPU oracle (port of the reference's compiled core + kernel logic) timed on the
    host cores of this box with OpenMP over walkers, on a bounded sample of the workload.
    It runs in a fresh interpreter: libgomp reads its environment once, when first loaded,
    and torch has already loaded it in this process (bound, passive-wait threads are ~2.3x
    faster than the defaults under this box's cgroup quota)."""
    import subprocess

    cores = usable_cores()
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="close", OMP_PLACES="cores",
               OMP_WAIT_POLICY="passive")
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child"],
                         env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        raise RuntimeError("cpu baseline failed: " + out.stderr[-2000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--replicas", type=int, default=N_REPLICAS)
    ap.add_argument("--mc-per-step", type=int, default=MC_PER_STEP)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        cpu_baseline_child()
        return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (the engine has no CPU fallback)")
    # one process per GPU; if the launcher masks devices per rank only device 0 is visible
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from smol_amd import capi
    from smol_amd.engine import Engine

    model, sc, tab = build_workload()
    R = args.replicas
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP, device=local_rank)
    eng = Engine(tab, cfg)
    first = rank * R
    seeds = (np.arange(first, first + R, dtype=np.uint64) + np.uint64(12345))
    eng.set_state(initial_occupancies(sc, first, R), seeds, TEMPERATURE)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.run(args.mc_per_step)
    eng.sync()
    s0 = eng.get_state(occupancy=False)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        eng.run(args.mc_per_step)
        kernel_ms.append(eng.last_kernel_ms())  # HIP events on the launch stream
    eng.sync()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    s1 = eng.get_state(occupancy=False)
    acc_local = float((s1["n_accepted"] - s0["n_accepted"]).sum())
    stats = torch.tensor(
        [acc_local, float(s1["enthalpy"].sum()), float((s1["enthalpy"] ** 2).sum()), float(R)],
        dtype=torch.float64, device="cuda",
    )
    if world > 1:  # global averages: the only collective on this path (RCCL all-reduce)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    stats = stats.cpu().numpy()

    if rank == 0:
        total_steps = float(args.steps) * args.mc_per_step * R * world
        flips = 2.0 * total_steps
        value = flips / dt
        k_ms = float(np.mean(kernel_ms))
        flips_per_launch = 2.0 * args.mc_per_step * R
        achieved = flips_per_launch * ALGO_BYTES_PER_FLIP / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "attempted MC flips/s (node) + ns/flip/replica, 4096-site FCC canonical",
            "value": value,
            "unit": "attempted flips/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "binary FCC 16x16x16 (4096 sites), point+4 pair+2 triplet CE, canonical "
                            "swap Metropolis, cluster-interaction trace, T=2500K",
                "replicas_per_gpu": R,
                "mc_steps_per_replica_per_step": args.mc_per_step,
                "parallelism": f"replica-shard x{world}",
            },
            "ns_per_flip_per_replica": dt / (2.0 * args.steps * args.mc_per_step) * 1e9,
            "mc_steps_per_s": total_steps / dt,
            "acceptance_ratio": stats[0] / (args.steps * args.mc_per_step * stats[3]),
            "mean_enthalpy_per_site_eV": stats[1] / stats[3] / sc.num_sites,
            "roofline": {
                "bound": "hbm",
                "kernel": "mc_lean_kernel<NSLOT=2,MM=2,SWAP,noMU>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": MEASURED_TRAFFIC_BYTES.get((R, args.mc_per_step)),
                "traffic_unit": "bytes per launch (rocprofv3 PMC, profiles/)",
                "algorithmic_bytes_per_launch": flips_per_launch * ALGO_BYTES_PER_FLIP,
                "kernel_ms_avg": k_ms,
                "algorithmic_bytes_per_flip": ALGO_BYTES_PER_FLIP,
                "lds_gathers_per_s": flips_per_launch * 174.0 / (k_ms * 1e-3),
                "note": "CE flips are LDS/L2-latency bound by construction (SURVEY 8d); the "
                        "HBM fraction is reported as the contract asks, the gather rate beside it",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
                out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            except Exception as exc:  # the GPU measurement must not be lost with it
                out["cpu_baseline"] = {"value": None, "unit": "attempted flips/s", "cores": usable_cores(),
                                       "kind": "port", "sample": f"failed: {exc}"[:300]}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
