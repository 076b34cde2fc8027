#!/usr/bin/env python3
"""Benchmark of the MI355X ensemble Monte-Carlo hot path (BASELINE.json metric).

Workload (BASELINE.json configs[1], SURVEY.md §8d "Config 2"): binary FCC 16x16x16
primitive supercell (4096 sites), point + 4 pair + 2 triplet orbits (115 clusters per
site), canonical swap Metropolis, independent replica walkers, 50/50 composition, ECI
U(-0.02, 0.02) eV (seed 20260928), T = 2500 K (acceptance ~0.38, tuned once and frozen).
Feature trace = cluster-interaction vector (the reference's default
ClusterDecompositionProcessor), tracked every step.

A bench "step" = one launch of the engine advancing every walker MC_PER_STEP Metropolis
steps (one swap step = 2 attempted flips).  value = attempted flips/s over the whole job
(all ranks), inputs resident in HBM before the timed region.

    python bench.py                                  # 1 GPU, defaults
    python bench.py --gpus 8 --steps 20 --warmup 5   # spawns one rank per GPU itself
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 ...   # also fine

--scaling weak   (default) 4096 walkers per GPU: per-GPU work fixed, job grows with N
--scaling strong 4096 walkers in total (north_star's "at 4096 replicas"): 4096/N per GPU
--dry-run        no GPU: the N-rank launch, rendezvous (gloo), barriers and the all-reduce of
                 the statistics run on CPU with made-up numbers; proves the launcher, prints a
                 line with "dry_run": true and value null
--oversubscribe  ranks share the visible GPUs round-robin and reduce over gloo (lets the N-rank
                 path run on a 1-GPU box; the value is then not a scaling measurement)
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_REPLICAS = 4096  # per GPU (weak) or in total (strong)
MC_PER_STEP = 10000  # Metropolis steps per walker per bench step (launch)
ALGO_BYTES_PER_FLIP = 56.0  # D*s_occ + p_acc*s_occ, D=55 distinct sites, int8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md: aggregate L2 bandwidth
METRIC = "attempted MC flips/s (node) + ns/flip/replica, 4096-site FCC canonical"
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_constants.json")


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def pmc_constants():
    """Per-launch / per-step counter figures of the headline kernel taken from the rocprofv3
    PMC passes committed under profiles/ (written by tools/pmc_to_json.py from the rocpd
    databases; counters cannot be read from inside an un-profiled run).  Keyed by
    "R x mc_per_step"."""
    try:
        return json.load(open(PMC_FILE))
    except (OSError, ValueError):
        return {}


# --------------------------------------------------------------------------------------
# CPU baseline (oracle timed on the host cores; fresh interpreter)
# --------------------------------------------------------------------------------------
def cpu_baseline_child(seconds=12.0):
    """Runs inside a fresh interpreter (see cpu_baseline): the CPU oracle timed on the host.
    Uses the -ffast-math build of the oracle (the reference's flag set, setup.py:18-25) when it
    has been built; the replay tests keep the IEEE build."""
    cores = int(os.environ["OMP_NUM_THREADS"])
    from oracle import oracle as orc
    from smol_amd import capi, workloads

    flags = orc.use_fast_math_build() if hasattr(orc, "use_fast_math_build") else "-O3 -fopenmp"
    R = max(cores, 1) * 4
    wl = workloads.config2(0, R)
    cfg = wl.make_config()
    mc = orc.OracleMC(wl.tables, cfg)
    mc.set_state(wl.occupancy, wl.seeds, wl.temperature)
    mc.run(2000)
    chunk, done, t0 = 5000, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        mc.run(chunk)
        done += chunk
    dt = time.perf_counter() - t0
    # (i) of SURVEY 8d: one walker on one thread, a few seconds
    one = orc.OracleMC(wl.tables, capi.make_config(1, capi.KERNEL_METROPOLIS, capi.STEP_SWAP))
    one.set_state(wl.occupancy[:1], wl.seeds[:1], wl.temperature)
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
        "cpu_model": cpu_model(),
        "hardware_threads": os.cpu_count(),
        "kind": "port",
        "build_flags": flags,
        "sample": f"{R} walkers x {done} swap steps of the same 4096-site workload, "
                  f"OpenMP over walkers ({cores} bound threads), {dt:.1f} s",
    }))


def cpu_baseline():
    """The CPU oracle (port of the reference's compiled core + kernel logic) timed on the
    host cores of this box with OpenMP over walkers, on a bounded sample of the workload.
    It runs in a fresh interpreter: libgomp reads its environment once, when first loaded,
    and torch has already loaded it in this process (bound, passive-wait threads are ~2.3x
    faster than the defaults under this box's cgroup quota)."""
    cores = usable_cores()
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="close", OMP_PLACES="cores",
               OMP_WAIT_POLICY="passive")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child"],
                         env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        raise RuntimeError("cpu baseline failed: " + out.stderr[-2000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


# --------------------------------------------------------------------------------------
# launcher: python bench.py --gpus N  ->  N ranks, one per GPU
# --------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """Spawn one worker process per rank with the torchrun environment contract
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) and relay rank 0's JSON line.
    Plain subprocesses rather than torch.multiprocessing: every rank is a fresh interpreter that
    initialises its own HIP context, exactly as under torch.distributed.run."""
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SMOLMC_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + argv, env=env,
            stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit(bad[0][1] if bad[0][1] > 0 else 1)


# --------------------------------------------------------------------------------------
# other configurations (after the headline's timed region; rank 0, N=1 only)
# --------------------------------------------------------------------------------------
def time_other_configs(device, launches=3):
    """Configs 1, 3, 4, 5 of BASELINE.json: a few launches each, kernel time from HIP events on
    the launch stream, with the roofline that actually bounds each."""
    from smol_amd import capi, parallel, workloads
    from smol_amd.engine import Engine

    out = []

    def run(wl, extra):
        eng = Engine(wl.tables, wl.make_config(device))
        eng.set_state(wl.occupancy, wl.seeds, wl.temperature)
        eng.run(wl.mc_per_launch, sync=True)
        s0 = eng.get_state(occupancy=False)
        ms = []
        for _ in range(launches):
            eng.run(wl.mc_per_launch, sync=True)
            ms.append(eng.last_kernel_ms())
        s1 = eng.get_state(occupancy=False)
        k_ms = float(np.mean(ms))
        steps = wl.n_walkers * wl.mc_per_launch
        acc = float((s1["n_accepted"] - s0["n_accepted"]).sum()) / (launches * steps)
        rec = dict(config=wl.name, kernel=eng.kernel_info(), replicas=wl.n_walkers,
                   mc_steps_per_launch=wl.mc_per_launch, kernel_ms=k_ms,
                   mc_steps_per_s=steps / (k_ms * 1e-3),
                   flips_per_s=wl.flips_per_step * steps / (k_ms * 1e-3), acceptance=acc)
        rec["roofline"] = extra(rec, wl, eng)
        eng.close()
        out.append(rec)

    def hbm_ce(rec, wl, eng):  # 56 algorithmic bytes per CE flip
        a = rec["flips_per_s"] * ALGO_BYTES_PER_FLIP / 1e9
        return dict(bound="hbm", achieved=a, peak=HBM_PEAK_GBS, unit="GB/s", frac=a / HBM_PEAK_GBS,
                    note="CE flip, occupancy LDS-resident: nominal HBM fraction (56 B/flip)")

    run(workloads.config1(), hbm_ce)

    def ewald_field(rec, wl, eng):
        # potential-field formulation: a proposal reads O(1) LDS words; only an ACCEPTED flip
        # streams one row of the site kernel G (n_act doubles, from L2 / Infinity Cache: the
        # 48 MB kernel exceeds the 32 MB of L2) and read-modify-writes phi in LDS.  The
        # 2-rows-per-proposal figure of SURVEY 8d does not describe this algorithm.
        n_act = wl.sc.size
        row_bytes = n_act * 8.0
        a = rec["flips_per_s"] * rec["acceptance"] * row_bytes / 1e9
        return dict(bound="hbm", achieved=a, peak=HBM_PEAK_GBS, unit="GB/s", frac=a / HBM_PEAK_GBS,
                    l2_peak=L2_PEAK_GBS, acceptance=rec["acceptance"], row_bytes_per_accepted_flip=row_bytes,
                    note="accepted-flip row traffic (n_act*8 B each, served by L2/MALL) over kernel "
                         "time; depends on the acceptance; the dense two-row formulation "
                         "(58752 B/flip) is HBM-capped at 1.36e8 flips/s")

    wl3 = workloads.config3()
    run(wl3, ewald_field)

    # config 4: the window is centred on the starting enthalpy, evaluated on the engine
    wl4 = workloads.config4()
    probe = Engine(wl4.tables, capi.make_config(1, device=device))
    h0 = float(probe.natural_parameters @ probe.eval_full(wl4.occupancy[:1])[0])
    probe.close()
    run(workloads.config4(h0=h0), hbm_ce)

    # config 5: TableFlip + exchange ladder (single rank: decisions on the host, temperatures move)
    wl5 = workloads.config5()
    eng = Engine(wl5.tables, wl5.make_config(device))
    eng.set_state(wl5.occupancy, wl5.seeds, wl5.temperature)
    rex = parallel.ReplicaExchange(wl5.extras["ladder"], wl5.n_walkers, seed=11)
    parallel.run_replica_exchange(eng, rex, 1, wl5.mc_per_launch)
    s0 = eng.get_state(occupancy=False)
    eng.sync()
    t1 = time.perf_counter()
    kms = []
    for _ in range(launches):
        parallel.run_replica_exchange(eng, rex, 1, wl5.mc_per_launch)
        kms.append(eng.last_kernel_ms())
    eng.sync()
    wall = time.perf_counter() - t1
    s1 = eng.get_state(occupancy=False)
    steps = wl5.n_walkers * wl5.mc_per_launch * launches
    acc = float((s1["n_accepted"] - s0["n_accepted"]).sum()) / steps
    k_ms = float(np.mean(kms))
    out.append(dict(
        config=wl5.name + f", exchange every {wl5.mc_per_launch} steps", kernel=eng.kernel_info(),
        replicas=wl5.n_walkers, mc_steps_per_launch=wl5.mc_per_launch, kernel_ms=k_ms,
        mc_steps_per_s=steps / wall, mc_steps_per_s_kernel_only=steps / launches / (k_ms * 1e-3),
        acceptance=acc, exchange_acceptance_mean=float(rex.acceptance.mean()),
        roofline=dict(bound="issue", note="TableFlip proposal is scalar-issue bound (DESIGN.md §5); "
                                          "no byte roofline applies; wall time includes the exchange")))
    eng.close()
    return out


# --------------------------------------------------------------------------------------
def dry_run(args, rank, world):
    """The multi-rank control flow without a GPU (gloo): rendezvous, barrier, timed loop of
    no-ops, MAX-reduce of the time, SUM-reduce of the statistics."""
    import torch
    import torch.distributed as dist

    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from smol_amd import parallel

    first, count = (rank * args.replicas, args.replicas) if args.scaling == "weak" else \
        parallel.shard(args.replicas, rank, world)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * args.steps)
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64)
    stats = torch.tensor([0.0, 0.0, 0.0, float(count)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({
            "metric": METRIC, "value": None, "unit": "attempted flips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
            "scaling": args.scaling, "dry_run": True, "walkers_total": int(stats[3].item()),
            "walkers_rank0": [first, count],
        }))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--replicas", type=int, default=N_REPLICAS,
                    help="walkers per GPU (weak) or in total (strong)")
    ap.add_argument("--mc-per-step", type=int, default=MC_PER_STEP)
    ap.add_argument("--features", choices=("interactions", "correlations"), default="interactions")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--oversubscribe", action="store_true")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        cpu_baseline_child()
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args, sys.argv[1:])
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: pass --gpus equal to the "
                         "number of ranks (or run python bench.py --gpus N, which spawns them)")
    if args.dry_run:
        dry_run(args, rank, world)
        return

    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev == 0 or (n_dev < world and not args.oversubscribe):
        # prove the launch + rendezvous first, then say what is missing
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            sys.stderr.write(
                f"bench.py: needs {world} AMD GPU(s), found {n_dev} (the engine has no CPU fallback)."
                f"  The {world}-rank launch and rendezvous work (verified over gloo);"
                " use --dry-run to exercise the multi-rank control flow without GPUs.\n")
        raise SystemExit(3)
    # one process per GPU; if the launcher masks devices per rank only device 0 is visible
    device = local_rank % n_dev
    torch.cuda.set_device(device)
    backend = "gloo" if args.oversubscribe else "nccl"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from smol_amd import capi, parallel, workloads
    from smol_amd.engine import Engine

    if args.scaling == "weak":
        first, R = rank * args.replicas, args.replicas
        total_walkers = args.replicas * world
    else:
        first, R = parallel.shard(args.replicas, rank, world)
        total_walkers = args.replicas
    mode = capi.FEATURES_INTERACTIONS if args.features == "interactions" else capi.FEATURES_CORRELATIONS
    wl = workloads.config2(first, R, feature_mode=mode, mc=args.mc_per_step)
    eng = Engine(wl.tables, wl.make_config(device))
    eng.set_state(wl.occupancy, wl.seeds, wl.temperature)
    red_dev = "cuda" if backend == "nccl" else "cpu"

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
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    s1 = eng.get_state(occupancy=False)
    acc_local = float((s1["n_accepted"] - s0["n_accepted"]).sum())
    stats = torch.tensor(
        [acc_local, float(s1["enthalpy"].sum()), float((s1["enthalpy"] ** 2).sum()), float(R),
         float(np.mean(kernel_ms))],
        dtype=torch.float64, device=red_dev,
    )
    # global averages: the only collective on this path (RCCL all-reduce)
    stats = parallel.global_sums(stats).cpu().numpy()

    if rank == 0:
        n_walk = stats[3]
        total_steps = float(args.steps) * args.mc_per_step * n_walk
        flips = 2.0 * total_steps
        value = flips / dt
        k_ms = float(np.mean(kernel_ms))
        flips_per_launch = 2.0 * args.mc_per_step * R
        achieved = flips_per_launch * ALGO_BYTES_PER_FLIP / (k_ms * 1e-3) / 1e9
        pmc = pmc_constants().get(f"{R}x{args.mc_per_step}", {}) if args.features == "interactions" else {}
        roof = {
            "bound": "hbm",
            "kernel": eng.kernel_info(),
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc.get("hbm_bytes_per_launch"),
            "traffic_unit": "bytes per launch (rocprofv3 PMC passes, " + pmc.get("source", "none for this shape") + ")",
            "algorithmic_bytes_per_launch": flips_per_launch * ALGO_BYTES_PER_FLIP,
            "kernel_ms_avg": k_ms,
            "algorithmic_bytes_per_flip": ALGO_BYTES_PER_FLIP,
            "lds_gathers_per_s": flips_per_launch * 174.0 / (k_ms * 1e-3),
            "note": "CE flips are LDS/L2-latency bound by construction (SURVEY 8d); the HBM "
                    "fraction is reported as the contract asks; the binding ceiling is VALU issue "
                    "(valu_issue_frac, from the PMC passes of the same build)",
        }
        for k in ("valu_issue_frac", "valu_per_step", "salu_per_step", "lds_per_step",
                  "vmem_per_step", "wave_cycles_per_step", "waves_per_simd"):
            if k in pmc:
                roof[k] = pmc[k]
        out = {
            "metric": METRIC,
            "value": value,
            "unit": "attempted flips/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": wl.name,
                "replicas_per_gpu": R,
                "replicas_total": int(n_walk),
                "value_definition": (
                    "weak: 4096 walkers on every GPU, value = 2*steps*mc*4096*N / time"
                    if args.scaling == "weak" else
                    "strong: 4096 walkers in total, 4096/N per GPU, value = 2*steps*mc*4096 / time"),
                "mc_steps_per_replica_per_step": args.mc_per_step,
                "parallelism": f"replica-shard x{world}" + (" (oversubscribed, gloo)" if args.oversubscribe else ""),
            },
            "ns_per_flip_per_replica": dt / (2.0 * args.steps * args.mc_per_step) * 1e9,
            "mc_steps_per_s": total_steps / dt,
            "acceptance_ratio": stats[0] / (args.steps * args.mc_per_step * n_walk),
            "mean_enthalpy_per_site_eV": stats[1] / n_walk / wl.sc.num_sites,
            "kernel_ms_mean_over_ranks": stats[4] / world,
            "roofline": roof,
        }
        assert int(n_walk) == total_walkers
        eng.close()
        if world == 1 and not args.no_other_configs:
            try:
                out["other_configs"] = time_other_configs(device)
            except Exception as exc:  # the headline measurement must not be lost with it
                out["other_configs"] = [{"error": f"{type(exc).__name__}: {exc}"[:300]}]
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
                out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            except Exception as exc:  # the GPU measurement must not be lost with it
                out["cpu_baseline"] = {"value": None, "unit": "attempted flips/s", "cores": usable_cores(),
                                       "kind": "port", "sample": f"failed: {exc}"[:300]}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
