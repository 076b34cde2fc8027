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
# Metropolis steps per walker per bench step (launch).  SURVEY 8d asks for >= 1e6 timed steps per
# replica and the timed region should last >= 1 s: the driver's --steps 20 gives 2.5e6 steps per
# replica and ~1.2 s of kernel time on one MI355X (r2 timed 2e5 steps = 93 ms).
MC_PER_STEP = 125000
ALGO_BYTES_PER_FLIP = 56.0  # D*s_occ + p_acc*s_occ, D=55 distinct sites, int8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_MEASURED_FALLBACK_GBS = 6000.0  # tools/hbm_triad.py on this pool (read 6.0, triad 5.9 TB/s), used when the live probe fails
L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md: aggregate L2 bandwidth
METRIC = "attempted MC flips/s (node) + ns/flip/replica, 4096-site FCC canonical"
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_constants.json")


def _xcd_count(cus):
    """XCDs of the GPUs of this node: `num_xcc` of the KFD topology (simd_count > 0 marks a GPU node); when the
    topology is not readable, one XCD per 32 CUs (MI355X: 256 CUs in 8 XCDs)."""
    import glob

    best = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            kv = dict(ln.split()[:2] for ln in open(f) if len(ln.split()) >= 2)
            if int(kv.get("simd_count", 0)) > 0:
                best = max(best, int(kv.get("num_xcc", 0)))
        except (OSError, ValueError):
            pass
    return best or max(1, cus // 32)


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


def measure_hbm_peak(n=1 << 28):
    """Achievable HBM bandwidth of this GPU, measured live (tools/hbm_triad.py in short): read and
    triad over float64 arrays of 2 GiB each -- far beyond the 256 MiB Infinity Cache.  torch is
    only the allocator / launcher of the elementwise kernels.  Returns GB/s figures or None."""
    try:
        import torch

        a = torch.empty(n, dtype=torch.float64, device="cuda")
        b = torch.rand(n, dtype=torch.float64, device="cuda")
        c = torch.rand(n, dtype=torch.float64, device="cuda")

        def timed(fn, reps=10):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / reps

        out = {"read_GBs": n * 8 / timed(lambda: b.sum()) / 1e9,
               "triad_GBs": 3 * n * 8 / timed(lambda: torch.add(b, c, alpha=3.0, out=a)) / 1e9}
        del a, b, c
        torch.cuda.empty_cache()
        return out
    except Exception:  # the probe must never cost the benchmark line
        return None


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
    # BASELINE configs[0], "1 replica on smol CPU path": the 256-site pair-only model, one walker,
    # one thread (the arithmetic of the reference's compiled core + kernel logic; the reference's
    # own end-to-end Python rate for such a chain is 2e3 - 1e4 steps/s)
    w1 = workloads.config1(0, 1)
    c1 = orc.OracleMC(w1.tables, capi.make_config(1, capi.KERNEL_METROPOLIS, capi.STEP_SWAP))
    c1.set_state(w1.occupancy[:1], w1.seeds[:1], w1.temperature)
    c1.run(2000)
    m1, t2 = 0, time.perf_counter()
    while time.perf_counter() - t2 < 3.0:
        c1.run(20000)
        m1 += 20000
    t2 = time.perf_counter() - t2
    print(json.dumps({
        "value": 2.0 * R * done / dt,
        "single_thread_value": single,
        "config1_single_replica": {
            "value": 2.0 * m1 / t2, "unit": "attempted flips/s", "ns_per_flip": t2 / (2.0 * m1) * 1e9,
            "cores": 1, "sample": f"config 1 (256 sites, pair-only), 1 walker x {m1} swap steps, 1 thread, {t2:.1f} s"},
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
    # rank 0's stdout is drained by a thread while ALL children are polled: a rank that dies early
    # (bad device, import error) ends the job at once with its code, instead of leaving rank 0 in the
    # rendezvous until the store timeout
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = (r, p.returncode)
                break
        time.sleep(0.05)
    if failed is None:
        failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    if chunks and chunks[0]:
        sys.stdout.write(chunks[0])
        sys.stdout.flush()
    if failed is not None:
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with code {failed[1]}\n")
        raise SystemExit(failed[1] if failed[1] > 0 else 1)


# --------------------------------------------------------------------------------------
# other configurations (after the headline's timed region)
#   one rank : configs 1, 3, 4, 5
#   N ranks  : configs 4 and 5, the two BASELINE.json defines as N-rank workloads -- config 4 as
#              1024 independent Wang-Landau walkers per rank, config 5 as ONE temperature ladder
#              over 2048 N walkers whose exchange step (all-gather of 8 B per walker + temperature
#              moves, smol_amd/parallel.py) runs inside the timed region.  All ranks take part,
#              rank 0 reports.
# --------------------------------------------------------------------------------------
OTHER_CONFIGS_TIMEOUT_S = 300
EQUIL_STEPS = {3: 600_000, 5: 400_000}  # untimed steps per walker before the steady-state figures


class _Clock:
    """Barrier-bracketed wall clock, MAX over ranks (the bench contract's timing rule)."""

    def __init__(self, world, red_dev, gpu=True):
        self.world, self.red_dev, self.gpu = world, red_dev, gpu

    def barrier(self):
        import torch
        import torch.distributed as dist

        if self.gpu:
            torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        if self.gpu:
            torch.cuda.synchronize()

    def time(self, fn):
        import torch
        import torch.distributed as dist

        self.barrier()
        t0 = time.perf_counter()
        fn()
        self.barrier()
        dt = time.perf_counter() - t0
        if self.world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=self.red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    def sum(self, values):
        import torch
        from smol_amd import parallel

        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=self.red_dev)
        return parallel.global_sums(t).cpu().numpy()

    def max(self, values):
        import torch
        import torch.distributed as dist

        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=self.red_dev)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.cpu().numpy()


def issue_roof(key, note):
    """Roofline of a configuration whose kernel is bound on-chip: the VALU-issue and LDS-issue
    fractions of its rocprofv3 PMC passes (profiles/pmc_constants.json, entry `key`, written by
    tools/run_profile_configs.sh on the build whose source digest it carries) and the HBM bytes per
    launch the same passes counted.  No modelled byte rate: these kernels keep their working set in
    LDS / L2."""
    def roof(rec):
        from smol_amd import codeobj

        pmc = pmc_constants().get(key, {})
        out = dict(bound="issue", unit="fraction of issue cycles", note=note,
                   pmc_source=pmc.get("source", "no PMC pass for this configuration"),
                   pmc_stale=codeobj.isa_stale(pmc))
        for k in ("valu_issue_frac", "lds_issue_frac", "valu_per_step", "salu_per_step", "lds_per_step", "vmem_per_step",
                  "wave_cycles_per_step", "waves_per_simd", "lds_bank_conflict_per_step", "hbm_bytes_per_launch"):
            if k in pmc:
                out[k] = pmc[k]
        if "valu_issue_frac" in pmc:
            out["achieved"], out["peak"], out["frac"] = pmc["valu_issue_frac"], 1.0, pmc["valu_issue_frac"]
        return out
    return roof


def hbm_ce(rec):  # 56 algorithmic bytes per CE flip
    a = rec["flips_per_s"] * ALGO_BYTES_PER_FLIP / 1e9
    return dict(bound="hbm", achieved=a, peak=HBM_PEAK_GBS, unit="GB/s", frac=a / HBM_PEAK_GBS,
                note="CE flip, occupancy LDS-resident: nominal HBM fraction (56 B/flip)")


def _engine_run(Engine, wl, device, clock, launches, mc, equil=0, rex=None, transient_mc=None, device_decide=False):
    """transient (first `launches` launches of `transient_mc` steps after one warm-up launch of the
    same length) and, after `equil` more untimed steps per walker, steady-state figures (launches of
    `mc` steps) of one workload on this rank's walkers; with `rex` every launch is followed by one
    exchange attempt of the global ladder."""
    from smol_amd import parallel

    eng = Engine(wl.tables, wl.make_config(device))
    eng.set_state(wl.occupancy, wl.seeds, wl.temperature)

    def launch(n=1, steps=None):
        steps = mc if steps is None else steps
        if rex is None:
            for _ in range(n):
                eng.run(steps)
        else:
            parallel.run_replica_exchange(eng, rex, n, steps, device_decide=device_decide)

    def measure(mc=mc):
        launch(1, mc)
        eng.sync()
        s0 = eng.get_state(occupancy=False)
        kms = []

        def body():
            for _ in range(launches):
                launch(1, mc)
                kms.append(eng.last_kernel_ms())
            eng.sync()

        dt = clock.time(body)
        s1 = eng.get_state(occupancy=False)
        acc, walkers, kms_sum = clock.sum([float((s1["n_accepted"] - s0["n_accepted"]).sum()),
                                           float(wl.n_walkers), float(np.mean(kms))])
        steps = walkers * mc * launches
        return dict(kernel_ms=kms_sum / clock.world, wall_s=dt, mc_steps_per_s=steps / dt,
                    mc_steps_per_s_kernel_only=walkers * mc / (kms_sum / clock.world * 1e-3),
                    flips_per_s=wl.flips_per_step * walkers * mc / (kms_sum / clock.world * 1e-3),
                    acceptance=acc / steps, timed_steps_per_replica=mc * launches)

    keep = launches
    if transient_mc:  # the transient figure of round 2: launches 2-4 from the random start
        launches = 3
    first = measure(transient_mc or mc)
    launches = keep
    steady = None
    if equil:
        done = 4 * transient_mc if transient_mc else (launches + 1) * mc
        while done < equil:
            launch(10)
            done += 10 * mc
        eng.sync()
        steady = measure()
        steady["equilibration_steps_per_replica"] = done
    info = eng.kernel_info()
    eng.close()
    return info, first, steady


def time_other_configs(device, rank=0, world=1, red_dev="cuda"):
    """The entries measured before a failing one survive it (round 5's first run lost all of them to one walker of
    config 11 that started outside its Wang-Landau window)."""
    out = []
    try:
        _time_other_configs(device, rank, world, red_dev, out)
    except Exception as exc:
        out.append({"error": f"rank {rank}: {type(exc).__name__}: {exc}"[:300]})
        if rank != 0:
            sys.stderr.write(out[-1]["error"] + "\n")
    return out


def _time_other_configs(device, rank, world, red_dev, out):
    """Configs 1, 3, 4, 5 of BASELINE.json on one rank; configs 4 and 5 on N ranks.  Kernel time
    from HIP events on the launch stream, wall time barrier-bracketed with the MAX over ranks,
    and the roofline that actually bounds each.  Configs 3 and 5 start from random
    occupancies: their figures are reported for the first launches (transient) AND after
    EQUIL_STEPS more steps per walker (steady state), each with its acceptance."""
    from smol_amd import capi, parallel, workloads
    from smol_amd.engine import Engine

    clock = _Clock(world, red_dev)

    def record(wl, info, first, steady, roof, replicas, launches, **extra):
        main = steady or first
        rec = dict(config=wl.name, kernel=info, n_gpus=world, replicas=replicas,
                   mc_steps_per_launch=main["timed_steps_per_replica"] // launches,
                   kernel_ms=main["kernel_ms"], mc_steps_per_s=main["mc_steps_per_s"],
                   mc_steps_per_s_kernel_only=main["mc_steps_per_s_kernel_only"],
                   flips_per_s=main["flips_per_s"], acceptance=main["acceptance"],
                   timed_steps_per_replica=main["timed_steps_per_replica"], **extra)
        if steady:
            rec["state"] = "steady"
            rec["equilibration_steps_per_replica"] = steady["equilibration_steps_per_replica"]
            rec["transient"] = dict(kernel_ms=first["kernel_ms"], mc_steps_per_s=first["mc_steps_per_s"],
                                    flips_per_s=first["flips_per_s"], acceptance=first["acceptance"])
        rec["roofline"] = roof(rec)
        out.append(rec)

    if world == 1:
        wl = workloads.config1()
        info, first, _ = _engine_run(Engine, wl, device, clock, 5, 100_000)
        record(wl, info, first, None, hbm_ce, wl.n_walkers, launches=5)

        wl3 = workloads.config3()

        ewald_note = ("potential-field formulation: a proposal reads O(1) LDS words, an ACCEPTED flip sweeps the walker's "
                      "field (n_act gathers from the translation-compressed tables, 95 KB, L2 resident, + n_act LDS "
                      "read-modify-writes); bound by the issue rate of one wave's LDS / address instructions, not by "
                      "HBM (measured fetch 22 MB per launch); the dense two-row formulation of SURVEY 8d (58752 B/flip) is the "
                      "SMOLMC_DENSE_EWALD path: pmc entry config3_dense_ewald")

        # Config 3 at its frozen temperature (round 6: 40000 K, steady-state acceptance 0.38 -- workloads.CONFIG3_T):
        # the transient window (launches 2-11 of 2000 steps from the random start) and the equilibrated figure.
        info, first, steady = _engine_run(Engine, wl3, device, clock, 10, 20_000, equil=EQUIL_STEPS[3],
                                          transient_mc=2000)
        record(wl3, info, first, steady, issue_roof("config3", ewald_note), wl3.n_walkers, launches=10)
        # ... and the point rounds 2-5 quoted (3000 K): there the unconstrained flips run to a pure composition and the
        # steady figure is the cost of REJECTED proposals
        wl3r = workloads.config3(temperature=workloads.CONFIG3_T_REJECT)
        wl3r.name = wl3r.name.replace("config3:", "config3_reject_path:")
        info, first, steady = _engine_run(Engine, wl3r, device, clock, 10, 20_000, equil=EQUIL_STEPS[3],
                                          transient_mc=2000)
        record(wl3r, info, first, steady, issue_roof("config3_reject_path", ewald_note), wl3r.n_walkers, launches=10)
        out[-1]["state"] = "reject-path only (3000 K: the chain runs to a pure composition, acceptance ~ 0)"
        wl9 = workloads.config9()
        info, first, steady = _engine_run(Engine, wl9, device, clock, 10, 20_000, equil=EQUIL_STEPS[3],
                                          transient_mc=2000)
        record(wl9, info, first, steady, issue_roof("config9", ewald_note), wl9.n_walkers, launches=10)

    if world == 1:
        # H1, the function north_star names first (evaluator.pyx:211-265, ClusterExpansionProcessor): the headline
        # model and config 3 with the CORRELATION-function trace.  K = 1 per orbit (binary): the lean kernel with
        # tables made from the correlation tensors; K = 3 / 4 / 6 (ternary): the KF instantiations (lean_corr_n*.hip).
        wl12 = workloads.config12()
        info, first, _ = _engine_run(Engine, wl12, device, clock, 5, 100_000)
        record(wl12, info, first, None,
               issue_roof("config12", "H1 (correlation trace), K = 1 function per orbit: the headline kernel on tables made "
                                      "from the correlation tensors (DESIGN.md 4.1a)"), wl12.n_walkers, launches=5)
        wl13 = workloads.config13()
        info, first, steady = _engine_run(Engine, wl13, device, clock, 10, 20_000, equil=EQUIL_STEPS[3], transient_mc=2000)
        record(wl13, info, first, steady,
               issue_roof("config13", "H1 (correlation trace), K = 3 / 4 / 6 functions per orbit: decision from one folded "
                                      "table per slot, the K function tables read on accepted steps only (DESIGN.md 4.1a)"),
               wl13.n_walkers, launches=10)
        # (config 13 follows config 3 to its frozen temperature: a genuine steady state now)
        # the reference's own model (LiNiO2, two active sublattices, Ewald) under Wang-Landau: mc_lean_multi_kernel<..., WLK>
        try:
            # (the window must hold EVERY walker's random start: the Ewald term gives the starting enthalpies a long
            # upper tail -- round 5's first bench run lost this entry to walker 1225 of 4096 -- so the upper edge
            # sits 30 eV above the highest of them)
            wl11 = workloads.config11(count=4096)
            probe = Engine(wl11.tables, capi.make_config(1, device=device))
            hs = probe.eval_full(wl11.occupancy) @ probe.natural_parameters
            probe.close()
            h11 = float(hs.max()) + 30.0 - 95.63
            assert float(hs.min()) > h11 - 160.37, "config 11: starting enthalpies wider than the window"
            for count in (1024, 4096):
                wl11 = workloads.config11(count=count, h0=h11)
                info, first, _ = _engine_run(Engine, wl11, device, clock, 5, 2000)
                record(wl11, info, first, None,
                       issue_roof("config11", "Wang-Landau on two active sublattices + Ewald field in LDS: three steps of four "
                                              "are accepted and sweep the field (DESIGN.md 4.1c)"), wl11.n_walkers, launches=5)
        except Exception as e:  # (the slimmed model file travels with tests/golden; an entry outside BASELINE.json must not cost the others)
            out.append(dict(config="config11: LiNiO2 under Wang-Landau", error=f"{type(e).__name__}: {e}"[:300]))

    if world == 1:
        # the backstop kernel on the headline workload (round-5 review item 6) and the lazy-feature path on config 13
        for env, build, key, note in (
            ("SMOLMC_FORCE_UNIVERSAL", lambda: workloads.config2(mc=2000), "config2_universal",
             "config 2 forced onto mc_univ_kernel (the backstop: TableFlip x Wang-Landau / biases, any cluster and class count, "
             "HBM-resident occupancies); VALU-bound at four waves per SIMD (DESIGN.md 4.8)"),
            ("SMOLMC_LAZY_FEATURES_ONLY", lambda: workloads.config13(mc=3456), "config13_lazy",
             "config 13 with lazy cluster features (DESIGN.md 4.9): decisions from the folded tensors on the plain lean kernel, "
             "correlation functions evaluated where the trace is read; kernel-only figure"),
        ):
            os.environ[env] = "1"
            try:
                wlx = build()
                info, first, _ = _engine_run(Engine, wlx, device, clock, 5, wlx.mc_per_launch)
            finally:
                del os.environ[env]
            wlx.name = key + ": " + wlx.name
            record(wlx, info, first, None, issue_roof(key, note), wlx.n_walkers, launches=5)

    if world == 1:
        # config 3 on the LITERAL formulation of ewald.pyx:38-58 (two rows of the 382 MB matrix gathered per
        # proposal; what a matrix that does not factorise takes): the one HBM-bound kernel of the engine
        os.environ["SMOLMC_DENSE_EWALD"] = "1"
        try:
            wl3d = workloads.config3(mc=500)
            info, first, _ = _engine_run(Engine, wl3d, device, clock, 5, 500)
        finally:
            del os.environ["SMOLMC_DENSE_EWALD"]
        A_EW = 2.0 * wl3d.sc.num_sites * 8.0 + wl3d.sc.num_sites  # 58 752 B per flip at N = 3456

        def dense_roof(rec):
            from smol_amd import codeobj

            pmc = pmc_constants().get("config3_dense_ewald", {})
            a = rec["flips_per_s"] * A_EW / 1e9
            return dict(bound="hbm", achieved=a, peak=HBM_PEAK_GBS, unit="GB/s", frac=a / HBM_PEAK_GBS,
                        algorithmic_bytes_per_flip=A_EW, traffic=pmc.get("hbm_bytes_per_launch"),
                        traffic_unit="bytes per launch of 2048 x 500 flips (rocprofv3 FETCH_SIZE + WRITE_SIZE passes)",
                        pmc_source=pmc.get("source", "none"),
                        pmc_stale=codeobj.isa_stale(pmc),
                        note="dense Ewald rows (SMOLMC_DENSE_EWALD): streaming gather, eight sites per lane in flight")

        wl3d.name = "config3 with the dense Ewald rows (ewald.pyx:38-58 literally)"
        record(wl3d, info, first, None, dense_roof, wl3d.n_walkers, launches=5)

    # config 4: 1024 independent Wang-Landau walkers per rank; the window is centred on the
    # starting enthalpy, evaluated on the engine
    wl4 = workloads.config4(first=rank * 1024)
    probe = Engine(wl4.tables, capi.make_config(1, device=device))
    h0 = float(probe.natural_parameters @ probe.eval_full(workloads.config4(count=1).occupancy[:1])[0])
    probe.close()
    wl4 = workloads.config4(first=rank * 1024, h0=h0)
    info, first, _ = _engine_run(Engine, wl4, device, clock, 5, 50_000)
    record(wl4, info, first, None,
           issue_roof("config4", "one wave per SIMD at 1024 walkers: the step is its chain of dependent instructions "
                                 "(DESIGN.md section 5); occupancy and Wang-Landau state LDS-resident"),
           wl4.n_walkers * world, launches=5, sharding="independent walkers, no collective")

    # config 5: TableFlip + ONE replica-exchange ladder over all ranks' walkers; every launch is
    # followed by an exchange attempt (one rank: decisions from a direct read-back; N ranks: RCCL
    # all-gather of the enthalpies, identical decisions on every rank, temperatures move)
    per = 2048
    wl5 = workloads.config5(first=rank * per, count=per, total=per * world)
    rex = parallel.ReplicaExchange(wl5.extras["ladder"], per, rank, world, seed=11)
    # N ranks on RCCL: the swap decisions are taken by a kernel on the all-gathered device tensor (smolmc_exchange_dev,
    # round 6) -- the host NumPy decisions were the serial section of the loop; SMOLMC_REX_DEVICE_DECIDE=0 switches back
    dev_decide = world > 1 and red_dev == "cuda" and os.environ.get("SMOLMC_REX_DEVICE_DECIDE", "1") != "0"
    info, first, steady = _engine_run(Engine, wl5, device, clock, 30, wl5.mc_per_launch,
                                      equil=EQUIL_STEPS[5], rex=rex, device_decide=dev_decide)
    record(wl5, info, first, steady,
           issue_roof("config5", "TableFlip step (three flips + proposal + a-priori factor) at two waves per SIMD: "
                                 "instruction-issue / latency bound (DESIGN.md section 5); mc_steps_per_s is wall time "
                                 "including the exchange step, mc_steps_per_s_kernel_only from the HIP events"),
           per * world, launches=30, exchange_every_steps=wl5.mc_per_launch,
           exchange_acceptance_mean=float(rex.acceptance.mean()),
           exchange_latency_ms_per_sweep=float(clock.max([rex.exchange_seconds / max(rex.exchange_timed, 1) * 1e3])[0]),
           exchange_path="collective (all-gather over %d ranks)" % world if world > 1 else "single rank (direct read-back)",
           exchange_decisions="device kernel (smolmc_exchange_dev)" if dev_decide else "host (NumPy)")
    if world == 1:
        # BASELINE's own ladder for config 5 (400-2000 K): equilibrated it accepts 0.1 % of its steps
        # (why the default ladder above is hotter), so only the window rounds 1-2 quoted is timed
        wl5c = workloads.config5(first=0, count=per, total=per, t_lo=400.0, t_hi=2000.0)
        rexc = parallel.ReplicaExchange(wl5c.extras["ladder"], per, 0, 1, seed=11)
        info, first, _ = _engine_run(Engine, wl5c, device, clock, 3, wl5c.mc_per_launch, rex=rexc)
        record(wl5c, info, first, None,
               issue_roof("config5", "as config 5 above (PMC passes of the hot ladder); launches 2-4 from the random start"),
               per, launches=3, state="transient", exchange_every_steps=wl5c.mc_per_launch,
               exchange_acceptance_mean=float(rexc.acceptance.mean()))
    return out


# --------------------------------------------------------------------------------------
class _DryEngine:
    """Stand-in for smol_amd.engine.Engine in --dry-run (no GPU): made-up enthalpies, so that the
    N-rank control flow of configs 4 / 5 -- sharding, ladder, host-staged all-gather, identical
    decisions, temperature moves, barrier-bracketed timing, reductions -- runs over gloo."""

    def __init__(self, n, seed):
        self.n, self.rng, self.T = n, np.random.default_rng(seed), np.full(n, 1000.0)
        self.steps = 0

    def set_temperature(self, t):
        self.T = np.broadcast_to(np.asarray(t, dtype=np.float64), (self.n,)).copy()

    def run(self, nsteps, sync=False):
        self.steps += int(nsteps)

    def sync(self):
        pass

    def get_enthalpy(self):
        return self.rng.normal(0.0, 0.5, self.n) - 3000.0 / self.T


def dry_run(args, rank, world):
    """The multi-rank control flow without a GPU (gloo): rendezvous, barrier, timed loop of
    no-ops, MAX-reduce of the time, SUM-reduce of the statistics; then the N-rank forms of
    configs 4 (independent shards) and 5 (global ladder, exchange inside the timed region)."""
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
    ones = torch.ones(1, dtype=torch.float64)  # the same rank census the GPU path takes
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    if int(round(float(ones.item()))) != args.gpus:
        raise SystemExit(4)
    # configs 4 / 5 as N-rank workloads
    clock = _Clock(world, "cpu", gpu=False)
    per4, per5, sweeps = 1024, 2048, 6
    e4 = _DryEngine(per4, 100 + rank)
    t4 = clock.time(lambda: e4.run(50_000))
    w4 = clock.sum([per4])[0]
    ladder = parallel.geometric_ladder(400.0, 2000.0, per5 * world)
    rex = parallel.ReplicaExchange(ladder, per5, rank, world, seed=11)
    e5 = _DryEngine(per5, 200 + rank)
    t5 = clock.time(lambda: parallel.run_replica_exchange(e5, rex, sweeps, 3456))
    check = float((rex.rung_of * np.arange(1, rex.n + 1)).sum())  # order-sensitive checksum of the ladder
    check_mean = clock.sum([check])[0] / world                   # == check when every rank decided alike
    if rank == 0:
        print(json.dumps({
            "metric": METRIC, "value": None, "unit": "attempted flips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
            "scaling": args.scaling, "dry_run": True, "walkers_total": int(stats[3].item()),
            "rccl_ranks": int(round(float(ones.item()))), "collective_backend": "gloo (dry run, no GPU)" if world > 1 else "none (single rank)",
            "walkers_rank0": [first, count],
            "other_configs": [
                {"config": "config4 (dry run)", "n_gpus": world, "replicas": int(w4), "wall_s": t4,
                 "sharding": "independent walkers, no collective"},
                {"config": "config5 (dry run)", "n_gpus": world, "replicas": rex.n, "wall_s": t5,
                 "exchanges": rex.calls, "exchange_attempts": int(rex.attempted.sum()),
                 "exchange_accepted": int(rex.accepted.sum()),
                 "rungs_are_a_permutation": bool(sorted(rex.rung_of) == list(range(rex.n))),
                 "ranks_agree": bool(abs(check_mean - check) < 0.5),
                 "exchange_path": "collective (all-gather over %d ranks, host-staged: gloo)" % world
                 if world > 1 else "single rank"},
            ],
        }))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--replicas", type=int, default=N_REPLICAS,
                    help="walkers per GPU (weak) or in total (strong)")
    ap.add_argument("--mc-per-step", type=int, default=MC_PER_STEP)
    ap.add_argument("--features", choices=("interactions", "correlations"), default="interactions")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="skip the 4096-walkers-in-total measurement on N > 1 ranks")
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

    # how many ranks actually take part in the collectives: an all-reduce of ones over the backend the
    # job runs on; the job refuses to report a number when that is not --gpus
    red_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        ones = torch.ones(1, dtype=torch.float64, device=red_dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        rccl_ranks = int(round(float(ones.item())))
        if backend == "nccl":
            try:  # (the version is decoration: it must never cost the run)
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                ver = "?"
            collective_backend = "nccl (RCCL %s over xGMI)" % ver
        else:
            collective_backend = "gloo (host-staged; --oversubscribe)"
    else:
        rccl_ranks, collective_backend = 1, "none (single rank: no process group, no collective on the path)"
    if rccl_ranks != args.gpus:
        sys.stderr.write(f"bench.py: {rccl_ranks} ranks took part in the all-reduce, --gpus {args.gpus}\n")
        raise SystemExit(4)

    # what every rank runs on (the first N > 1 line must describe itself): device name, architecture, CUs and
    # XCDs (the KFD topology's num_xcc of the GPU nodes -- identical GPUs on one node; else derived from the CUs)
    prop = torch.cuda.get_device_properties(device)
    my_dev = dict(rank=rank, device=device, name=prop.name, arch=getattr(prop, "gcnArchName", "?"),
                  compute_units=int(prop.multi_processor_count), xcds=_xcd_count(int(prop.multi_processor_count)),
                  hbm_gib=round(prop.total_memory / 2**30, 1))
    if world > 1:
        devs = [None] * world
        dist.all_gather_object(devs, my_dev)
    else:
        devs = [my_dev]

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
    kms_lo = torch.tensor([float(np.mean(kernel_ms))], dtype=torch.float64, device=red_dev)
    kms_hi = kms_lo.clone()
    if world > 1:
        dist.all_reduce(kms_lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(kms_hi, op=dist.ReduceOp.MAX)
    # north_star's "at 4096 replicas" next to the weak figure: 4096 walkers IN TOTAL, 4096 / N per GPU,
    # the same number of launches (one rank: the headline measurement is that figure already)
    strong = None
    if world > 1 and args.scaling == "weak" and not args.no_strong:
        eng.close()
        f2, R2 = parallel.shard(N_REPLICAS, rank, world)
        wl2 = workloads.config2(f2, R2, feature_mode=mode, mc=args.mc_per_step)
        eng = Engine(wl2.tables, wl2.make_config(device))
        eng.set_state(wl2.occupancy, wl2.seeds, wl2.temperature)
        for _ in range(min(args.warmup, 2)):
            eng.run(args.mc_per_step)
        eng.sync()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            eng.run(args.mc_per_step)
        eng.sync()
        barrier()
        dt2 = time.perf_counter() - t1
        tt = torch.tensor([dt2], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt2 = float(tt.item())
        strong = {"replicas_total": N_REPLICAS, "replicas_per_gpu": R2, "ms_per_step": dt2 / args.steps * 1e3,
                  "value": 2.0 * args.steps * args.mc_per_step * N_REPLICAS / dt2, "unit": "attempted flips/s",
                  "scaling": "strong"}

    if rank == 0:
        n_walk = stats[3]
        total_steps = float(args.steps) * args.mc_per_step * n_walk
        flips = 2.0 * total_steps
        value = flips / dt
        k_ms = float(np.mean(kernel_ms))
        flips_per_launch = 2.0 * args.mc_per_step * R
        achieved = flips_per_launch * ALGO_BYTES_PER_FLIP / (k_ms * 1e-3) / 1e9
        pmc = pmc_constants().get(f"{R}x{args.mc_per_step}", {}) if args.features == "interactions" else {}
        from smol_amd import codeobj

        # the counters were collected on the kernel whose machine-code digest the entry carries (codeobj.py)
        pmc_stale = codeobj.isa_stale(pmc)
        probe = measure_hbm_peak()
        measured_peak = max(probe.values()) if probe else HBM_MEASURED_FALLBACK_GBS
        roof = {
            "bound": "hbm",
            "kernel": eng.kernel_info(),
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "measured_peak": measured_peak,
            "measured_peak_source": ("live read / triad probe over 2 GiB float64 arrays: " + json.dumps(probe))
            if probe else "tools/hbm_triad.py figure of this pool (live probe failed)",
            "frac_of_measured_peak": achieved / measured_peak,
            "pmc_stale": pmc_stale,
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
            "devices": devs,
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
            "kernel_ms_min_over_ranks": float(kms_lo.item()),
            "kernel_ms_max_over_ranks": float(kms_hi.item()),
            "rccl_ranks": rccl_ranks,
            "collective_backend": collective_backend,
            "strong_scaling_4096_total": strong if strong is not None else (
                {"replicas_total": int(n_walk), "value": value, "unit": "attempted flips/s", "scaling": "strong",
                 "note": "one rank: identical to the headline measurement"} if world == 1 and R == N_REPLICAS else None),
            "roofline": roof,
        }
        assert int(n_walk) == total_walkers
    eng.close()
    # Every rank takes part in the N-rank forms of configs 4 / 5 (rank 0 reports).  The headline
    # line must survive whatever happens there: a rank that fails alone leaves the others inside a
    # collective, so on N ranks a watchdog ends every rank after OTHER_CONFIGS_TIMEOUT_S, rank 0
    # printing the line with what it has.
    printed = []

    def emit():
        if rank == 0 and not printed:
            printed.append(True)
            print(json.dumps(out), flush=True)

    def bail():
        if rank == 0:
            out.setdefault("other_configs", [{"error": "the N-rank configs 4/5 did not finish within "
                                                       f"{OTHER_CONFIGS_TIMEOUT_S} s"}])
        emit()
        os._exit(0)

    watchdog = None
    if world > 1:
        import threading

        watchdog = threading.Timer(OTHER_CONFIGS_TIMEOUT_S, bail)
        watchdog.daemon = True
        watchdog.start()
    other = None
    if not args.no_other_configs:
        try:
            other = time_other_configs(device, rank, world, red_dev)
        except Exception as exc:  # the headline measurement must not be lost with it
            other = [{"error": f"rank {rank}: {type(exc).__name__}: {exc}"[:300]}]
            if rank != 0:
                sys.stderr.write(other[0]["error"] + "\n")
    if rank == 0:
        if other is not None:
            out["other_configs"] = other
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
                out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            except Exception as exc:  # the GPU measurement must not be lost with it
                out["cpu_baseline"] = {"value": None, "unit": "attempted flips/s", "cores": usable_cores(),
                                       "kind": "port", "sample": f"failed: {exc}"[:300]}
        emit()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if watchdog is not None:
        watchdog.cancel()


if __name__ == "__main__":
    main()
