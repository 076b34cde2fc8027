#!/usr/bin/env python3
"""Probe OpenMP settings for the CPU-oracle baseline on the GPU box's host cores."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, numpy as np
sys.path.insert(0, %r)
import bench
from oracle import oracle as orc
from smol_amd import capi
model, sc, tab = bench.build_workload()
T = int(os.environ["OMP_NUM_THREADS"]); R = T * 4
mc = orc.OracleMC(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP))
mc.set_state(bench.initial_occupancies(sc, 0, R), np.arange(R, dtype=np.uint64) + np.uint64(12345), 2500.0)
mc.run(2000); t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 4.0:
    mc.run(5000); n += 5000
dt = time.perf_counter() - t0
print(T, os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_WAIT_POLICY"), "flips/s %%.3e" %% (2.0 * R * n / dt), "per-thread steps/s %%.0f" %% (R * n / dt / T))
''' % ROOT
for threads in (1, 8, 16, 32, 64):
    for bind, wait in ((None, None), ("close", "passive"), ("spread", "active")):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads))
        if bind: env.update(OMP_PROC_BIND=bind, OMP_PLACES="cores", OMP_WAIT_POLICY=wait)
        subprocess.run([sys.executable, "-c", code], env=env)
