#!/usr/bin/env python3
"""Achievable HBM bandwidth of this box ("confirm the 8 TB/s nominal peak with a
stream-triad micro-benchmark and quote the measured figure beside it").  torch is used only as
the allocator / launcher of three elementwise kernels on large float64 arrays (4 GiB each, far
beyond the 256 MiB Infinity Cache)."""
import json

import torch


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    n = 1 << 29  # 4 GiB per float64 array
    a = torch.empty(n, dtype=torch.float64, device="cuda")
    b = torch.rand(n, dtype=torch.float64, device="cuda")
    c = torch.rand(n, dtype=torch.float64, device="cuda")
    out = {}
    t = timed(lambda: a.copy_(b))
    out["copy_GBs"] = 2 * n * 8 / t / 1e9
    t = timed(lambda: torch.add(b, c, alpha=3.0, out=a))
    out["triad_GBs"] = 3 * n * 8 / t / 1e9
    t = timed(lambda: a.fill_(1.0))
    out["fill_GBs"] = n * 8 / t / 1e9
    t = timed(lambda: b.sum())
    out["read_GBs"] = n * 8 / t / 1e9
    out["nominal_peak_GBs"] = 8000.0
    print(json.dumps(out))


if __name__ == "__main__":
    main()
