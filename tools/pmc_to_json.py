#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes of `bench.py` (rocpd databases under gpurun_out/) into the
per-launch / per-step figures bench.py attaches to its roofline object
(profiles/pmc_constants.json).  Counters cannot be read from inside an un-profiled run, so the
bench line cites the passes of the same build instead.

    python tools/pmc_to_json.py --kernel mc_lean_kernel --replicas 4096 --mc 10000 \
        --source profiles/r02_headline_pmc.txt gpurun_out/pmc_r02_* > profiles/pmc_constants.json

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* are quad-cycles summed over
waves (x4 = SIMD cycles); FETCH_SIZE / WRITE_SIZE are KiB.  FETCH_SIZE is used undoubled: this
kernel's occupancy stream is 4 B/lane and the raw counter reproduces its known byte count
(profiles/README.md)."""
import argparse
import glob
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smol_amd import codeobj  # noqa: E402
from smol_amd.engine import source_digest  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="+")
    ap.add_argument("--kernel", default="mc_lean_kernel")
    ap.add_argument("--replicas", type=int, default=4096)
    ap.add_argument("--mc", type=int, default=10000)
    ap.add_argument("--source", default="profiles/")
    ap.add_argument("--merge", default=None, help="existing json to update")
    ap.add_argument("--key", default=None, help='entry name (default "<replicas>x<mc>"; other configurations: "config3" ...)')
    ap.add_argument("--waves-per-simd", type=float, default=0.0, help="resident waves per SIMD (default replicas / 1024)")
    a = ap.parse_args()
    by_name = {}
    for p in a.paths:
        for db in (sorted(glob.glob(os.path.join(p, "**", "*.db"), recursive=True)) if os.path.isdir(p) else [p]):
            con = sqlite3.connect(db)
            try:
                rows = con.execute(
                    "select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                    "group by kernel_name, counter_name").fetchall()
            except sqlite3.Error:
                continue
            for name, ctr, val, n in rows:
                if a.kernel in name:
                    by_name.setdefault(name, {})[ctr] = (float(val), int(n))
    if not by_name:
        raise SystemExit("no counters for kernel " + a.kernel)
    # several instantiations may match the fragment (a replay or set-up kernel of the same family): the
    # entry is the one the passes spent their waves in
    full_name = max(by_name, key=lambda k: max(v for v, _ in by_name[k].values()))
    tot = by_name[full_name]
    waves_per_simd = a.waves_per_simd if a.waves_per_simd > 0 else max(1, a.replicas // 1024)

    def per_step(c):
        v, n = tot[c]
        return v / (n * a.mc * a.replicas)

    def per_launch(c):
        v, n = tot[c]
        return v / n

    # the digest of the kernel sources these counters were collected on: bench.py compares it with
    # the tree it runs from and prints "pmc_stale": true when they differ
    # ... and the kernel's full name with the digest of its machine code in the library of this tree
    # (smol_amd/codeobj.py): bench.py's "pmc_stale" compares THAT, so an entry goes stale when its kernel's
    # instruction stream moves and not when a comment does
    rec = {"source": a.source, "kernel": a.kernel, "waves_per_simd": waves_per_simd,
           "csrc_sha256": source_digest()}
    hit = codeobj.find_kernel(full_name)
    if hit is not None:
        rec["kernel_symbol"], rec["isa_sha256"] = hit
    for c, k in (("SQ_INSTS_VALU", "valu_per_step"), ("SQ_INSTS_SALU", "salu_per_step"),
                 ("SQ_INSTS_LDS", "lds_per_step"), ("SQ_INSTS_VMEM_RD", "vmem_per_step")):
        if c in tot:
            rec[k] = round(per_step(c), 2)
    if "SQ_WAVE_CYCLES" in tot:
        rec["wave_cycles_per_step"] = round(4.0 * per_step("SQ_WAVE_CYCLES"), 1)
    if "SQ_ACTIVE_INST_VALU" in tot and "SQ_WAVE_CYCLES" in tot:
        busy = 4.0 * per_step("SQ_ACTIVE_INST_VALU") * waves_per_simd
        rec["valu_busy_cycles_per_step_per_simd"] = round(busy, 1)
        rec["valu_issue_frac"] = round(busy / (4.0 * per_step("SQ_WAVE_CYCLES")), 4)
    if "SQ_ACTIVE_INST_LDS" in tot and "SQ_WAVE_CYCLES" in tot:
        # LDS-instruction issue cycles of the waves of a SIMD over the cycles of a step (the share of the
        # step during which the SIMD has an LDS instruction in its pipe)
        rec["lds_issue_frac"] = round(per_step("SQ_ACTIVE_INST_LDS") * waves_per_simd / per_step("SQ_WAVE_CYCLES"), 4)
    if "SQ_LDS_BANK_CONFLICT" in tot:
        rec["lds_bank_conflict_per_step"] = round(per_step("SQ_LDS_BANK_CONFLICT"), 1)
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        rec["fetch_bytes_per_launch"] = per_launch("FETCH_SIZE") * 1024.0
        rec["write_bytes_per_launch"] = per_launch("WRITE_SIZE") * 1024.0
        rec["hbm_bytes_per_launch"] = rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"]
    out = json.load(open(a.merge)) if a.merge and os.path.exists(a.merge) else {}
    rec["replicas"], rec["mc_steps_per_launch"] = a.replicas, a.mc
    out[a.key or f"{a.replicas}x{a.mc}"] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
