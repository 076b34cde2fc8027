#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel durations (the --stats view) and,
when present, PMC counters per kernel.  Usage: rocpd_summary.py <dir-or-db> [...]"""
import glob
import os
import sqlite3
import sys


def dbs(path):
    if os.path.isdir(path):
        return sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    return [path]


def short(name, n=200):
    """torch's elementwise kernels carry kilobytes of template arguments: keep the head"""
    return name if len(name) <= n else name[:n] + "...>"


def main():
    for arg in sys.argv[1:]:
        for db in dbs(arg):
            con = sqlite3.connect(db)
            print(f"== {db}")
            try:
                rows = con.execute(
                    "select name, total_calls, total_duration, average, percentage from top_kernels"
                ).fetchall()
                print("KERNEL_STATS name | calls | total_ms | avg_ms | pct")
                for r in rows:
                    print(f"  {short(r[0])} | {r[1]} | {r[2]/1e3:.3f} | {r[3]/1e3:.3f} | {r[4]:.2f}")
            except sqlite3.Error as e:
                print("  (no kernel stats)", e)
            try:
                rows = con.execute(
                    "select kernel_name, counter_name, sum(value), count(*), max(vgpr_count), "
                    "max(sgpr_count), max(lds_block_size) from counters_collection "
                    "group by kernel_name, counter_name"
                ).fetchall()
                if rows:
                    print("PMC kernel | counter | sum | dispatches | vgpr | sgpr | lds")
                    for r in rows:
                        if r[0].startswith("__amd"):
                            continue
                        print(f"  {short(r[0])} | {r[1]} | {r[2]:.6g} | {r[3]} | {r[4]} | {r[5]} | {r[6]}")
            except sqlite3.Error:
                pass


if __name__ == "__main__":
    main()
