#!/usr/bin/env python3
"""Print (or summarise) the hot loop of a kernel from hipcc -S output: the instructions between the
first `s_setprio 1` and the last back-edge after `s_setprio 0`.  Usage:
    tools/isa_loop.py file.s <mangled-kernel-name-substring> [--dump]"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    s = open(path).read()
    m = re.search(r"^(_Z\w*%s\w*):" % re.escape(key), s, re.M)
    i = m.start()
    j = s.index(".Lfunc_end", i)
    lines = [l.split(";")[0].rstrip() for l in s[i:j].splitlines()]
    lines = [l for l in lines if l.strip() and not l.strip().startswith(".") or l.strip().endswith(":")]
    a = next(k for k, l in enumerate(lines) if "s_setprio 1" in l)
    b = max(k for k, l in enumerate(lines) if "s_setprio 0" in l)
    # extend to the end of the block after the last s_setprio 0 that branches back
    e = b
    while e < len(lines) - 1 and not re.match(r"\s*s_cbranch_scc[01]|\s*s_branch", lines[e]):
        e += 1
    region = lines[a:e + 1]
    c = collections.Counter()
    for l in region:
        t = l.strip()
        if t.endswith(":"):
            continue
        c[t.split()[0]] += 1
    tot = sum(c.values())
    cls = collections.Counter()
    for op, n in c.items():
        k = ("VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_nop", "s_setprio"))
             else "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("buffer_", "global_", "flat_")) else op.split("_")[1] if op.startswith("s_") else op)
        cls[k] += n
    print(m.group(1), "static instructions in the step region:", tot, dict(cls))
    print("spill traffic: v_readlane", c["v_readlane_b32"], "v_writelane", c["v_writelane_b32"])
    if dump:
        print("\n".join(region))


if __name__ == "__main__":
    main()
