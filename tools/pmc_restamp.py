#!/usr/bin/env python3
"""Give the entries of profiles/pmc_constants.json the per-kernel stamp (kernel_symbol, isa_sha256;
smol_amd/codeobj.py) without new counter passes -- for entries collected before that stamp existed.

An entry is stamped only when it can be tied to machine code:
  * the kernel's full name is taken from the rocprofv3 summary the entry cites (`source`; for the
    per-configuration file the section of its key), the SQ_WAVE_CYCLES row of the family it names;
  * the digest is taken from a library BUILT FROM THE TREE THE COUNTERS WERE COLLECTED ON: --tree is a
    checkout whose engine.source_digest equals the entry's csrc_sha256 (refused otherwise), --lib the
    libsmolmc_hip.so made from it.

    git archive <commit> smol_amd/csrc include | tar -x -C /tmp/t && make -C /tmp/t/smol_amd/csrc -j
    python tools/pmc_restamp.py --tree /tmp/t --lib /tmp/t/smol_amd/libsmolmc_hip.so

Whether the entry is stale for the CURRENT library is then bench.py's comparison, as for any entry."""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smol_amd import codeobj  # noqa: E402
from smol_amd.engine import source_digest  # noqa: E402

SECTION = {"config3_dense_ewald": "d3", "config2_universal": "u2", "config13_lazy": "l13"}


def full_name(entry, key):
    """The kernel of `entry` as the summary it cites spells it."""
    path = os.path.join(ROOT, entry["source"])
    lines = open(path).read().split("\n")
    if "configs" in os.path.basename(path):
        tag = SECTION.get(key, key.replace("config", ""))
        start = next(i for i, l in enumerate(lines) if l.startswith(f"######## config {tag}:"))
        end = next((i for i in range(start + 1, len(lines)) if lines[i].startswith("########")), len(lines))
        lines = lines[start:end]
    best = None
    for l in lines:
        m = re.match(r"\s*(.*\S)\s*\|\s*SQ_WAVE_CYCLES\s*\|\s*([0-9.e+]+)\s*\|", l)
        if m and entry["kernel"] in m.group(1) and (best is None or float(m.group(2)) > best[1]):
            best = (m.group(1), float(m.group(2)))
    return best[0] if best else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tree", required=True, help="checkout of the tree the counters were collected on")
    ap.add_argument("--lib", required=True, help="libsmolmc_hip.so built from --tree")
    ap.add_argument("--file", default=os.path.join(ROOT, "profiles", "pmc_constants.json"))
    a = ap.parse_args()
    tree_digest = source_digest(a.tree)
    data = json.load(open(a.file))
    for key, e in data.items():
        if e.get("isa_sha256"):
            continue
        if e.get("csrc_sha256") != tree_digest:
            print(f"{key}: collected on another tree ({e.get('csrc_sha256', '?')[:12]}), left alone")
            continue
        name = full_name(e, key)
        hit = codeobj.find_kernel(name, a.lib) if name else None
        if hit is None:
            print(f"{key}: kernel {name!r} not found in {a.lib}, left alone")
            continue
        e["kernel_symbol"], e["isa_sha256"] = hit
        now = codeobj.find_kernel(hit[0])
        print(f"{key}: {hit[0]} {hit[1][:16]} | current library: "
              f"{'same machine code' if now and now[1] == hit[1] else 'DIFFERENT (entry is stale)'}")
    with open(a.file, "w") as fh:
        json.dump(data, fh, indent=1)
        fh.write("\n")


if __name__ == "__main__":
    main()
