#!/usr/bin/env python3
"""Host-side model of LDS bank conflicts of the lean kernel's occupancy gathers.

For a site s, the wave issues NSLOT*MM ds_read_u8 instructions; instruction (it, m) reads,
in lane l, the byte of member m of the lane's slot it.  A ds_read_u8/b32 is serviced in two
32-lane groups; each group costs max over the 32 banks of the number of DISTINCT dwords
requested on that bank.  This script evaluates candidate site->LDS-address permutations."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smol_amd import synth


def lean_rows(sc):
    """[N][64][NSLOT][MM] member sites in the kernel's slot order (mirrors build_mc_tables)."""
    loc = sc.local_tables()
    N = sc.num_sites
    rows_all = None
    for s in range(N):
        slots = []
        for pos, rows, ratio in loc[s]:
            rec = []
            for row in rows:
                p = int(np.flatnonzero(row == s)[-1])
                rec.append((p, [int(x) for i, x in enumerate(row) if i != p]))
            rec.sort(key=lambda t: t[0])
            slots += rec
        slots.sort(key=lambda t: -len(t[1]))
        if rows_all is None:
            C = len(slots)
            NSL = 2 if C <= 128 else 4
            MM = max(2, max(len(t[1]) for t in slots))
            rows_all = np.tile(np.arange(N)[:, None, None, None], (1, 64, NSL, MM))
        for q, (_, mem) in enumerate(slots):
            for m, x in enumerate(mem):
                rows_all[s, q % 64, q // 64, m] = x
    return rows_all


def cost(rows, addr_of, sites):
    tot = 0
    for s in sites:
        r = addr_of[rows[s]]  # [64][NSL][MM]
        for it in range(r.shape[1]):
            for m in range(r.shape[2]):
                a = r[:, it, m]
                for g in (a[:32], a[32:]):
                    dw = np.unique(g >> 2)
                    tot += np.bincount(dw & 31, minlength=32).max()
    return tot / len(sites)


def main():
    model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [16, 16, 16])
    rows = lean_rows(sc)
    N = sc.num_sites
    sites = np.random.default_rng(0).choice(N, 64, replace=False)
    ident = np.arange(N)
    ideal = rows.shape[2] * rows.shape[3] * 2
    print("instructions per flip:", rows.shape[2] * rows.shape[3], "ideal LDS cycles:", ideal)
    print("identity            :", cost(rows, ident, sites))
    for sh in (2, 3, 4):
        for src in (4, 8):
            sw = ident ^ (((ident >> src) & 15) << sh)
            print(f"xor bits{src}.. << {sh}     :", cost(rows, sw, sites))
    sw = ident ^ (((ident >> 8) & 15) << 2) ^ (((ident >> 4) & 3) << 5)
    print("xor x<<2 ^ y<<5     :", cost(rows, sw, sites))
    rp = np.random.default_rng(1).permutation(N)
    print("random permutation  :", cost(rows, rp, sites))


if __name__ == "__main__":
    main()
