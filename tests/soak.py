#!/usr/bin/env python3
"""Long-run soak of the two main lean shapes on one GPU: many steps per walker, then the running
trace against a from-scratch evaluation of every walker (Engine.audit_drift), composition
conservation for the canonical shape, and a short oracle continuation of a few walkers from the
final state (same streams).  Usage: python tests/soak.py [--steps N]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smol_amd import capi, ewald, synth  # noqa: E402
from smol_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20_000_000)
    a = ap.parse_args()
    out = {}
    # headline shape
    model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [16] * 3)
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=20260928))
    R = 4096
    rng = np.random.default_rng(1)
    occ = np.zeros((R, sc.num_sites), np.int32)
    for r in range(R):
        occ[r, rng.permutation(sc.num_sites)[: sc.num_sites // 2]] = 1
    eng = Engine(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP))
    eng.set_state(occ, np.arange(R, dtype=np.uint64) + np.uint64(99), 2500.0)
    t0 = time.time()
    eng.run(a.steps)
    eng.sync()
    dt = time.time() - t0
    st = eng.get_state()
    df, dh = eng.audit_drift()
    out["canonical"] = dict(kernel=eng.kernel_info(), steps=a.steps, seconds=dt, steps_per_s=R * a.steps / dt,
                            acceptance=float(st["n_accepted"].sum() / st["n_steps"].sum()),
                            feature_drift=df, enthalpy_drift=dh,
                            composition_conserved=bool((st["occupancy"].sum(axis=1) == sc.num_sites // 2).all()))
    # a few walkers continued by the oracle from the final state, same streams
    from oracle import oracle as orc

    k = 4
    sub_cfg = capi.make_config(k, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
    ora, e2 = orc.OracleMC(tab, sub_cfg), Engine(tab, sub_cfg)
    for e in (ora, e2):
        e.set_state(st["occupancy"][:k], np.arange(k, dtype=np.uint64) + np.uint64(99), 2500.0)
        e.run(2000)
    out["canonical"]["oracle_continuation_equal"] = bool(
        np.array_equal(ora.get_state()["occupancy"], e2.get_state()["occupancy"]))
    eng.close()
    # Ewald-field shape (config 3), fewer steps
    model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [12] * 3)
    mu = np.zeros((sc.num_sites, 3))
    mu[: sc.size] = np.random.default_rng(7).uniform(-0.5, 0.5, 3)[None, :]
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model), ewald=ewald.supercell_ewald(sc), ewald_coef=0.1,
                                   mu_table=mu)
    R = 2048
    nsp = np.array([model.prim.nspecies[b] for b in sc.site_b])
    occ = (np.random.default_rng(3).random((R, sc.num_sites)) * nsp).astype(np.int32)
    eng = Engine(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_FLIP))
    eng.set_state(occ, np.arange(R, dtype=np.uint64) + np.uint64(5), 3000.0)
    n3 = max(1, a.steps // 10)
    t0 = time.time()
    eng.run(n3)
    eng.sync()
    dt = time.time() - t0
    st = eng.get_state(occupancy=False)
    df, dh = eng.audit_drift()
    out["semigrand_ewald"] = dict(kernel=eng.kernel_info(), steps=n3, seconds=dt, steps_per_s=R * n3 / dt,
                                  acceptance=float(st["n_accepted"].sum() / st["n_steps"].sum()),
                                  feature_drift=df, enthalpy_drift=dh)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
