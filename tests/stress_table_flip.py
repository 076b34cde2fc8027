"""Longer GPU-vs-oracle runs of the TableFlip kernels than the test suite affords: several cell
sizes (small cells name a site twice often -> the duplicate / fallback paths), seeds and walker
counts, single- and multi-sublattice models; prints one line per case and exits non-zero on the
first mismatch.  (The oracle is the checker here, as in tests/.)

  python tests/stress_table_flip.py [--steps 20000]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from smol_amd import capi, mson  # noqa: E402
from smol_amd.engine import Engine  # noqa: E402
from tests.test_table_flip import _model, _neutral_occ  # noqa: E402


def compare(tag, tab, occ, seeds, T, steps):
    cfg = capi.make_config(len(occ), capi.KERNEL_METROPOLIS, capi.STEP_TABLE_FLIP)
    eng, ora = Engine(tab, cfg), orc.OracleMC(tab, cfg)
    eng.set_state(occ, seeds, T)
    ora.set_state(occ, seeds, T)
    done = 0
    for chunk in (1, 17, steps // 3, steps - steps // 3 - 18):
        eng.run(chunk)
        ora.run(chunk)
        done += chunk
        a, b = eng.get_state(), ora.get_state()
        same = (np.array_equal(a["occupancy"], b["occupancy"]) and np.array_equal(a["n_accepted"], b["n_accepted"])
                and np.allclose(a["enthalpy"], b["enthalpy"], rtol=1e-10, atol=1e-7))
        if not same:
            print(f"MISMATCH {tag} after {done} steps ({eng.kernel_info()})")
            sys.exit(1)
    print(f"ok  {tag}: {eng.kernel_info()}, {done} steps x {len(occ)} walkers, acceptance "
          f"{a['n_accepted'].sum() / a['n_steps'].sum():.3f}", flush=True)
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20000)
    a = ap.parse_args()
    for dim in (3, 4, 6):
        for seed in (1, 2):
            for kw in (dict(coef_scale=0.05, mu=[0.1, -0.2, 0.05]), dict(coef_scale=0.05, mu=[0.1, -0.2, 0.05], ewald=True)):
                sc, tab = _model(dim, **kw)
                rng = np.random.default_rng(seed)
                R = 13
                occ = np.array([_neutral_occ(sc, (sc.size & 1) + 2 * (r % 3 + 1), rng) for r in range(R)])
                compare(f"rocksalt {dim}^3 seed {seed} {'ewald' if kw.get('ewald') else 'ce'}", tab, occ,
                        np.arange(R, dtype=np.uint64) + np.uint64(1000 * seed), 2500.0, a.steps)
    # two active sublattices: the reference's LiNiO2 model, flip table from its CompositionSpace
    from smol_amd import moca

    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                        "lno_ce_ewald.mson.json.gz")
    ce = mson.load_mson(gold)
    for dim in (2, 4):
        ens = moca.Ensemble.from_mson(ce, np.diag([dim] * 3))
        s = moca.Sampler.from_ensemble(ens, temperature=1500.0, nwalkers=9, step_type="table-flip",
                                       seeds=list(range(9)))
        tab = ens.make_tables(**s.mckernels[0].usher_kwargs)
        cell = ens.processor.supercell
        P = cell.size
        rng = np.random.default_rng(dim)
        occ = np.ones((9, cell.num_sites), dtype=np.int32)
        occ[:, 2 * P:] = 0
        for r in range(9):
            n = P // 2
            occ[r, rng.permutation(P)[:n]] = 0
            occ[r, P + rng.permutation(P)[:n]] = 0
        compare(f"LiNiO2 {dim}^3", tab, occ, np.arange(9, dtype=np.uint64) + np.uint64(77), 1500.0, a.steps // 4)


if __name__ == "__main__":
    main()
