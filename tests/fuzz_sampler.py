#!/usr/bin/env python3
"""Randomised differential campaign one level up from tests/fuzz_campaign.py: the same random models
and kernels, driven through the smol-shaped host API (`moca.Sampler.from_ensemble`, `run`
continuing / restarting / after `clear_samples`, `anneal`, Wang-Landau with a callable `mod_update`,
bias objects, per-walker temperatures) and mirrored call for call on the CPU oracle.  Every recorded
sample must be the oracle's state at that step: occupancies bit-equal, enthalpy / features / bias /
Wang-Landau arrays to 1e-10.  This pins the host bookkeeping -- which state a run continues from,
when the device state may be reused, which steps of a run are never taken (`nsteps % thin_by`), that
seeds / counters / Wang-Landau arrays persist across runs of one sampler (sampler.py:254-262).

    python tests/fuzz_sampler.py [--cases 300] [--seed 1] [--minutes 8] [--only <case seed>]

Time-boxed, not collected by pytest; tests/test_gpu_fuzz_campaign.py runs a fixed handful."""
import argparse
import json
import os
import sys
import time
import traceback
import warnings

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from smol_amd import moca  # noqa: E402
from tests import fuzz_campaign as fc  # noqa: E402

RTOL, ATOL = 1e-10, 1e-8
ENV = ("SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL")


def make_sampler(case, rng):
    ens, R = case["ens"], len(case["occ"])
    kw = dict(step_type=case["step"], nwalkers=R, seeds=[int(s) for s in case["seeds"]], **case["usher"])
    desc = case["desc"]
    if case["wl"]:
        w = case["wl_kw"]
        extra = {}
        if w["update_period"] == 1 and rng.random() < 0.4:
            extra["mod_update"] = lambda m: m / 2.0  # host-side flatness checks (wanglandau.py:100-105)
            desc["mod_update"] = "callable"
        sampler = moca.Sampler.from_ensemble(ens, w["min_enthalpy"], w["max_enthalpy"], w["bin_size"],
                                             kernel_type="Wang-Landau", flatness=w["flatness"],
                                             check_period=w["check_period"], update_period=w["update_period"],
                                             **extra, **kw)
    else:
        b = case["bias"]
        if b is not None:  # the bias through the kernel's own arguments (kernel/base.py:229-235)
            kind = desc["bias"]
            bkw = ({"fugacity_fractions": b.fugacity_fractions} if kind == "fugacity" else
                   {"penalty": b.penalty} if kind == "square-charge" else
                   {"hyperplane_normals": b._A, "hyperplane_intercepts": b._b, "penalty": b.penalty})
            kw.update(bias_type=kind, bias_kwargs=bkw)
        sampler = moca.Sampler.from_ensemble(ens, temperature=1000.0, kernel_type="Metropolis", **kw)
        for k, t in zip(sampler.mckernels, case["temps"]):
            k.temperature = float(t)
    return sampler


def _run_case(case, rng):
    from oracle import oracle as orc

    desc = case["desc"]
    R = len(case["occ"])
    sampler = make_sampler(case, rng)
    ora = orc.OracleMC(case["tab"], case["cfg"])
    temps = np.array(case["temps"], dtype=np.float64)
    seeds = case["seeds"]
    try:
        ora.set_state(case["occ"], seeds, temps)
    except Exception as e:
        return dict(status="void", why=f"the oracle refuses the start: {e}")
    expect = []  # oracle states at the recorded samples since the last clear

    def mirror_run(nsteps, thin):
        for _ in range(nsteps // thin):
            ora.run(thin)
            st = ora.get_state()
            row = dict(occupancy=st["occupancy"].copy(), enthalpy=st["enthalpy"].copy(), features=st["features"].copy(),
                       accepted=st["accepted"].copy())
            if case["bias"] is not None:
                row["bias"] = ora.get_bias().copy()
            if case["wl"]:
                row.update({k: v.copy() for k, v in ora.get_wl().items()})
            expect.append(row)

    def check(stage):
        desc["stage"] = stage
        c = sampler.samples
        assert c.num_samples == len(expect), f"{c.num_samples} samples recorded, {len(expect)} expected"
        if not expect:
            return
        occ = c.get_occupancies(flat=False).reshape(len(expect), R, -1)
        H = c.get_enthalpies(flat=False).reshape(len(expect), R)
        feats = c.get_feature_vectors(flat=False).reshape(len(expect), R, -1)
        acc = c.get_trace_value("accepted", flat=False).reshape(len(expect), R)
        for i, row in enumerate(expect):
            assert np.array_equal(occ[i], row["occupancy"]), f"sample {i} of {len(expect)}: occupancies differ"
            np.testing.assert_allclose(H[i], row["enthalpy"], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(feats[i], row["features"], rtol=RTOL, atol=ATOL)
            assert np.array_equal(acc[i], row["accepted"]), f"sample {i}: accepted flags differ"
        if case["bias"] is not None:
            b = c.get_trace_value("bias", flat=False).reshape(len(expect), R)
            np.testing.assert_allclose(b[-1], expect[-1]["bias"], rtol=RTOL, atol=ATOL)
        if case["wl"]:
            last = expect[-1]
            L = last["entropy"].shape[-1]
            np.testing.assert_allclose(c.get_trace_value("entropy", flat=False)[-1].reshape(R, L), last["entropy"], rtol=1e-12, atol=1e-12)
            assert np.array_equal(c.get_trace_value("histogram", flat=False)[-1].reshape(R, L), last["histogram"])
            assert np.array_equal(c.get_trace_value("occurrences", flat=False)[-1].reshape(R, L), last["occurrences"])
            np.testing.assert_allclose(c.get_trace_value("mod_factor", flat=False)[-1].reshape(R), last["mod_factor"], rtol=0, atol=0)
            np.testing.assert_allclose(c.get_trace_value("cumulative_mean_features", flat=False)[-1].reshape(R, L, -1),
                                       last["mean_features"], rtol=RTOL, atol=ATOL)

    for name in ENV:
        os.environ.pop(name, None)
    if case["env"]:
        os.environ[case["env"]] = "1"
    ops = []
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            thin = int(rng.integers(1, 30))
            n = thin * int(rng.integers(1, 6)) + int(rng.integers(0, thin))  # (the remainder is ignored)
            desc["stage"] = "first run"
            sampler.run(n, case["occ"], thin_by=thin)
            desc["kernel_info"] = sampler.engine.kernel_info()
            mirror_run(n, thin)
            ops.append(("run", n, thin))
            check("first run")
            for _ in range(int(rng.integers(2, 6))):
                op = fc.pick(rng, ["continue", "continue", "restart", "clear", "anneal"])
                thin = int(rng.integers(1, 30))
                n = thin * int(rng.integers(1, 5)) + int(rng.integers(0, thin))
                desc["stage"] = op
                if op == "continue":
                    sampler.run(n, thin_by=thin)
                    ora.set_temperature(temps)
                    mirror_run(n, thin)
                elif op == "restart":  # new occupancies on top of the recorded samples: the chain's stream carries on
                    occ2 = case["occ"][rng.permutation(R)] if R > 1 else case["occ"]
                    if "split" in desc:
                        occ2 = case["occ"]
                    sampler.run(n, occ2, thin_by=thin)
                    ora.set_state(occ2, seeds, temps, reset_aux=False)
                    mirror_run(n, thin)
                elif op == "clear":
                    last = sampler.samples.get_occupancies(flat=False)[-1].reshape(R, -1)
                    sampler.clear_samples()
                    expect.clear()
                    sampler.run(n, last, thin_by=thin)
                    ora.set_state(last, seeds, temps, reset_aux=False)
                    mirror_run(n, thin)
                elif op == "anneal":
                    if case["wl"]:
                        continue
                    ladder = sorted(rng.uniform(300.0, 5000.0, int(rng.integers(1, 4))), reverse=True)
                    sampler.anneal(ladder, n, thin_by=thin)
                    for T in ladder:
                        temps = np.full(R, float(T))
                        ora.set_temperature(temps)
                        mirror_run(n, thin)
                ops.append((op, n, thin))
                check(op)
    finally:
        for name in ENV:
            os.environ.pop(name, None)
    desc.pop("stage", None)
    desc["ops"] = ops
    return dict(status="ok", desc=desc)


def run_case(case_seed, profile="any"):
    rng = np.random.default_rng(case_seed)
    case = fc.build_case(rng, profile)
    if case is None:
        return dict(seed=case_seed, status="void")
    try:
        res = _run_case(case, rng)
    except Exception as e:
        res = dict(status="FAIL", error=f"{type(e).__name__}: {e}".strip().splitlines()[0][:300], desc=case["desc"],
                   trace=traceback.format_exc().splitlines()[-8:])
    res["seed"] = case_seed
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--minutes", type=float, default=8.0)
    ap.add_argument("--only", type=int, default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--profile", default="any", choices=["any", "lean"])
    args = ap.parse_args()
    seeds = [args.only] if args.only is not None else [args.seed * 1000003 + 500000 + i for i in range(args.cases)]
    t0 = time.time()
    counts = {"ok": 0, "void": 0, "FAIL": 0}
    kernels, ops = {}, {}
    out = open(args.out, "w") if args.out else None
    for s in seeds:
        if time.time() - t0 > 60.0 * args.minutes:
            break
        res = run_case(s, args.profile)
        counts[res["status"]] += 1
        if res["status"] == "ok":
            k = res["desc"]["kernel_info"].split()[0]
            kernels[k] = kernels.get(k, 0) + 1
            for op in res["desc"]["ops"]:
                ops[op[0]] = ops.get(op[0], 0) + 1
        if res["status"] == "FAIL" or args.only is not None:
            print(json.dumps(res, default=str), flush=True)
        if out:
            out.write(json.dumps(res, default=str) + "\n")
            out.flush()
    summary = dict(cases=sum(counts.values()), **counts, kernels=kernels, operations=ops, seconds=round(time.time() - t0, 1),
                   first_seed=seeds[0])
    print(json.dumps(summary))
    if out:
        out.write(json.dumps(dict(summary=summary)) + "\n")
    return 1 if counts["FAIL"] else 0


if __name__ == "__main__":
    sys.exit(main())
