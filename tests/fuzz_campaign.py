#!/usr/bin/env python3
"""Randomised differential campaign: random models (lattice, species, cutoffs, supercell shape,
processor type, Ewald term, chemical potentials), random kernels (Metropolis / Wang-Landau,
Flip / Swap / TableFlip, the three bias terms, restricted sites, split sublattices) and random
dispatch overrides (auto / SMOLMC_FORCE_GENERAL / SMOLMC_FORCE_UNIVERSAL), each stepped on the GPU
engine and on the CPU oracle with the same Philox streams: occupancies, accept counters and
Wang-Landau histograms bit-equal, enthalpies / features / bias / entropies to 1e-10.

    python tests/fuzz_campaign.py [--cases 200] [--seed 1] [--minutes 10] [--profile any|lean|big|fast|univ] [--out gpurun_out/fuzz.jsonl]

The campaign itself is time-boxed and not collected by pytest (tests/test_gpu_fuzz_campaign.py runs a
fixed handful of its cases); the oracle is the checker here, as everywhere under tests/.  A failing case
prints the seed that reproduces it:  python tests/fuzz_campaign.py --only <case seed>."""
import argparse
import json
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from smol_amd import capi, moca, synth  # noqa: E402
from smol_amd.engine import Engine  # noqa: E402

RTOL, ATOL = 1e-10, 1e-8


def pick(rng, seq):
    return seq[int(rng.integers(len(seq)))]


def neutral_rocksalt(sc, rng, cation_charges, anion):
    """Random charge-neutral occupancy of a rocksalt cell (cations of the given charges against
    the anion charges of ``anion``), or None when the draw cannot be balanced."""
    P = sc.size
    occ = np.zeros(sc.num_sites, dtype=np.int32)
    an_codes = rng.integers(0, len(anion), P) if len(anion) > 1 else np.zeros(P, dtype=np.int64)
    occ[P:2 * P] = an_codes
    need = -float(np.sum(np.asarray(anion, float)[an_codes]))
    q = np.array([0.0 if c is None else c for c in cation_charges])
    for _ in range(200):
        cat = rng.integers(0, len(q), P)
        tot = q[cat].sum()
        # repair by single substitutions
        for _ in range(4 * P):
            if abs(tot - need) < 1e-9:
                break
            i = int(rng.integers(P))
            c = int(rng.integers(len(q)))
            new = tot - q[cat[i]] + q[c]
            if abs(new - need) < abs(tot - need):
                tot, cat[i] = new, c
        if abs(tot - need) < 1e-9:
            occ[:P] = cat
            return occ
    return None


def build_case(rng, profile="any"):
    """``profile`` "lean": larger unaliased cells, no restrictions / overrides (the lean kernel families);
    "big": the same on cells of 2000-14000 sites.
    -> dict(desc, ens, tab, cfg, occ, seeds, temps, env, bias) or None when the draw is void."""
    desc = {}
    # "fast": the shapes of the specialised kernel families the BASELINE configurations and the reference's own
    # model run on -- two active sublattices (lean-multi), TableFlip on one and on two sublattices (table /
    # table-multi), Wang-Landau on one class (mc_wl_kernel) and on several / with update_period 3 (the
    # Wang-Landau variant of the multi-class kernel) -- on unaliased cells, the handle's own kernel only
    fast = profile == "fast"
    ionic = fast or rng.random() < 0.6
    if ionic:
        cations = pick(rng, [(1.0, 3.0, 4.0), (1.0, 3.0), (1.0, 3.0, None), (1.0, 3.0, 4.0, 5.0), (2.0, 4.0)])
        anion = pick(rng, [(-2.0,), (-2.0, -1.0), (-2.0, -1.0)] if fast else [(-2.0,), (-2.0,), (-2.0, -1.0)])
        prim = synth.rocksalt_prim(cation_charges=cations, anion_charges=anion)
        cut = {2: float(rng.uniform(4.3, 6.5))}
        if rng.random() < 0.6:
            cut[3] = float(rng.uniform(3.0, 4.6))
        desc.update(lattice="rocksalt", cations=[c for c in cations], anion=list(anion))
    else:
        S = int(rng.integers(2, 5))
        prim = synth.fcc_prim(nspecies=S)
        cut = {2: float(rng.uniform(3.0, 6.2))}
        if rng.random() < 0.6:
            cut[3] = float(rng.uniform(2.95, 5.1))
            if rng.random() < 0.3:
                cut[4] = float(rng.uniform(2.95, 3.3))
        desc.update(lattice="fcc", nspecies=S)
    desc["cutoffs"] = cut
    big = profile == "big"  # cells of 2000-14000 sites: potential field in HBM, pending-update lists, gx tables
    # round 6: "aliased" -- cells of two or three primitive cells per direction, shorter than their clusters (a cluster
    # row holds a site twice): the lean families with the site's own positions folded into the slot tables;
    # "relabel" -- restricted sites / a sublattice split by species on every case: the site renumbering behind the
    # C-ABI.  Both on the handle's own kernel only.
    aliased_p, relabel_p = profile == "aliased", profile == "relabel"
    lean = profile == "lean" or big or fast or relabel_p
    dims = ([int(rng.integers(9, 16)) for _ in range(3)] if big else
            [int(rng.integers(2, 4)) for _ in range(3)] if aliased_p else
            [int(rng.integers(4, 9)) for _ in range(3)] if fast else
            [int(rng.integers(4, 11)) for _ in range(3)] if lean else [int(rng.integers(2, 7)) for _ in range(3)])
    if not lean and rng.random() < 0.15:
        scm = np.diag(dims)
        scm[0, 1], scm[1, 2] = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        scm = scm.tolist()
    else:
        scm = dims
    desc["supercell"] = scm
    model = synth.build_cluster_model(prim, cut)
    sc = synth.build_supercell(model, scm)
    if sc.num_sites > (14000 if big else 2200 if lean else 600) or (big and sc.num_sites < 2000):
        return None
    coefs = synth.random_coefs(model, seed=int(rng.integers(1 << 30)))
    ptype = "decomposition" if fast else pick(rng, ["decomposition", "decomposition", "decomposition", "expansion"] if lean else ["decomposition", "expansion"])
    use_ewald = ionic and rng.random() < 0.6
    desc.update(processor=ptype, ewald=use_ewald)
    ens = moca.Ensemble.from_cluster_expansion(sc, coefs, processor_type=ptype,
                                               ewald_coefficient=float(rng.uniform(0.05, 0.4)) if use_ewald else None)
    kernel = pick(rng, ["metropolis", "metropolis", "wang-landau"])
    steps = ["flip", "swap"]
    table_ok = ionic and None not in desc["cations"]
    if table_ok:
        steps += ["table-flip", "table-flip"] + (["table-flip"] * 2 if fast else [])
    step = pick(rng, steps)
    if fast:  # (Wang-Landau TableFlip: on the lean table kernel since round 6, part of this profile)
        kernel = pick(rng, ["metropolis", "wang-landau", "wang-landau"])
    desc.update(kernel=kernel, step=step)
    R = int(rng.integers(1, 4 if big else 7))
    P = sc.size
    # occupancies
    if ionic and (step == "table-flip" or rng.random() < 0.3):
        occ = [neutral_rocksalt(sc, rng, desc["cations"], desc["anion"]) for _ in range(R)]
        if any(o is None for o in occ):
            return None
        occ = np.array(occ)
    else:
        nsp = np.array([prim.nspecies[b] for b in sc.site_b])
        occ = (rng.random((R, sc.num_sites)) * nsp).astype(np.int32)
    # split the first active sublattice by species (ensemble.py:288-321): every walker must then
    # share the partition, so the walkers permute walker 0's species inside it
    if step != "table-flip" and rng.random() < (0.4 if relabel_p else 0.12):
        sub_id = next(i for i, s in enumerate(ens.sublattices) if s.is_active)
        sub = ens.sublattices[sub_id]
        if len(sub.species) >= 3:
            codes = list(map(int, sub.encoding))
            k = int(rng.integers(1, len(codes) - 1))
            parts = [codes[:k + 1], codes[k + 1:]]
            for w in range(1, R):
                occ[w] = occ[0]
                for part in parts:
                    sites = sub.sites[np.isin(occ[0, sub.sites], part)]
                    occ[w, sites] = rng.permutation(occ[0, sites])
            ens.split_sublattice_by_species(sub_id, occ[0], parts)
            desc["split"] = parts
    if not ens.active_sublattices:
        return None
    if step == "flip" or rng.random() < 0.2:
        ens.chemical_potentials = {sp: float(rng.uniform(-0.3, 0.3)) for sp in ens.species}
        desc["mu"] = True
    if rng.random() < (1.0 if relabel_p and "split" not in desc else 0.1 if fast else 0.3 if lean else 0.25):
        act = np.concatenate([s.active_sites for s in ens.active_sublattices])
        ens.restrict_sites(rng.choice(act, size=max(1, len(act) // 10), replace=False))
        desc["restricted"] = True
        if not ens.active_sublattices:
            return None
    usher = {}
    if step == "table-flip":
        try:
            usher["flip_table"] = ens.composition_space().flip_table
        except Exception as e:  # (a site space the solver refuses)
            desc["void"] = f"composition space: {e}"
            return None
        if len(usher["flip_table"]) == 0:
            return None
        usher["swap_weight"] = float(pick(rng, [0.0, 0.1, 0.4]))
        if rng.random() < 0.3:
            usher["flip_weights"] = rng.uniform(0.5, 2.0, len(usher["flip_table"]))
    tab = ens.make_tables(**usher)
    # engine and oracle get the SAME tables in the caller's numbering: where restricted sites / a split sublattice
    # scatter the active sites, smolmc_create renumbers them behind the C-ABI (kernel_info: "relabelled=1")
    # (half of the unforced cases switch the renumbering off: scattered layouts on mc_kernel / the universal kernel)
    tab_engine = tab
    if not (lean or aliased_p or rng.random() < 0.5):
        desc["no_relabel"] = True
    bias = None
    if kernel == "metropolis" and rng.random() < 0.35 and not (fast and step == "table-flip"):
        kind = pick(rng, ["fugacity", "square-charge"] if fast else ["fugacity", "square-charge", "square-hyperplane"])
        if kind == "fugacity":
            fr = []
            for s in ens.active_sublattices:
                while True:  # (the reference demands sum == 1 exactly, bias.py:161-162)
                    k = rng.multinomial(16, np.ones(len(s.species)) / len(s.species)) + 1
                    w = k / k.sum()
                    if sum(float(x) for x in w) == 1:
                        break
                fr.append({sp: float(x) for sp, x in zip(s.species, w)})
            bias = moca.FugacityBias(ens.sublattices, fr)
        elif kind == "square-charge":
            bias = moca.SquareChargeBias(ens.sublattices, penalty=float(rng.uniform(0.01, 0.2)))
        else:
            d = sum(len(s.species) for s in ens.sublattices)
            rows = int(rng.integers(1, 3))
            bias = moca.SquareHyperplaneBias(ens.sublattices, rng.integers(-2, 3, (rows, d)),
                                             rng.integers(-3, 4, rows), penalty=float(rng.uniform(0.001, 0.02)))
        for tb in ({id(tab): tab, id(tab_engine): tab_engine}).values():
            tb.set_bias(bias.bias_type, bias._table, bias.penalty, intercepts=getattr(bias, "intercepts", None))
        desc["bias"] = kind
    st = moca.STEP_TYPES[step]
    wl_kw = None
    if kernel == "metropolis":
        cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, st)
    else:
        from oracle import oracle as orc

        probe = orc.OracleEvaluator(tab)
        h = np.array([probe.natural_parameters() @ probe.feature_vector(o) for o in occ])
        up = int(pick(rng, [1, 1, 1, 3]))
        wl_kw = dict(min_enthalpy=float(h.min()) - 3.0371, max_enthalpy=float(h.max()) + 3.0113,
                     bin_size=float(pick(rng, [0.25, 0.5, 0.11])), check_period=int(pick(rng, [50, 20, 1000])),
                     update_period=up, flatness=float(pick(rng, [0.8, 0.3])))
        cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, st, **wl_kw)
        desc["update_period"] = up
    env = None if (lean or aliased_p) else pick(rng, [None, None, None, "SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL"])
    if profile == "univ":  # the models of `any`, every one on the universal kernel (rewritten in round 5)
        env = "SMOLMC_FORCE_UNIVERSAL"
    desc.update(walkers=R, sites=int(sc.num_sites), env=env)
    seeds = rng.integers(1, 2 ** 62, size=R).astype(np.uint64)
    temps = rng.uniform(400.0, 6000.0, size=R)
    return dict(desc=desc, ens=ens, tab=tab, tab_engine=tab_engine, cfg=cfg, occ=occ, seeds=seeds, temps=temps, env=env, bias=bias,
                wl=kernel == "wang-landau", usher=usher, wl_kw=wl_kw, step=step)


def run_case(case_seed, profile="any"):
    """One case; a mismatch / refusal / crash comes back as status FAIL with the case description
    and the stage it happened in."""
    rng = np.random.default_rng(case_seed)
    case = build_case(rng, profile)
    if case is None:
        return dict(seed=case_seed, status="void")
    try:
        return _run_case(case_seed, case, rng)
    except Exception as e:
        return dict(seed=case_seed, status="FAIL", error=f"{type(e).__name__}: {e}".strip().splitlines()[0][:300],
                    desc=case["desc"], trace=traceback.format_exc().splitlines()[-8:])


def _run_case(case_seed, case, rng):
    from oracle import oracle as orc

    desc = case["desc"]
    for name in ("SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL"):
        os.environ.pop(name, None)
    if case["env"]:
        os.environ[case["env"]] = "1"
    if desc.get("no_relabel"):
        os.environ["SMOLMC_NO_SITE_RELABEL"] = "1"
    try:
        eng = Engine(case["tab_engine"], case["cfg"])
    finally:
        for name in ("SMOLMC_FORCE_GENERAL", "SMOLMC_FORCE_UNIVERSAL", "SMOLMC_NO_SITE_RELABEL"):
            os.environ.pop(name, None)
    ora = orc.OracleMC(case["tab"], case["cfg"])
    desc["kernel_info"] = eng.kernel_info()
    if "relabelled=1" in desc["kernel_info"]:
        desc["relabelled"] = True
    try:
        eng.set_state(case["occ"], case["seeds"], case["temps"])
    except Exception as e:
        # the oracle must refuse the same state (e.g. a walker outside the Wang-Landau window,
        # an infeasible TableFlip start)
        try:
            ora.set_state(case["occ"], case["seeds"], case["temps"])
        except Exception:
            return dict(seed=case_seed, status="void", desc=desc, why=f"both refuse the state: {e}")
        raise
    ora.set_state(case["occ"], case["seeds"], case["temps"])
    total = 0
    desc["stage"] = "native steps"
    for chunk in (1, 7, 64, 200):
        eng.run(chunk)
        ora.run(chunk)
        total += chunk
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"]), f"occupancies differ after {total} steps"
        assert np.array_equal(a["n_accepted"], b["n_accepted"]), f"accept counters differ after {total} steps"
        np.testing.assert_allclose(a["enthalpy"], b["enthalpy"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(a["features"], b["features"], rtol=RTOL, atol=ATOL)
        if case["bias"] is not None:
            np.testing.assert_allclose(eng.get_bias(), ora.get_bias(), rtol=RTOL, atol=ATOL)
    # device-side thinning (smolmc_run_sampled): every recorded row is the oracle's state at that step
    desc["stage"] = "sampled"
    ns, thin = int(rng.integers(1, 5)), int(rng.integers(1, 40))
    # (ABI 7: with the `bias` column / the Wang-Landau trace of every sample where the kernel has them)
    ring = eng.run_sampled(ns, thin, occupancy=True, bias=case["bias"] is not None, wl=case["wl"])
    for i in range(ns):
        ora.run(thin)
        b = ora.get_state()
        assert np.array_equal(ring["occupancy"][i], b["occupancy"]), f"sampled row {i} of {ns} (thin {thin}) differs"
        np.testing.assert_allclose(ring["enthalpy"][i], b["enthalpy"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(ring["features"][i], b["features"], rtol=RTOL, atol=ATOL)
        assert np.array_equal(ring["accepted"][i], b["accepted"])
        if case["bias"] is not None:
            np.testing.assert_allclose(ring["bias"][i], ora.get_bias(), rtol=RTOL, atol=ATOL)
        if case["wl"]:
            wb = ora.get_wl()
            assert np.array_equal(ring["histogram"][i], wb["histogram"]), f"sampled WL histogram {i} differs"
            assert np.array_equal(ring["occurrences"][i], wb["occurrences"])
            np.testing.assert_allclose(ring["entropy"][i], wb["entropy"], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(ring["mean_features"][i], wb["mean_features"], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(ring["mod_factor"][i], wb["mod_factor"], rtol=0, atol=0)
    # the running features are the recomputed ones (checked BEFORE the replay: a record that flips a
    # site twice prices the chemical work of both flips against the occupancy before the step,
    # ensemble.py:368-374, so the running feature legitimately leaves the recomputed one there)
    desc["stage"] = "recomputed features"
    b = eng.get_state()
    for w in range(len(case["occ"])):
        np.testing.assert_allclose(b["features"][w], case["ens"].compute_feature_vector(b["occupancy"][w]), rtol=1e-9, atol=1e-7)
    # replayed records (smolmc_replay) on Flip handles: single flips (the handle's own kernel) or
    # records of 0..8 flips, a site possibly twice (the universal kernel), then native steps again
    if desc["step"] == "flip" and rng.random() < 0.6:
        ens = case["ens"]
        act = np.concatenate([s.active_sites for s in ens.active_sublattices])
        codes_of = {}
        for sub in ens.active_sublattices:
            for site in sub.active_sites:
                codes_of[int(site)] = sub.encoding
        R, n = len(case["occ"]), int(rng.integers(1, 120))
        multi = rng.random() < 0.5
        steps = -np.ones((R, n, 16), dtype=np.int32)
        for r in range(R):
            for k in range(n):
                nf = int(rng.integers(0, 9)) if multi else 1
                sites = rng.choice(act, size=nf) if nf else []
                if nf >= 3 and rng.random() < 0.4:
                    sites[-1] = sites[0]
                for j, site in enumerate(sites):
                    steps[r, k, 2 * j], steps[r, k, 2 * j + 1] = site, pick(rng, codes_of[int(site)])
        us = rng.random((R, n))
        desc["stage"] = "replay " + ("multi" if multi else "single")
        a_acc, a_H = eng.replay(steps, us)
        b_acc, b_H = ora.replay(steps, us)
        assert np.array_equal(a_acc, b_acc), "replayed accept flags differ"
        np.testing.assert_allclose(a_H, b_H, rtol=RTOL, atol=ATOL)
        desc["replayed"] = "multi" if multi else "single"
        a, b = eng.get_state(), ora.get_state()
        assert np.array_equal(a["occupancy"], b["occupancy"]), "occupancies differ after the replay"
        np.testing.assert_allclose(a["features"], b["features"], rtol=RTOL, atol=ATOL)
        desc["stage"] = "native steps after replay"
        eng.run(33)
        ora.run(33)
    a, b = eng.get_state(), ora.get_state()
    assert np.array_equal(a["occupancy"], b["occupancy"]), "final occupancies differ"
    assert np.array_equal(a["n_accepted"], b["n_accepted"]), "final accept counters differ"
    np.testing.assert_allclose(a["features"], b["features"], rtol=RTOL, atol=ATOL)
    if case["bias"] is not None:
        np.testing.assert_allclose(eng.get_bias(), ora.get_bias(), rtol=RTOL, atol=ATOL)
    if case["wl"]:
        wa, wb = eng.get_wl(), ora.get_wl()
        assert np.array_equal(wa["histogram"], wb["histogram"]), "WL histograms differ"
        assert np.array_equal(wa["occurrences"], wb["occurrences"]), "WL occurrences differ"
        np.testing.assert_allclose(wa["entropy"], wb["entropy"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(wa["mean_features"], wb["mean_features"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(wa["mod_factor"], wb["mod_factor"], rtol=0, atol=0)
    acc = float(a["n_accepted"].sum()) / float(a["n_steps"].sum())
    desc.pop("stage")
    return dict(seed=case_seed, status="ok", acceptance=acc, desc=desc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--only", type=int, default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--profile", default="any", choices=["any", "lean", "big", "fast", "univ", "aliased", "relabel"])
    args = ap.parse_args()
    seeds = [args.only] if args.only is not None else [args.seed * 1000003 + i for i in range(args.cases)]
    t0 = time.time()
    counts = {"ok": 0, "void": 0, "FAIL": 0}
    kernels, not_lean = {}, {}
    unforced = {"lean": 0, "other": 0, "relabelled": 0}  # cases on the handle's own kernel (no SMOLMC_FORCE_*, renumbering on)
    out = open(args.out, "w") if args.out else None
    for s in seeds:
        if time.time() - t0 > 60.0 * args.minutes:
            break
        try:
            res = run_case(s, args.profile)
        except Exception as e:  # a mismatch, a refusal or a crash of the host code: all are findings
            res = dict(seed=s, status="FAIL", error=f"{type(e).__name__}: {e}".splitlines()[0][:300],
                       trace=traceback.format_exc().splitlines()[-6:])
        counts[res["status"]] += 1
        if res["status"] == "ok":
            d = res["desc"]  # kernel family / step type (/ Wang-Landau): which kernels the campaign actually reached
            k = d["kernel_info"].split()[0] + "/" + d["step"] + ("/wl" if d["kernel"] == "wang-landau" else "")
            kernels[k] = kernels.get(k, 0) + 1
            if not d.get("env") and not d.get("no_relabel"):
                unforced["lean" if d["kernel_info"].startswith("lean") else "other"] += 1
                unforced["relabelled"] += int(bool(d.get("relabelled")))
            if " | not lean: " in d["kernel_info"] and not d.get("env"):  # (why the model left the lean families, unforced cases)
                why = d["kernel_info"].split(" | not lean: ", 1)[1]
                not_lean[why] = not_lean.get(why, 0) + 1
        if res["status"] == "FAIL" or args.only is not None:
            print(json.dumps(res, default=str), flush=True)
        if out:
            out.write(json.dumps(res, default=str) + "\n")
            out.flush()
    nun = unforced["lean"] + unforced["other"]
    summary = dict(cases=sum(counts.values()), **counts, kernels=kernels, not_lean=not_lean,
                   unforced=dict(unforced, lean_share=round(unforced["lean"] / nun, 3) if nun else None),
                   seconds=round(time.time() - t0, 1), first_seed=seeds[0])
    print(json.dumps(summary))
    if out:
        out.write(json.dumps(dict(summary=summary)) + "\n")
    return 1 if counts["FAIL"] else 0


if __name__ == "__main__":
    sys.exit(main())
