#!/usr/bin/env python3
"""Generate golden vectors by driving the REFERENCE's compiled native core.

Run in the build container only (needs /root/reference, Cython, gcc):

    python tests/golden/make_golden.py

What it does
------------
1. Builds the reference's five Cython sources (smol/utils/cluster/{evaluator,
   container,ewald,correlations}.pyx + struct.pxd, smol/utils/_openmp_helpers.pyx)
   OUT OF TREE in a scratch directory under /tmp with the reference's own flags
   (-O3 -ffast-math -fopenmp, setup.py:18-25).  No reference source or binary is
   written into this repository; nothing built here travels to the GPU box.
2. Feeds the build's own synthetic tables (smol_amd.synth / smol_amd.ewald) to
   ClusterSpaceEvaluator.{correlations,interactions}_from_occupancy,
   delta_{correlations,interactions}_from_occupancies (evaluator.pyx:121-317),
   delta_ewald_single_flip (ewald.pyx:9-59) and the legacy list-of-tuples
   functions (correlations.pyx:18,61,164,209,308), and records inputs + outputs.
3. Records replayed Metropolis / Wang-Landau trajectories whose arithmetic (every
   delta) comes from the reference core and whose control flow restates
   smol/moca/kernel/{base,metropolis,wanglandau,mcusher}.py with the same
   numpy Generator (PCG64) call order as the reference ushers.

The Python half of the reference (smol.moca / smol.cofe) cannot be imported here
(pymatgen, monty, h5py absent), so step 3's control flow is a restatement pinned
by the reference-core arithmetic; see DESIGN.md "Oracle".

Outputs: tests/golden/*.npz (data only: inputs and expected outputs).
"""

import importlib
import os
import shutil
import subprocess
import sys
from math import log

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference"
SCRATCH = "/tmp/smol_ref_core_build"
kB = 8.617333262145e-5  # smol/constants.py:4


def build_reference_core():
    pkg = os.path.join(SCRATCH, "smol", "utils", "cluster")
    os.makedirs(pkg, exist_ok=True)
    for rel in [
        "smol/utils/cluster/evaluator.pyx",
        "smol/utils/cluster/evaluator.pxd",
        "smol/utils/cluster/container.pyx",
        "smol/utils/cluster/container.pxd",
        "smol/utils/cluster/struct.pxd",
        "smol/utils/cluster/ewald.pyx",
        "smol/utils/cluster/correlations.pyx",
        "smol/utils/_openmp_helpers.pyx",
    ]:
        dst = os.path.join(SCRATCH, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(REF, rel), dst)
    for d in ["smol", "smol/utils", "smol/utils/cluster"]:
        open(os.path.join(SCRATCH, d, "__init__.py"), "a").close()
    setup_py = os.path.join(SCRATCH, "setup_core.py")
    with open(setup_py, "w") as fh:
        fh.write(
            "import numpy\n"
            "from setuptools import setup, Extension\n"
            "from Cython.Build import cythonize\n"
            "flags=['-O3','-ffast-math','-fopenmp']\n"
            "names=['smol.utils.cluster.evaluator','smol.utils.cluster.container',"
            "'smol.utils.cluster.ewald','smol.utils.cluster.correlations',"
            "'smol.utils._openmp_helpers']\n"
            "exts=[Extension(n,[n.replace('.','/')+'.pyx'],include_dirs=[numpy.get_include()],"
            "extra_compile_args=flags,extra_link_args=['-fopenmp'],"
            "define_macros=[('NPY_NO_DEPRECATED_API','NPY_1_7_API_VERSION')]) for n in names]\n"
            "setup(name='smolcore',ext_modules=cythonize(exts,language_level=3,"
            "compiler_directives={'boundscheck':False,'wraparound':False,"
            "'initializedcheck':False,'cdivision':True}))\n"
        )
    marker = os.path.join(SCRATCH, ".built")
    if not os.path.exists(marker):
        subprocess.check_call(
            [sys.executable, "setup_core.py", "build_ext", "--inplace", "-q"], cwd=SCRATCH
        )
        open(marker, "w").close()
    sys.path.insert(0, SCRATCH)
    ev = importlib.import_module("smol.utils.cluster.evaluator")
    ct = importlib.import_module("smol.utils.cluster.container")
    ew = importlib.import_module("smol.utils.cluster.ewald")
    lg = importlib.import_module("smol.utils.cluster.correlations")
    return ev, ct, ew, lg


# --------------------------------------------------------------------------- #
class RefProcessor:
    """Drives the reference core the way smol/moca/processor/expansion.py does."""

    def __init__(self, core, model, sc, coefs, ewald=None, ewald_coef=None, mu_table=None):
        ev, ct, ew, lg = core
        self.core = core
        self.model, self.sc = model, sc
        self.size = sc.size
        self.coefs = np.asarray(coefs, float)
        self.itens = model.cluster_interaction_tensors(coefs)
        flat_it = tuple(np.ravel(t, order="C") for t in self.itens[1:])
        od = model.orbit_data()
        # full evaluators: expansion.py:106-110 and :328-337
        self.ev_corr = ev.ClusterSpaceEvaluator(od, model.num_orbits, model.num_corr_functions)
        self.ev_int = ev.ClusterSpaceEvaluator(
            od, model.num_orbits, model.num_corr_functions, 1, self.itens[0], flat_it
        )
        self.full_arrays = tuple(sc.full_indices)
        self.full_cont = ct.IntArray2DContainer(self.full_arrays)
        # local evaluators per site: expansion.py:120-156 / :343-389
        self.local = {}
        for s, data in sc.local_tables().items():
            lod = tuple(od[pos] for pos, _, _ in data)
            idx = tuple(rows for _, rows, _ in data)
            ratio = np.array([r for _, _, r in data])
            lit = tuple(flat_it[pos] for pos, _, _ in data)
            e_c = ev.ClusterSpaceEvaluator(lod, model.num_orbits, model.num_corr_functions)
            e_i = ev.ClusterSpaceEvaluator(
                lod, model.num_orbits, model.num_corr_functions, 1, self.itens[0], lit
            )
            self.local[s] = (e_c, e_i, idx, ct.IntArray2DContainer(idx), ratio, lod, lit)
        self.ewald = ewald  # (inds, matrix)
        self.ewald_coef = ewald_coef
        self.mu_table = mu_table

    # expansion.py:165-189 / :391-414
    def corr_full(self, occ):
        return self.ev_corr.correlations_from_occupancy(occ, self.full_cont) * self.size

    def int_full(self, occ):
        return self.ev_int.interactions_from_occupancy(occ, self.full_cont) * self.size

    # expansion.py:191-231 / :420-464 (sequential flips)
    def delta(self, occ, flips, mode):
        occu_i = np.array(occ, dtype=np.int32)
        n = self.model.num_corr_functions if mode == "corr" else self.model.num_orbits
        out = np.zeros(n)
        for site, code in flips:
            occu_f = occu_i.copy()
            occu_f[site] = code
            e_c, e_i, _, cont, ratio, _, _ = self.local[site]
            if mode == "corr":
                out += e_c.delta_correlations_from_occupancies(occu_f, occu_i, ratio, cont)
            else:
                out += e_i.delta_interactions_from_occupancies(occu_f, occu_i, ratio, cont)
            occu_i = occu_f
        return out * self.size

    # legacy correlations.pyx:61 / :209
    def delta_legacy(self, occ, flips, mode):
        lg = self.core[3]
        occu_i = np.array(occ, dtype=np.int32)
        n = self.model.num_corr_functions if mode == "corr" else self.model.num_orbits
        out = np.zeros(n)
        for site, code in flips:
            occu_f = occu_i.copy()
            occu_f[site] = code
            _, _, idx, _, ratio, lod, lit = self.local[site]
            if mode == "corr":
                lst = [
                    (d[1], float(r), d[3], d[2], rows) for d, rows, r in zip(lod, idx, ratio)
                ]
                out += lg.delta_corr_single_flip(
                    occu_f, occu_i, self.model.num_corr_functions, lst
                )
            else:
                lst = [
                    (d[0], float(r), d[3], t, rows)
                    for d, rows, r, t in zip(lod, idx, ratio, lit)
                ]
                out += lg.delta_interactions_single_flip(
                    occu_f, occu_i, self.model.num_orbits, lst
                )
            occu_i = occu_f
        return out * self.size

    def corr_full_legacy(self, occ):
        lg = self.core[3]
        lst = [
            (o.bit_id, o.flat_tensor_indices, o.flat_correlation_tensors, rows)
            for o, rows in zip(self.model.orbits, self.full_arrays)
        ]
        return lg.corr_from_occupancy(occ, self.model.num_corr_functions, lst) * self.size

    # processor/ewald.py:128-182
    def ewald_full(self, occ):
        inds, mat = self.ewald
        sel = inds[np.arange(len(occ)), occ]
        mask = np.zeros(mat.shape[0], dtype=bool)
        mask[sel[sel >= 0]] = True
        return np.sum(mat[mask, :][:, mask])

    def ewald_delta(self, occ, flips, legacy=False):
        ew, lg = self.core[2], self.core[3]
        inds, mat = self.ewald
        occu_i = np.array(occ, dtype=np.int32)
        out = 0.0
        for site, code in flips:
            occu_f = occu_i.copy()
            occu_f[site] = code
            fn = lg.delta_ewald_single_flip if legacy else ew.delta_ewald_single_flip
            out += fn(occu_f, occu_i, mat, inds, int(site))
            occu_i = occu_f
        return out

    # composite.py:135-157 + ensemble.py:353-376
    def feature_change(self, occ, flips, mode):
        d = self.delta(occ, flips, "corr" if mode == "corr" else "int")
        if self.ewald is not None:
            d = np.append(d, self.ewald_delta(occ, flips))
        if self.mu_table is not None:
            dw = sum(self.mu_table[s][c] - self.mu_table[s][occ[s]] for s, c in flips)
            d = np.append(d, dw)
        return d

    def features(self, occ, mode):
        f = self.corr_full(occ) if mode == "corr" else self.int_full(occ)
        if self.ewald is not None:
            f = np.append(f, self.ewald_full(occ))
        if self.mu_table is not None:
            f = np.append(f, sum(self.mu_table[s][c] for s, c in enumerate(occ)))
        return f

    def natural_params(self, mode):
        p = self.coefs if mode == "corr" else self.model.orbit_multiplicities.astype(float)
        if self.ewald is not None:
            p = np.append(p, self.ewald_coef)
        if self.mu_table is not None:
            p = np.append(p, -1.0)
        return p


# --------------------------------------------------------------------------- #
# restated ushers with the reference's Generator call order
# --------------------------------------------------------------------------- #
class Sub:
    def __init__(self, sites, ncodes):
        self.sites = np.asarray(sites)
        self.active_sites = self.sites.copy()
        self.encoding = np.arange(ncodes, dtype=np.int32)


def propose_flip(rng, subs, probs, occ):  # mcusher.py:154-170
    sub = rng.choice(subs, p=probs)
    site = rng.choice(sub.active_sites)
    choices = set(sub.encoding) - {occ[site]}
    return [(int(site), int(rng.choice(list(choices))))]


def propose_swap(rng, subs, probs, occ):  # mcusher.py:176-200
    sub = rng.choice(subs, p=probs)
    site1 = rng.choice(sub.active_sites)
    species1 = occ[site1]
    sub_occ = occ[sub.active_sites]
    opts = sub.active_sites[sub_occ != species1]
    if opts.size > 0:
        site2 = rng.choice(opts)
        return [(int(site1), int(occ[site2])), (int(site2), int(species1))]
    return []


def run_metropolis(proc, mode, subs, step_type, temperature, occ0, nsteps, seed):
    """kernel/base.py:145-166,291-343 + metropolis.py:31-49 + sampler.py:195-207."""
    rng = np.random.default_rng(seed)
    probs = np.full(len(subs), 1.0 / len(subs))
    prop = propose_flip if step_type == "flip" else propose_swap
    nat = proc.natural_params(mode)
    beta = 1.0 / (kB * temperature)
    # constructor priming step on all-zeros occupancy (kernel/base.py:237-239)
    z = np.zeros(len(occ0), dtype=np.int32)
    st = prop(rng, subs, probs, z)
    dfe = proc.feature_change(z, st, mode)
    ex = -beta * float(np.dot(nat, dfe))
    if not ex >= 0:
        rng.random()
    occ = np.array(occ0, dtype=np.int32)
    feats = proc.features(occ, mode)
    enth = float(np.dot(nat, feats))
    steps = -np.ones((nsteps, 4), dtype=np.int32)
    us = np.full(nsteps, np.nan)
    acc = np.zeros(nsteps, dtype=bool)
    dH = np.zeros(nsteps)
    H = np.zeros(nsteps)
    for k in range(nsteps):
        st = prop(rng, subs, probs, occ)
        for j, (s, c) in enumerate(st):
            steps[k, 2 * j], steps[k, 2 * j + 1] = s, c
        dfe = proc.feature_change(occ, st, mode)
        dh = np.array(np.dot(nat, dfe), dtype=np.float64)
        exponent = -beta * dh
        if exponent >= 0:
            a = True
        else:
            u = rng.random()
            us[k] = u
            a = bool(exponent > log(u))
        if a:
            for s, c in st:
                occ[s] = c
            feats = feats + dfe
            enth = enth + float(dh)
        acc[k], dH[k], H[k] = a, float(dh), enth
    return dict(steps=steps, u=us, accepted=acc, dH=dH, H=H, occ_final=occ, feat_final=feats)


def run_wanglandau(proc, mode, subs, step_type, window, occ0, nsteps, seed,
                   flatness=0.8, mod_factor=1.0, check_period=1000, update_period=1):
    """wanglandau.py:107-300 restated; arithmetic from the reference core."""
    rng = np.random.default_rng(seed)
    probs = np.full(len(subs), 1.0 / len(subs))
    prop = propose_flip if step_type == "flip" else propose_swap
    nat = proc.natural_params(mode)
    emin, emax, bsz = window
    levels = np.arange(emin, emax, bsz)
    L, F = len(levels), len(nat)
    entropy = np.zeros(L)
    hist = np.zeros(L, dtype=np.int64)
    occur = np.zeros(L, dtype=np.int64)
    meanf = np.zeros((L, F))
    m = mod_factor
    # priming step in the constructor: current enthalpy is inf => bin inf, rejected
    # by the window test (inf + dh >= max); draws only the proposal numbers.
    z = np.zeros(len(occ0), dtype=np.int32)
    prop(rng, subs, probs, z)
    counter = 0
    occ = np.array(occ0, dtype=np.int32)
    cur_f = proc.features(occ, mode)
    cur_h = float(np.dot(cur_f, nat))

    def bin_id(e):
        return int((e - emin) // bsz)

    steps = -np.ones((nsteps, 4), dtype=np.int32)
    us = np.full(nsteps, np.nan)
    acc = np.zeros(nsteps, dtype=bool)
    H = np.zeros(nsteps)
    for k in range(nsteps):
        st = prop(rng, subs, probs, occ)
        for j, (s, c) in enumerate(st):
            steps[k, 2 * j], steps[k, 2 * j + 1] = s, c
        dfe = proc.feature_change(occ, st, mode)
        dh = float(np.dot(nat, dfe))
        b = bin_id(cur_h)
        new_h = cur_h + dh
        if new_h < emin or new_h >= emax:
            a = False
        else:
            nb = bin_id(new_h)
            exponent = entropy[b] - entropy[nb] + 0.0
            if exponent >= 0:
                a = True
            else:
                u = rng.random()
                us[k] = u
                a = bool(exponent > log(u))
        if a:
            for s, c in st:
                occ[s] = c
            cur_f = cur_f + dfe
            cur_h = cur_h + dh
        b = bin_id(cur_h)
        if 0 <= b < L:
            counter += 1
            total = occur[b]
            meanf[b, :] = 1 / (total + 1) * (cur_f + total * meanf[b, :])
            if counter % update_period == 0:
                entropy[b] += m
                hist[b] += 1
                occur[b] += 1
        if counter % check_period == 0:
            h = hist[entropy > 0]
            if len(h) >= 2 and (h > flatness * h.mean()).all():
                hist[:] = 0
                m = m / 2.0
        acc[k], H[k] = a, cur_h
    return dict(steps=steps, u=us, accepted=acc, H=H, occ_final=occ, entropy=entropy,
                histogram=hist, occurrences=occur, mean_features=meanf,
                mod_factor=np.array([m]), feat_final=cur_f, levels=levels)


# --------------------------------------------------------------------------- #
def rand_occ(rng, sc):
    prim = sc.model.prim
    nsp = np.array([prim.nspecies[b] for b in sc.site_b])
    return (rng.random(sc.num_sites) * nsp).astype(np.int32)


def model_arrays(model, coefs):
    d = {"coefs": np.asarray(coefs)}
    for o in model.orbits:
        d[f"ct_{o.id}"] = o.flat_correlation_tensors
    for i, t in enumerate(model.cluster_interaction_tensors(coefs)):
        d[f"it_{i}"] = np.ravel(np.asarray(t))
    return d


def golden_case(core, name, prim, cutoffs, scmatrix, basis="sinusoid", with_ewald=False,
                nocc=3, nflips=120, seed=0, write=True):
    from smol_amd import ewald as ewmod
    from smol_amd import synth

    rng = np.random.default_rng(seed)
    model = synth.build_cluster_model(prim, cutoffs, basis=basis)
    sc = synth.build_supercell(model, scmatrix)
    coefs = synth.random_coefs(model, seed=seed + 11)
    coefs[0] = 0.37  # exercise the offset / empty-cluster coefficient
    ew = ewmod.supercell_ewald(sc) if with_ewald else None
    proc = RefProcessor(core, model, sc, coefs, ewald=ew, ewald_coef=0.1)
    active = np.flatnonzero(np.array([prim.nspecies[b] for b in sc.site_b]) > 1)
    out = model_arrays(model, coefs)
    out["scmatrix"] = sc.scmatrix
    occs, fc, fi, fl = [], [], [], []
    flips, dcs, dis, dcl, dil, des, del_, few = [], [], [], [], [], [], [], []
    for _ in range(nocc):
        occ = rand_occ(rng, sc)
        occs.append(occ)
        fc.append(proc.corr_full(occ))
        fi.append(proc.int_full(occ))
        fl.append(proc.corr_full_legacy(occ))
        if with_ewald:
            few.append(proc.ewald_full(occ))
        for _ in range(nflips):
            nf = int(rng.integers(1, 3))
            st = []
            cur = occ.copy()
            for _ in range(nf):
                s = int(rng.choice(active))
                S = prim.nspecies[sc.site_b[s]]
                c = int((cur[s] + 1 + rng.integers(0, S - 1)) % S)
                st.append((s, c))
                cur[s] = c
            row = -np.ones(4, dtype=np.int32)
            for j, (s, c) in enumerate(st):
                row[2 * j], row[2 * j + 1] = s, c
            flips.append(row)
            dcs.append(proc.delta(occ, st, "corr"))
            dis.append(proc.delta(occ, st, "int"))
            dcl.append(proc.delta_legacy(occ, st, "corr"))
            dil.append(proc.delta_legacy(occ, st, "int"))
            if with_ewald:
                des.append(proc.ewald_delta(occ, st))
                del_.append(proc.ewald_delta(occ, st, legacy=True))
    out.update(
        occ=np.array(occs), full_corr=np.array(fc), full_int=np.array(fi),
        full_corr_legacy=np.array(fl), flips=np.array(flips),
        delta_corr=np.array(dcs), delta_int=np.array(dis),
        delta_corr_legacy=np.array(dcl), delta_int_legacy=np.array(dil),
    )
    if with_ewald:
        out.update(full_ewald=np.array(few), delta_ewald=np.array(des),
                   delta_ewald_legacy=np.array(del_),
                   ewald_diag=np.diag(ew[1]).copy(), ewald_row0=ew[1][0].copy())
    if write:  # (make_golden_v6.py rebuilds the same models without rewriting the fixtures)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "sites", sc.num_sites, "F", model.num_corr_functions, "flips", len(flips))
    return model, sc, coefs, proc, active


def main():
    from smol_amd import synth

    core = build_reference_core()
    # case A: config-1 shape (FCC conventional 4-site cell, 4x4x4 = 256 sites, pairs only)
    mA = golden_case(core, "fcc_conv444_pairs", synth.fcc_conventional_prim(), {2: 6.0},
                     [4, 4, 4], seed=1)
    # case B: config-2 shape at reduced size (FCC primitive 6^3, pairs + triplets)
    mB = golden_case(core, "fcc_prim666_triplets", synth.fcc_prim(), {2: 6.0, 3: 5.0},
                     [6, 6, 6], seed=2)
    # case C: config-3 shape at reduced size (ternary rocksalt 4^3 + Ewald), sinusoid basis
    mC = golden_case(core, "rocksalt444_ewald", synth.rocksalt_prim(), {2: 6.0, 3: 5.0},
                     [4, 4, 4], with_ewald=True, seed=3)
    # case D: indicator basis, ternary FCC, non-diagonal supercell matrix
    mD = golden_case(core, "fcc3_indicator_skew", synth.fcc_prim(nspecies=3), {2: 5.0, 3: 3.0},
                     [[3, 0, 0], [1, 4, 0], [0, 1, 5]], basis="indicator", seed=4)
    # case E: aliased tiny supercell (duplicate sites inside cluster rows are kept,
    # smol/cofe/space/clusterspace.py:1353-1359)
    mE = golden_case(core, "fcc_prim222_aliased", synth.fcc_prim(), {2: 6.0, 3: 5.0},
                     [2, 2, 2], nflips=40, seed=5)

    # case F: vacancies on the cation site (Ewald index -1 paths of ewald.pyx:46-57)
    mF = golden_case(core, "rocksalt333_vacancy_ewald",
                     synth.rocksalt_prim(cation_charges=(1.0, 3.0, None)), {2: 6.0, 3: 4.5},
                     [3, 3, 3], with_ewald=True, seed=6)
    # case G: two ACTIVE sublattices (cations Li/Mn/Ti + anions O/F): mixed site spaces in
    # one orbit, two site classes, sublattice choice in the ushers
    mG = golden_case(core, "rocksalt333_two_sublattices",
                     synth.rocksalt_prim(anion_charges=(-2.0, -1.0)), {2: 4.5, 3: 3.2},
                     [3, 3, 3], with_ewald=True, seed=7)

    # trajectories (replay mode)
    from smol_amd import ewald as ewmod

    traj = {}
    model, sc, coefs, proc, active = mB
    subs = [Sub(active, 2)]
    rng = np.random.default_rng(77)
    occ0 = np.zeros(sc.num_sites, dtype=np.int32)
    occ0[rng.permutation(sc.num_sites)[: sc.num_sites // 2]] = 1
    for mode in ("int", "corr"):
        r = run_metropolis(proc, mode, subs, "swap", 900.0, occ0, 3000, seed=1234)
        traj.update({f"B_swap_{mode}_{k}": v for k, v in r.items()})
    traj["B_occ0"] = occ0
    traj["B_T"] = np.array([900.0])
    # WL on case B (canonical swap)
    r0 = run_metropolis(proc, "int", subs, "swap", 2000.0, occ0, 500, seed=5)
    hs = r0["H"]
    window = (float(hs.min() - 2.0), float(hs.max() + 2.0), 0.25)
    r = run_wanglandau(proc, "int", subs, "swap", window, occ0, 4000, seed=4321,
                       check_period=200)
    traj.update({f"B_wl_{k}": v for k, v in r.items()})
    traj["B_wl_window"] = np.array(window)
    traj["B_wl_check"] = np.array([200])
    # coarse bins + short check period so the flatness branch (wanglandau.py:253-264) fires
    window2 = (float(hs.min() - 1.0), float(hs.max() + 3.0), 2.0)
    r = run_wanglandau(proc, "int", subs, "swap", window2, occ0, 4000, seed=4322,
                       check_period=50)
    traj.update({f"B_wlflat_{k}": v for k, v in r.items()})
    traj["B_wlflat_window"] = np.array(window2)
    traj["B_wlflat_check"] = np.array([50])
    # semigrand flip with mu + Ewald on case C
    model, sc, coefs, proc, active = mC
    mu = np.zeros((sc.num_sites, 3))
    mu[active] = np.random.default_rng(7).uniform(-0.5, 0.5, 3)[None, :]
    proc.mu_table = mu
    subs = [Sub(active, 3)]
    occ0 = rand_occ(np.random.default_rng(9), sc)
    for mode in ("int", "corr"):
        r = run_metropolis(proc, mode, subs, "flip", 1500.0, occ0, 2000, seed=99)
        traj.update({f"C_flip_{mode}_{k}": v for k, v in r.items()})
    r = run_metropolis(proc, "int", subs, "swap", 1500.0, occ0, 1500, seed=98)
    traj.update({f"C_swap_int_{k}": v for k, v in r.items()})
    traj["C_occ0"], traj["C_mu"], traj["C_T"] = occ0, mu, np.array([1500.0])
    # two active sublattices (case G): swap and semigrand flip
    model, sc, coefs, proc, active = mG
    nsp = np.array([model.prim.nspecies[b] for b in sc.site_b])
    subs = [Sub(np.flatnonzero(sc.site_b == 0), 3), Sub(np.flatnonzero(sc.site_b == 1), 2)]
    occ0 = rand_occ(np.random.default_rng(19), sc)
    mu = np.zeros((sc.num_sites, 3))
    mu[sc.site_b == 0] = [0.2, -0.1, 0.05]
    mu[sc.site_b == 1, :2] = [-0.3, 0.15]
    r = run_metropolis(proc, "int", subs, "swap", 2500.0, occ0, 1500, seed=71)
    traj.update({f"G_swap_int_{k}": v for k, v in r.items()})
    proc.mu_table = mu
    r = run_metropolis(proc, "corr", subs, "flip", 2500.0, occ0, 1500, seed=72)
    traj.update({f"G_flip_corr_{k}": v for k, v in r.items()})
    traj["G_occ0"], traj["G_mu"], traj["G_T"] = occ0, mu, np.array([2500.0])
    np.savez_compressed(os.path.join(HERE, "trajectories.npz"), **traj)
    print("trajectories written")


if __name__ == "__main__":
    main()
