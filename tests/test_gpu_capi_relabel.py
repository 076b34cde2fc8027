"""A plain C-ABI client (ctypes only -- no smol_amd.engine.Engine, nothing translated in Python) on a model with
RESTRICTED sites: the restricted-sites model of profiles/r04_restricted.jsonl (tools/bench_restricted.py: the config-3
lattice with 10 % of the cations frozen, smol/moca/sublattice.py:84-107) at a test size.  Until ABI 8 such a client got
"the active sites of a sublattice are not one site range" and mc_kernel; smolmc_create now renumbers the sites itself
and every entry point speaks the CALLER's numbering: set_state / run / get_state / run_sampled + get_samples (int32 and
u8) / replay / eval_full / eval_delta are compared with the CPU oracle on the same tables."""

import ctypes as C

import numpy as np
import pytest

from smol_amd import capi, engine, moca, synth

pytestmark = pytest.mark.gpu

_i32, _f64, _u64, _u8 = C.c_int32, C.c_double, C.c_uint64, C.c_uint8


def P(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _model(step, dim=6):
    model = synth.build_cluster_model(synth.rocksalt_prim(), {2: 6.0, 3: 4.5})
    sc = synth.build_supercell(model, [dim] * 3)
    ens = moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model), ewald_coefficient=0.1)
    rng = np.random.default_rng(0)
    cations = ens.sublattices[0]
    frozen = rng.choice(cations.sites, len(cations.sites) // 10, replace=False)
    ens.restrict_sites(frozen)
    if step == capi.STEP_FLIP:
        ens.chemical_potentials = {sp: 0.03 * i for i, sp in enumerate(ens.species)}
    Pn = sc.size
    base = np.zeros(Pn, dtype=np.int32)  # charge neutral: n_Li + 3 n_Mn + 4 n_Ti = 2 P
    n_ti = Pn // 6
    n_mn = (Pn - 3 * n_ti) // 2
    base[:n_mn] = 1
    base[n_mn:n_mn + n_ti] = 2
    R = 6
    occ = np.zeros((R, sc.num_sites), dtype=np.int32)
    for w in range(R):
        occ[w, :Pn] = rng.permutation(base)
    kw = dict(flip_table=[[1, -3, 2, 0]], swap_weight=0.1) if step == capi.STEP_TABLE_FLIP else {}
    return sc, ens, ens.make_tables(**kw), occ, np.sort(frozen)


@pytest.mark.parametrize("step", [capi.STEP_SWAP, capi.STEP_FLIP, capi.STEP_TABLE_FLIP], ids=["swap", "flip", "table-flip"])
def test_c_client_with_restricted_sites_runs_lean_in_its_own_numbering(step, monkeypatch):
    from oracle import oracle as orc

    monkeypatch.delenv("SMOLMC_NO_SITE_RELABEL", raising=False)
    lib = engine.load_library()  # (a ctypes.CDLL with argtypes set: the C-ABI and nothing else)
    sc, ens, tab, occ, frozen = _model(step)
    R, N = occ.shape
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, step)
    h = C.c_void_p()
    assert lib.smolmc_create(C.byref(tab.struct), C.byref(cfg), C.byref(h)) == 0, lib.smolmc_last_error()
    try:
        buf = C.create_string_buffer(512)
        assert lib.smolmc_kernel_info(h, buf, 512) == 0
        info = buf.value.decode()
        assert info.startswith("lean") and "relabelled=1" in info, info
        F = lib.smolmc_num_features(h)
        seeds = np.arange(R, dtype=np.uint64) + np.uint64(9)
        T = np.full(R, 4000.0)
        assert lib.smolmc_set_state(h, P(occ, _i32), P(seeds, _u64), P(T, _f64), 1) == 0, lib.smolmc_last_error()
        ora = orc.OracleMC(tab, cfg)
        ora.set_state(occ, seeds, 4000.0)

        def state():
            o = np.empty((R, N), dtype=np.int32)
            f, H = np.empty((R, F)), np.empty(R)
            na, ns = np.empty(R, dtype=np.uint64), np.empty(R, dtype=np.uint64)
            assert lib.smolmc_get_state(h, P(o, _i32), P(f, _f64), P(H, _f64), P(na, _u64), P(ns, _u64), None) == 0
            return o, f, H, na

        def same():
            o, f, H, na = state()
            so = ora.get_state()
            assert np.array_equal(o, so["occupancy"])  # the CALLER's numbering
            assert np.array_equal(na, so["n_accepted"])
            np.testing.assert_allclose(H, so["enthalpy"], rtol=1e-10, atol=1e-8)
            np.testing.assert_allclose(f, so["features"], rtol=1e-10, atol=1e-8)
            return o

        same()
        for n in (1, 37, 400):
            assert lib.smolmc_run(h, n) == 0
            ora.run(n)
            o = same()
        assert np.all(o[:, frozen] == occ[:, frozen]) and np.any(o != occ)
        # the device ring: int32 and u8 occupancy rows in the caller's numbering
        for u8 in (False, True):
            ns_, thin = 3, 25
            assert lib.smolmc_run_sampled(h, ns_, thin, capi.SAMPLE_OCCUPANCY) == 0, lib.smolmc_last_error()
            H = np.empty((ns_, R))
            rows = np.empty((ns_, R, N), dtype=np.uint8 if u8 else np.int32)
            if u8:
                assert lib.smolmc_get_samples_u8(h, P(H, _f64), None, None, P(rows, _u8)) == 0
            else:
                assert lib.smolmc_get_samples(h, P(H, _f64), None, None, P(rows, _i32)) == 0
            for j in range(ns_):
                ora.run(thin)
                so = ora.get_state()
                assert np.array_equal(rows[j], so["occupancy"])
                np.testing.assert_allclose(H[j], so["enthalpy"], rtol=1e-10, atol=1e-8)
        # evaluator level: full vectors and deltas of steps given in the caller's numbering
        ev = orc.OracleEvaluator(tab)
        o = same()
        full = np.empty((R, F))
        assert lib.smolmc_eval_full(h, P(o, _i32), R, P(full, _f64)) == 0
        np.testing.assert_allclose(full, [ev.feature_vector(x) for x in o], rtol=1e-10, atol=1e-8)
        active = np.asarray(ens.active_sublattices[0].active_sites)
        rng = np.random.default_rng(5)
        recs = np.full((8, capi.STEP_ROW), -1, dtype=np.int32)
        want = []
        for i in range(8):
            s1, s2 = rng.choice(active, 2, replace=False)
            fl = [(int(s1), int((o[0, s1] + 1) % 3)), (int(s2), int((o[0, s2] + 2) % 3))]
            recs[i, :4] = [fl[0][0], fl[0][1], fl[1][0], fl[1][1]]
            want.append(ev.feature_vector_change(o[0], fl))
        d = np.empty((8, F))
        assert lib.smolmc_eval_delta(h, P(o[0].copy(), _i32), P(recs, _i32), 8, P(d, _f64)) == 0, lib.smolmc_last_error()
        np.testing.assert_allclose(d, want, rtol=1e-10, atol=1e-8)
        # a restricted site is not a changeable site: the replay refuses it by ITS (the caller's) index
        if step == capi.STEP_FLIP:
            nrep = 40
            steps = np.full((R, nrep, capi.STEP_ROW), -1, dtype=np.int32)
            so = ora.get_state()["occupancy"]
            for r in range(R):
                for k, s1 in enumerate(rng.choice(active, nrep, replace=False)):  # (distinct sites: code != current)
                    steps[r, k, :2] = [int(s1), (so[r, s1] + 1 + rng.integers(2)) % 3]
            us = rng.random((R, nrep))
            acc = np.zeros((R, nrep), dtype=np.uint8)
            Hs = np.zeros((R, nrep))
            assert lib.smolmc_replay(h, nrep, P(steps, _i32), P(us, _f64), None, P(acc, _u8), P(Hs, _f64), None) == 0, lib.smolmc_last_error()
            acc_o, H_o = ora.replay(steps, us)
            assert np.array_equal(acc.astype(bool), acc_o)
            np.testing.assert_allclose(Hs, H_o, rtol=1e-10, atol=1e-8)
            same()
            bad = steps.copy()
            bad[0, 0, 0] = int(frozen[0])
            assert lib.smolmc_replay(h, nrep, P(bad, _i32), P(us, _f64), None, P(acc, _u8), P(Hs, _f64), None) != 0
            assert b"not changeable" in lib.smolmc_last_error()
        # an out-of-range code is reported on the caller's site
        wrong = occ.copy()
        wrong[1, int(active[3])] = 7
        assert lib.smolmc_set_state(h, P(wrong, _i32), P(seeds, _u64), P(T, _f64), 1) != 0
        assert f"on site {int(active[3])} ".encode() in lib.smolmc_last_error(), lib.smolmc_last_error()
    finally:
        lib.smolmc_destroy(h)


def test_the_switch_puts_the_model_back_on_mc_kernel(monkeypatch):
    """SMOLMC_NO_SITE_RELABEL (A/B): without the renumbering the same tables run on mc_kernel, same chain."""
    from oracle import oracle as orc
    from smol_amd.engine import Engine

    sc, ens, tab, occ, frozen = _model(capi.STEP_SWAP)
    R = len(occ)
    cfg = capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP)
    monkeypatch.setenv("SMOLMC_NO_SITE_RELABEL", "1")
    eng = Engine(tab, cfg)
    assert eng.kernel_info().startswith("general") and "relabelled" not in eng.kernel_info(), eng.kernel_info()
    assert "not one site range" in eng.kernel_info()
    ora = orc.OracleMC(tab, cfg)
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(9)
    eng.set_state(occ, seeds, 4000.0)
    ora.set_state(occ, seeds, 4000.0)
    eng.run(300)
    ora.run(300)
    assert np.array_equal(eng.get_state()["occupancy"], ora.get_state()["occupancy"])
    eng.close()
