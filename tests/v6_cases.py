"""Handles for the ABI-v6 golden trajectories (tests/golden/trajectories_v6.npz, made by
tests/golden/make_golden_v6.py from the reference's compiled core with the reference's Generator
call order): one spec per trajectory -- tables, config, initial state -- shared by the oracle tests
(CPU) and the engine tests (GPU)."""

import os

import numpy as np

from smol_amd import capi
from tests.cases import GOLD, load_case

T6 = np.load(os.path.join(GOLD, "trajectories_v6.npz"))
MODES = {"int": capi.FEATURES_INTERACTIONS, "corr": capi.FEATURES_CORRELATIONS}
CASE = {"B": "fcc_prim666_triplets", "C": "rocksalt444_ewald", "G": "rocksalt333_two_sublattices"}


def _hyperplane_tables(A, dim_ids):
    """table_r[site][code] = A[r][dim_id(site, code)], 0 where the code does not exist
    (smolmc.h, SMOLMC_BIAS_SQUARE_HYPERPLANE)."""
    A = np.asarray(A, float)
    out = np.zeros((A.shape[0],) + dim_ids.shape)
    for r in range(A.shape[0]):
        out[r] = np.where(dim_ids >= 0, A[r][np.clip(dim_ids, 0, None)], 0.0)
    return out


# tag -> (case letter, prefix of the shared arrays, mode, step type, kernel, bias)
SPECS = {
    "TC_tf_int": dict(case="C", pre="TC", mode="int", step="table"),
    "TC_tf_corr": dict(case="C", pre="TC", mode="corr", step="table"),
    "TC_tfw_int": dict(case="C", pre="TC", mode="int", step="table", weights="TC_tfw_flip_weights"),
    "TC_tflim_int": dict(case="C", pre="TC", mode="int", step="table", occ0="TC_tflim_occ0", T="TC_tflim_T"),
    "TC_tffug_int": dict(case="C", pre="TC", mode="int", step="table", bias=("fug", "TC_fug_table")),
    "TC_tfwl_int": dict(case="C", pre="TC", mode="int", step="table", wl="TC_tfwl"),
    "TG_tf_int": dict(case="G", pre="TG", mode="int", step="table"),
    "TG_tf_corr": dict(case="G", pre="TG", mode="corr", step="table"),
    "TG6_tf_int": dict(case="G", pre="TG", mode="int", step="table", table="TG6_flip_table"),
    "BC_fug_flip_int": dict(case="C", pre="TC", mode="int", step="flip", T="BC_fug_flip_int_T", bias=("fug", "TC_fug_table")),
    "BC_sqc_flip_corr": dict(case="C", pre="TC", mode="corr", step="flip", T="BC_sqc_flip_corr_T", bias=("sqc", "BC_sqc")),
    "BG_hyp_flip_int": dict(case="G", pre="TG", mode="int", step="flip", T="BG_hyp_flip_int_T", bias=("hyp", "BG_hyp")),
    "BG_sqc_swap_int": dict(case="G", pre="TG", mode="int", step="swap", bias=("sqc", "BG_sqc")),
    "BG_fug_flip_corr": dict(case="G", pre="TG", mode="corr", step="flip", T="BG_fug_flip_corr_T", bias=("fug", "BG_fug_table")),
    "B_wlup3": dict(case="B", pre="B", mode="int", step="swap", wl="B_wlup3", mu=False),
}
STEP = {"flip": capi.STEP_FLIP, "swap": capi.STEP_SWAP, "table": capi.STEP_TABLE_FLIP}


def build(tag, n_replicas=1):
    """-> (tables, config, occ0, temperature) of trajectory `tag`."""
    sp = SPECS[tag]
    c = load_case(CASE[sp["case"]])
    pre = sp["pre"]
    mu = T6[f"{pre}_mu"] if sp.get("mu", True) else None
    kw = {}
    if sp["step"] == "table":
        kw = dict(flip_table=T6[sp.get("table", f"{pre}_flip_table")], swap_weight=float(T6[f"{pre}_swap_weight"][0]),
                  flip_weights=T6[sp["weights"]] if "weights" in sp else None)
    tab = capi.TableSet.from_synth(c["sc"], c["coefs"], feature_mode=MODES[sp["mode"]], ewald=c["ewald"],
                                   ewald_coef=0.1, mu_table=mu, **kw)
    if "bias" in sp:
        kind, key = sp["bias"]
        if kind == "fug":
            tab.set_bias(capi.BIAS_FUGACITY, T6[key])
        elif kind == "sqc":
            tab.set_bias(capi.BIAS_SQUARE_CHARGE, T6[f"{key}_table"], float(T6[f"{key}_penalty"][0]))
        else:
            tab.set_bias(capi.BIAS_SQUARE_HYPERPLANE, _hyperplane_tables(T6[f"{key}_A"], T6[f"{key}_dim_ids"]),
                         float(T6[f"{key}_penalty"][0]), intercepts=T6[f"{key}_b"].astype(float))
    if "wl" in sp:
        w = T6[f"{sp['wl']}_window"]
        up = int(T6[f"{sp['wl']}_update"][0]) if f"{sp['wl']}_update" in T6 else 1
        cfg = capi.make_config(n_replicas, capi.KERNEL_WANGLANDAU, STEP[sp["step"]], min_enthalpy=w[0],
                               max_enthalpy=w[1], bin_size=w[2], check_period=int(T6[f"{sp['wl']}_check"][0]),
                               update_period=up)
        temp = 0.0
    else:
        cfg = capi.make_config(n_replicas, capi.KERNEL_METROPOLIS, STEP[sp["step"]])
        temp = float(T6[sp.get("T", f"{pre}_T")][0])
    if sp["case"] == "B":
        occ0 = np.load(os.path.join(GOLD, "trajectories.npz"))["B_occ0"]
    else:
        occ0 = T6[sp.get("occ0", f"{pre}_occ0")]
    return tab, cfg, occ0, temp


def check_replay(mc, tag, acc, H, lp_out=None, rtol=1e-10, h0=None):
    """Accept flags, final occupancy, counters: bit-exact; enthalpies / features / bias: 1e-10
    relative (north_star's tolerance) with the absolute floor of the stored doubles' rounding."""
    g = lambda k: T6[f"{tag}_{k}"]  # noqa: E731
    assert np.array_equal(acc, g("accepted")), f"{tag}: first differing step {np.flatnonzero(acc != g('accepted'))[:5]}"
    np.testing.assert_allclose(H, g("H"), rtol=rtol, atol=1e-9)
    if f"{tag}_dH" in T6:
        # the enthalpy change of every ACCEPTED step, 1e-10 relative (north_star) -- read off the running
        # enthalpies, so with the rounding of the two stored doubles it is the difference of as the floor
        a = g("accepted").astype(bool)
        h_prev = np.concatenate(([h0 if h0 is not None else np.nan], H[:-1]))
        dH, want = (H - h_prev)[a], g("dH")[a]
        ok = ~np.isnan(dH)
        err = np.abs(dH[ok] - want[ok])
        bound = 1e-10 * np.abs(want[ok]) + 8 * np.finfo(float).eps * np.abs(H[a][ok])
        assert (err <= bound).all(), (tag, float((err / np.maximum(np.abs(want[ok]), 1e-300)).max()))
        # (every reference-order chain sits at an acceptance of 0.15-0.8: the accepted branch -- occupancy, running
        # bias / charge / hyperplane sums, Wang-Landau state -- is exercised hundreds of times, not a few dozen)
        assert a.sum() >= 200 and 0.14 <= a.mean() <= 0.85, (tag, int(a.sum()), float(a.mean()))
    st = mc.get_state()
    assert np.array_equal(st["occupancy"][0], g("occ_final"))
    np.testing.assert_allclose(st["features"][0], g("feat_final"), rtol=rtol, atol=1e-8)
    assert st["n_accepted"][0] == g("accepted").sum()
    if f"{tag}_bias" in T6:
        np.testing.assert_allclose(mc.get_bias()[0], g("bias")[-1], rtol=rtol, atol=1e-9)
    if lp_out is not None:
        ref = g("log_priori")
        ok = ~np.isnan(ref)  # (Wang-Landau never computes it for steps the window rejects)
        np.testing.assert_allclose(lp_out[ok], ref[ok], rtol=1e-10, atol=1e-10)
    return st


def check_wl(mc, tag):
    g = lambda k: T6[f"{tag}_{k}"]  # noqa: E731
    wl = mc.get_wl()
    assert mc.L == len(g("levels"))
    np.testing.assert_allclose(wl["entropy"][0], g("entropy"), rtol=0, atol=0)
    assert np.array_equal(wl["histogram"][0], g("histogram"))
    assert np.array_equal(wl["occurrences"][0], g("occurrences"))
    np.testing.assert_allclose(wl["mean_features"][0], g("mean_features"), rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(wl["mod_factor"][0], g("mod_factor")[0])
