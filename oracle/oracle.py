"""ctypes wrapper of the CPU oracle (oracle/smolmc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under smol_amd/ imports this module.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from smol_amd.capi import smolmc_config, smolmc_tables
from smol_amd.capi import step_rows as capi_step_rows

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_SO_NAME = "libsmolmc_oracle.so"


def use_fast_math_build():
    """Select the second build of the same source with the reference's own flag set
    (-O3 -ffast-math -fopenmp, setup.py:18-25).  Only the cpu_baseline leg of bench.py calls
    this, before the library is first used; the tests keep the IEEE build because replayed
    uniforms carry NaN for "not drawn".  Returns the flags of the build in use."""
    global _SO_NAME
    if _LIB is not None:
        raise RuntimeError("oracle library already loaded")
    _SO_NAME = "libsmolmc_oracle_fast.so"
    return "-O3 -ffast-math -fopenmp"



def build(force=False):
    so = os.path.join(_HERE, _SO_NAME)
    src = os.path.join(_HERE, "smolmc_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "smolmc.h")
    stale = (not os.path.exists(so)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(so) for p in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        i32p, f64p = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        u64p, u8p, i64p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.POINTER(C.c_int64)
        tp = C.POINTER(smolmc_tables)
        L.orc_correlations_from_occupancy.argtypes = [tp, i32p, f64p]
        L.orc_interactions_from_occupancy.argtypes = [tp, i32p, f64p]
        L.orc_delta_correlations.argtypes = [tp, i32p, i32p, C.c_int, f64p]
        L.orc_delta_interactions.argtypes = [tp, i32p, i32p, C.c_int, f64p]
        L.orc_delta_ewald_single_flip.argtypes = [tp, i32p, i32p, C.c_int]
        L.orc_delta_ewald_single_flip.restype = C.c_double
        L.orc_num_features.argtypes = [tp]
        L.orc_natural_parameters.argtypes = [tp, f64p]
        L.orc_feature_vector.argtypes = [tp, i32p, f64p]
        L.orc_feature_vector_change.argtypes = [tp, i32p, i32p, C.c_int, i32p, i32p, f64p]
        L.orc_philox4x32.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.orc_mc_create.argtypes = [tp, C.POINTER(smolmc_config), C.POINTER(C.c_void_p)]
        L.orc_mc_destroy.argtypes = [C.c_void_p]
        L.orc_mc_num_features.argtypes = [C.c_void_p]
        L.orc_mc_wl_num_levels.argtypes = [C.c_void_p]
        L.orc_mc_set_state.argtypes = [C.c_void_p, i32p, u64p, f64p, C.c_int]
        L.orc_mc_set_temperature.argtypes = [C.c_void_p, f64p]
        L.orc_mc_get_state.argtypes = [C.c_void_p, i32p, f64p, f64p, u64p, u64p, u8p]
        L.orc_mc_get_wl.argtypes = [C.c_void_p, f64p, i64p, i64p, f64p, f64p]
        L.orc_mc_get_bias.argtypes = [C.c_void_p, f64p]
        L.orc_compute_bias.restype = C.c_double
        L.orc_compute_bias.argtypes = [tp, i32p]
        L.orc_compute_bias_change.restype = C.c_double
        L.orc_compute_bias_change.argtypes = [tp, i32p, i32p, C.c_int]
        L.orc_mc_run.argtypes = [C.c_void_p, C.c_int64]
        L.orc_mc_replay.argtypes = [C.c_void_p, C.c_int64, i32p, f64p, f64p, u8p, f64p, f64p]
        L.orc_table_step_log_priori.restype = C.c_double
        L.orc_table_step_log_priori.argtypes = [tp, i32p, i32p, C.c_int, C.POINTER(C.c_int)]
        L.orc_mc_propose.argtypes = [C.c_void_p, C.c_int, C.c_uint64, i32p, f64p]
        _LIB = L
    return _LIB


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*[int(x) for x in ctr])
    k = (C.c_uint32 * 2)(*[int(x) for x in key])
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32(c, k, o)
    return [int(x) for x in o]


class OracleEvaluator:
    """Evaluator / processor-level oracle bound to one TableSet."""

    def __init__(self, tables):
        self.tables = tables
        self.t = C.byref(tables.struct)
        self.N = tables.struct.num_sites
        self.F = lib().orc_num_features(self.t)

    def _occ(self, occ):
        occ = np.ascontiguousarray(occ)
        if occ.dtype != np.int32:
            raise ValueError("Buffer dtype mismatch, expected 'const int32_t'")
        return occ

    def correlations(self, occ):
        occ = self._occ(occ)
        out = np.zeros(self.tables.struct.num_corr)
        lib().orc_correlations_from_occupancy(self.t, _p(occ, C.c_int32), _p(out, C.c_double))
        return out

    def interactions(self, occ):
        occ = self._occ(occ)
        out = np.zeros(self.tables.struct.num_orbits)
        lib().orc_interactions_from_occupancy(self.t, _p(occ, C.c_int32), _p(out, C.c_double))
        return out

    def delta_ewald(self, occ_f, occ_i, site):
        return lib().orc_delta_ewald_single_flip(
            self.t, _p(self._occ(occ_f), C.c_int32), _p(self._occ(occ_i), C.c_int32), int(site)
        )

    def feature_vector(self, occ):
        occ = self._occ(occ)
        out = np.zeros(self.F)
        lib().orc_feature_vector(self.t, _p(occ, C.c_int32), _p(out, C.c_double))
        return out

    def feature_vector_change(self, occ, flips):
        """flips: sequence of (site, code)."""
        occ = self._occ(occ)
        fl = np.ascontiguousarray(np.asarray(flips, dtype=np.int32).reshape(-1, 2))
        wf, wi = occ.copy(), occ.copy()
        out = np.zeros(self.F)
        lib().orc_feature_vector_change(
            self.t, _p(occ, C.c_int32), _p(fl, C.c_int32), len(fl),
            _p(wf, C.c_int32), _p(wi, C.c_int32), _p(out, C.c_double),
        )
        return out

    def bias(self, occ):
        """MCBias.compute_bias (bias.py:174-186, :264-277)."""
        return lib().orc_compute_bias(self.t, _p(self._occ(occ), C.c_int32))

    def bias_change(self, occ, flips):
        """MCBias.compute_bias_change (bias.py:75-93, :188-206)."""
        fl = np.ascontiguousarray(np.asarray(flips, dtype=np.int32).reshape(-1, 2))
        return lib().orc_compute_bias_change(self.t, _p(self._occ(occ), C.c_int32), _p(fl, C.c_int32), len(fl))

    def natural_parameters(self):
        out = np.zeros(self.F)
        lib().orc_natural_parameters(self.t, _p(out, C.c_double))
        return out

    def table_log_priori(self, occ, flips):
        """TableFlip.compute_log_priori_factor (mcusher.py:656-711) of a step (sequence of (site, code))."""
        fl = np.ascontiguousarray(np.asarray(flips, dtype=np.int32).reshape(-1, 2))
        status = C.c_int(0)
        v = lib().orc_table_step_log_priori(self.t, _p(self._occ(occ), C.c_int32), _p(fl, C.c_int32), len(fl),
                                            C.byref(status))
        if status.value:
            raise ValueError("Step is not in flip table.")
        return v


class OracleMC:
    """Batched CPU walkers: the oracle twin of smol_amd's engine handle."""

    def __init__(self, tables, config):
        self.tables, self.config = tables, config
        self.h = C.c_void_p()
        rc = lib().orc_mc_create(C.byref(tables.struct), C.byref(config), C.byref(self.h))
        if rc:
            raise RuntimeError("oracle create failed")
        self.R, self.N = config.n_replicas, tables.struct.num_sites
        self.F = lib().orc_mc_num_features(self.h)
        self.L = lib().orc_mc_wl_num_levels(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_mc_destroy(self.h)
            self.h = None

    def set_state(self, occ, seeds, temperature, reset_aux=True):
        occ = np.ascontiguousarray(occ, dtype=np.int32).reshape(self.R, self.N)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        temperature = np.ascontiguousarray(
            np.broadcast_to(np.asarray(temperature, dtype=np.float64), (self.R,))
        )
        lib().orc_mc_set_state(
            self.h, _p(occ, C.c_int32), _p(seeds, C.c_uint64), _p(temperature, C.c_double),
            int(reset_aux),
        )

    def set_counters(self, n_steps=None, n_accepted=None):
        ns = None if n_steps is None else np.ascontiguousarray(n_steps, dtype=np.uint64)
        na = None if n_accepted is None else np.ascontiguousarray(n_accepted, dtype=np.uint64)
        lib().orc_mc_set_counters(self.h, _p(ns, C.c_uint64), _p(na, C.c_uint64))

    def set_temperature(self, temperature):
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(temperature, float), (self.R,)))
        lib().orc_mc_set_temperature(self.h, _p(t, C.c_double))

    def run(self, nsteps):
        lib().orc_mc_run(self.h, int(nsteps))

    def replay(self, steps, uniforms, log_priori=None, with_priori=False):
        """steps (R, n, 2k) int32 with k <= 8 flips per record (padded to SMOLMC_STEP_ROW with -1),
        uniforms (R, n); log_priori (R, n) or None (see smolmc_replay)."""
        uniforms = np.ascontiguousarray(uniforms, dtype=np.float64).reshape(self.R, -1)
        n = uniforms.shape[-1]
        steps = capi_step_rows(steps, self.R, n)
        lp = None if log_priori is None else np.ascontiguousarray(log_priori, dtype=np.float64).reshape(self.R, n)
        acc = np.zeros((self.R, n), dtype=np.uint8)
        H = np.zeros((self.R, n))
        lpo = np.zeros((self.R, n))
        rc = lib().orc_mc_replay(
            self.h, n, _p(steps, C.c_int32), _p(uniforms, C.c_double), _p(lp, C.c_double),
            _p(acc, C.c_uint8), _p(H, C.c_double), _p(lpo, C.c_double),
        )
        if rc:
            raise ValueError("Step is not in flip table.")  # mcusher.py:673-674
        return (acc.astype(bool), H, lpo) if with_priori else (acc.astype(bool), H)

    def get_state(self):
        occ = np.zeros((self.R, self.N), dtype=np.int32)
        feat = np.zeros((self.R, self.F))
        H = np.zeros(self.R)
        na = np.zeros(self.R, dtype=np.uint64)
        ns = np.zeros(self.R, dtype=np.uint64)
        la = np.zeros(self.R, dtype=np.uint8)
        lib().orc_mc_get_state(
            self.h, _p(occ, C.c_int32), _p(feat, C.c_double), _p(H, C.c_double),
            _p(na, C.c_uint64), _p(ns, C.c_uint64), _p(la, C.c_uint8),
        )
        return dict(occupancy=occ, features=feat, enthalpy=H, n_accepted=na, n_steps=ns,
                    accepted=la.astype(bool))

    def get_wl(self):
        S = np.zeros((self.R, self.L))
        hist = np.zeros((self.R, self.L), dtype=np.int64)
        occ = np.zeros((self.R, self.L), dtype=np.int64)
        mf = np.zeros((self.R, self.L, self.F))
        m = np.zeros(self.R)
        lib().orc_mc_get_wl(
            self.h, _p(S, C.c_double), _p(hist, C.c_int64), _p(occ, C.c_int64),
            _p(mf, C.c_double), _p(m, C.c_double),
        )
        return dict(entropy=S, histogram=hist, occurrences=occ, mean_features=mf, mod_factor=m)

    def get_bias(self):
        b = np.zeros(self.R)
        if lib().orc_mc_get_bias(self.h, _p(b, C.c_double)):
            raise RuntimeError("model has no bias term")
        return b

    def propose(self, r, step, with_priori=False):
        fl = np.zeros(16, dtype=np.int32)
        lp = C.c_double(0.0)
        n = lib().orc_mc_propose(self.h, int(r), int(step), _p(fl, C.c_int32), C.byref(lp))
        return (n, fl, lp.value) if with_priori else (n, fl)
