"""ctypes binding of the MI355X engine (libsmolmc_hip.so, C-ABI in include/smolmc.h).

There is NO CPU fallback: if the HIP library is missing, or no AMD GPU is visible,
every entry point raises.  (The CPU oracle under oracle/ is test infrastructure and
is never imported from this package.)
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsmolmc_hip.so")
_LIB = None

_i32p, _f64p = C.POINTER(C.c_int32), C.POINTER(C.c_double)
_u64p, _u8p, _i64p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.POINTER(C.c_int64)

# every symbol include/smolmc.h declares: (name, restype, argtypes)
_HP = C.c_void_p
SYMBOLS = {
    "smolmc_create": (C.c_int, [C.POINTER(capi.smolmc_tables), C.POINTER(capi.smolmc_config), C.POINTER(_HP)]),
    "smolmc_destroy": (C.c_int, [_HP]),
    "smolmc_last_error": (C.c_char_p, []),
    "smolmc_abi_version": (C.c_int, []),
    "smolmc_num_features": (C.c_int, [_HP]),
    "smolmc_wl_num_levels": (C.c_int, [_HP]),
    "smolmc_natural_parameters": (C.c_int, [_HP, _f64p]),
    "smolmc_set_state": (C.c_int, [_HP, _i32p, _u64p, _f64p, C.c_int]),
    "smolmc_set_temperature": (C.c_int, [_HP, _f64p]),
    "smolmc_get_state": (C.c_int, [_HP, _i32p, _f64p, _f64p, _u64p, _u64p, _u8p]),
    "smolmc_get_wl": (C.c_int, [_HP, _f64p, _i64p, _i64p, _f64p, _f64p]),
    "smolmc_set_wl": (C.c_int, [_HP, _f64p, _i64p, _i64p, _f64p, _f64p]),
    "smolmc_set_counters": (C.c_int, [_HP, _u64p, _u64p]),
    "smolmc_get_bias": (C.c_int, [_HP, _f64p]),
    "smolmc_kernel_info": (C.c_int, [_HP, C.c_char_p, C.c_int]),
    "smolmc_run": (C.c_int, [_HP, C.c_int64]),
    "smolmc_sync": (C.c_int, [_HP]),
    "smolmc_run_sampled": (C.c_int, [_HP, C.c_int64, C.c_int64, C.c_int]),
    "smolmc_pending_samples": (C.c_int, [_HP, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "smolmc_discard_samples": (C.c_int, [_HP]),
    "smolmc_get_samples": (C.c_int, [_HP, _f64p, _f64p, _u8p, _i32p]),
    "smolmc_get_samples_u8": (C.c_int, [_HP, _f64p, _f64p, _u8p, _u8p]),
    "smolmc_get_samples_ex": (C.c_int, [_HP, _f64p, _f64p, _u8p, _u8p, _f64p, _f64p, _i64p, _i64p, _f64p, _f64p]),
    "smolmc_replay": (C.c_int, [_HP, C.c_int64, _i32p, _f64p, _f64p, _u8p, _f64p, _f64p]),
    "smolmc_last_kernel_ms": (C.c_int, [_HP, C.POINTER(C.c_float)]),
    "smolmc_eval_full": (C.c_int, [_HP, _i32p, C.c_int, _f64p]),
    "smolmc_eval_delta": (C.c_int, [_HP, _i32p, _i32p, C.c_int, _f64p]),
    "smolmc_set_stream": (C.c_int, [_HP, C.c_void_p]),
    "smolmc_export_enthalpy_dev": (C.c_int, [_HP, C.c_void_p]),
    "smolmc_import_temperature_dev": (C.c_int, [_HP, C.c_void_p]),
    "smolmc_exchange_dev": (C.c_int, [_HP, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}


def source_digest(root=None):
    """sha256 over the kernel sources (smol_amd/csrc/*.h, *.hip, Makefile and include/smolmc.h, by
    sorted name) of the repository at ``root`` (default: this one).  profiles/pmc_constants.json
    carries the digest of the tree its counters were collected on, next to the per-kernel machine-code
    digest (codeobj.py) that bench.py compares: this one names the tree, that one says whether the
    kernel an entry was measured on is still the kernel that runs."""
    import glob
    import hashlib

    h = hashlib.sha256()
    here = _HERE if root is None else os.path.join(root, "smol_amd")
    src = os.path.join(here, "csrc")
    files = sorted(glob.glob(os.path.join(src, "*.h")) + glob.glob(os.path.join(src, "*.hip")))
    files += [os.path.join(src, "Makefile"), os.path.join(os.path.dirname(here), "include", "smolmc.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def load_library(path=None):
    """Load libsmolmc_hip.so; raises RuntimeError when it has not been built."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or os.environ.get("SMOLMC_LIB") or LIB_PATH  # SMOLMC_LIB: A/B builds of the kernels
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: build the HIP engine first (python -c 'import __graft_entry__ as g; "
            "g.build()' or make -C smol_amd/csrc). smol_amd has no CPU fallback."
        )
    # One HIP runtime per process.  PyTorch's wheel bundles a libamdhip64 with the same soname as
    # /opt/rocm's; whichever is mapped first serves both, and torch finds "No HIP GPUs" when the
    # system copy got in before it (engine created first, torch imported later for a collective
    # or a stream).  So when torch is installed it is imported before the engine library.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    if lib.smolmc_abi_version() != capi.ABI_VERSION:
        raise RuntimeError("smolmc ABI version mismatch")
    if path is None:
        _LIB = lib
    return lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


class EngineError(RuntimeError):
    pass


class RingFullError(EngineError):
    """smolmc_run_sampled refused a block: both ring slots hold blocks that were not fetched."""


class Engine:
    """One engine handle = R walkers of one ensemble on one GPU."""

    def __init__(self, tables: capi.TableSet, config: capi.smolmc_config):
        self._lib = load_library()
        self.tables, self.config = tables, config
        self._h = _HP()
        rc = self._lib.smolmc_create(C.byref(tables.struct), C.byref(config), C.byref(self._h))
        if rc:
            self._h = None
            self._chk(rc)
        # (scattered active sites -- restricted sites, sublattices split by species -- are renumbered INSIDE
        # smolmc_create since ABI 8, `kernel_info` says "relabelled=1": occupancies and step records cross the C-ABI
        # in the caller's numbering, nothing is translated here)
        self.R = config.n_replicas
        self.N = tables.struct.num_sites
        self.F = self._lib.smolmc_num_features(self._h)
        self.L = self._lib.smolmc_wl_num_levels(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.smolmc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            msg = self._lib.smolmc_last_error().decode()
            if rc == capi.ERR_RING_FULL:
                raise RingFullError(msg)
            # argument problems are ValueErrors like the reference's (expansion.py:97-103,
            # wanglandau.py:80-88); device problems are RuntimeErrors
            if any(k in msg for k in ("enthalpy", "mod_factor", "range", "must be", "larger")):
                raise ValueError(msg)
            raise EngineError(msg)

    @staticmethod
    def _occ32(occ, shape):
        occ = np.asarray(occ)
        if occ.dtype != np.int32:
            if not np.issubdtype(occ.dtype, np.integer):
                raise ValueError(
                    f"occupancy dtype is: {occ.dtype}, but should be integers!"
                )  # expansion.py:225-228
            occ = occ.astype(np.int32)
        return np.ascontiguousarray(occ).reshape(shape)

    # ---- state ----------------------------------------------------------------
    @property
    def natural_parameters(self):
        out = np.zeros(self.F)
        self._chk(self._lib.smolmc_natural_parameters(self._h, _p(out, C.c_double)))
        return out

    def set_state(self, occupancies, seeds=None, temperature=None, reset_aux=True):
        occ = self._occ32(occupancies, (self.R, self.N))
        seeds = (
            np.arange(self.R, dtype=np.uint64)
            if seeds is None
            else np.ascontiguousarray(seeds, dtype=np.uint64)
        )
        if len(seeds) != self.R:
            raise ValueError("Number of seeds does not match number of kernels!")  # sampler.py:107
        temp = np.ascontiguousarray(
            np.broadcast_to(np.asarray(1.0 if temperature is None else temperature, float), (self.R,))
        )
        self._chk(
            self._lib.smolmc_set_state(
                self._h, _p(occ, C.c_int32), _p(seeds, C.c_uint64), _p(temp, C.c_double), int(reset_aux)
            )
        )

    def set_temperature(self, temperature):
        temp = np.ascontiguousarray(np.broadcast_to(np.asarray(temperature, float), (self.R,)))
        self._chk(self._lib.smolmc_set_temperature(self._h, _p(temp, C.c_double)))

    def get_state(self, occupancy=True):
        occ = np.empty((self.R, self.N), dtype=np.int32) if occupancy else None
        feat = np.empty((self.R, self.F))
        H = np.empty(self.R)
        na = np.empty(self.R, dtype=np.uint64)
        ns = np.empty(self.R, dtype=np.uint64)
        la = np.empty(self.R, dtype=np.uint8)
        self._chk(
            self._lib.smolmc_get_state(
                self._h, _p(occ, C.c_int32), _p(feat, C.c_double), _p(H, C.c_double),
                _p(na, C.c_uint64), _p(ns, C.c_uint64), _p(la, C.c_uint8),
            )
        )
        return dict(occupancy=occ, features=feat, enthalpy=H, n_accepted=na, n_steps=ns,
                    accepted=la.astype(bool))

    def get_enthalpy(self):
        """Current enthalpy of every walker (one device-to-host copy; the replica-exchange loop
        needs nothing else of the state)."""
        H = np.zeros(self.R)
        self._chk(self._lib.smolmc_get_state(self._h, None, None, _p(H, C.c_double), None, None, None))
        return H

    def get_wl(self):
        # (np.empty: every array is overwritten by the copy; zero-filling 40 MB per call showed in the
        # Sampler's per-sample Wang-Landau trace)
        S = np.empty((self.R, self.L))
        hist = np.empty((self.R, self.L), dtype=np.int64)
        occ = np.empty((self.R, self.L), dtype=np.int64)
        mf = np.empty((self.R, self.L, self.F))
        m = np.empty(self.R)
        self._chk(
            self._lib.smolmc_get_wl(
                self._h, _p(S, C.c_double), _p(hist, C.c_int64), _p(occ, C.c_int64),
                _p(mf, C.c_double), _p(m, C.c_double),
            )
        )
        return dict(entropy=S, histogram=hist, occurrences=occ, mean_features=mf, mod_factor=m)

    def set_wl(self, entropy=None, histogram=None, occurrences=None, mean_features=None, mod_factor=None):
        """Upload Wang-Landau aux arrays (the inverse of get_wl; None = keep what the device has)."""
        def arr(x, dt, shape):
            if x is None:
                return None
            x = np.ascontiguousarray(x, dtype=dt)
            if x.shape != shape:
                raise ValueError(f"expected an array of shape {shape}, got {x.shape}")
            return x

        RL = (self.R, self.L)
        S, hist, occ = arr(entropy, np.float64, RL), arr(histogram, np.int64, RL), arr(occurrences, np.int64, RL)
        mf, m = arr(mean_features, np.float64, RL + (self.F,)), arr(mod_factor, np.float64, (self.R,))
        self._chk(self._lib.smolmc_set_wl(self._h, _p(S, C.c_double), _p(hist, C.c_int64), _p(occ, C.c_int64),
                                          _p(mf, C.c_double), _p(m, C.c_double)))

    def set_counters(self, n_steps=None, n_accepted=None):
        """Step / accept counters of every walker (n_steps is the position in the random stream)."""
        ns = None if n_steps is None else np.ascontiguousarray(n_steps, dtype=np.uint64)
        na = None if n_accepted is None else np.ascontiguousarray(n_accepted, dtype=np.uint64)
        for x in (ns, na):
            if x is not None and x.shape != (self.R,):
                raise ValueError(f"expected {self.R} counters")
        self._chk(self._lib.smolmc_set_counters(self._h, _p(ns, C.c_uint64), _p(na, C.c_uint64)))

    def audit_drift(self):
        """Drift of the running trace against a from-scratch evaluation of the current
        occupancies (the batched full-vector kernel, evaluator.pyx:121-209): returns
        (max |features - recomputed|, max |enthalpy - natural_parameters . recomputed|).
        The reference bounds the analogous per-flip drift in tests/test_moca/test_processor.py
        :170-172; here it is the accumulated value after any number of steps."""
        st = self.get_state()
        full = self.eval_full(st["occupancy"])
        df = float(np.max(np.abs(st["features"] - full)))
        dh = float(np.max(np.abs(st["enthalpy"] - full @ self.natural_parameters)))
        return df, dh

    def kernel_info(self):
        """Kernel family this handle dispatches to, e.g. 'lean nslot=2 mm=2 field=0 lds=19968'; for a model on
        ``mc_kernel`` / the universal kernel the string ends with ' | not lean: <the first condition that failed>'."""
        buf = C.create_string_buffer(512)
        self._chk(self._lib.smolmc_kernel_info(self._h, buf, 512))
        return buf.value.decode()

    def not_lean_reason(self):
        """Why the model runs on ``mc_kernel`` / the universal kernel instead of a lean family: the first condition that
        failed at ``smolmc_create`` (None on a lean handle)."""
        info = self.kernel_info()
        return info.split(" | not lean: ", 1)[1] if " | not lean: " in info else None

    def get_bias(self):
        """trace.bias of every walker (models created with an MCBias term)."""
        b = np.zeros(self.R)
        self._chk(self._lib.smolmc_get_bias(self._h, _p(b, C.c_double)))
        return b

    # ---- hot path -------------------------------------------------------------
    def run(self, nsteps, sync=False):
        self._chk(self._lib.smolmc_run(self._h, int(nsteps)))
        if sync:
            self.sync()

    def sync(self):
        self._chk(self._lib.smolmc_sync(self._h))

    def run_sampled(self, nsamples, thin_by, occupancy=True, packed=False, bias=False, wl=False):
        """Advance nsamples*thin_by steps recording one sample per walker every thin_by steps
        on the device; returns dict(enthalpy (ns,R), features (ns,R,F), accepted (ns,R) bool,
        occupancy (ns,R,N) or None[, bias (ns,R)][, the Wang-Landau trace]).  Occupancies come back as
        int32 (the reference's trace dtype) or, with ``packed``, as the ring's own uint8 -- a quarter
        of the transfer.  = ``run_sampled_async`` + ``fetch_samples``."""
        self.run_sampled_async(nsamples, thin_by, occupancy=occupancy, bias=bias, wl=wl)
        return self.fetch_samples(packed=packed)

    def run_sampled_async(self, nsamples, thin_by, occupancy=True, bias=False, wl=False):
        """Queue one block of the device ring (smolmc_run_sampled): returns at once.  The ring has two
        slots; queue block k + 1 BEFORE fetching block k and the download of k overlaps the kernel of
        k + 1 (the order ``fetch_samples`` delivers in: oldest first)."""
        flags = ((capi.SAMPLE_OCCUPANCY if occupancy else 0) | (capi.SAMPLE_BIAS if bias else 0) |
                 (capi.SAMPLE_WL if wl else 0))
        self._chk(self._lib.smolmc_run_sampled(self._h, int(nsamples), int(thin_by), flags))

    def pending_samples(self):
        """(blocks queued and not fetched, nsamples and flags of the block ``fetch_samples`` would deliver): the
        ring's own bookkeeping (smolmc_pending_samples) -- the arrays of a fetch are sized from it."""
        npend, ns, flags = C.c_int(), C.c_int64(), C.c_int()
        self._chk(self._lib.smolmc_pending_samples(self._h, C.byref(npend), C.byref(ns), C.byref(flags)))
        return int(npend.value), int(ns.value), int(flags.value)

    def discard_samples(self):
        """Forget every block of the ring (a sampling loop left half way must not hand its blocks to the next)."""
        self._chk(self._lib.smolmc_discard_samples(self._h))

    def fetch_samples(self, packed=False):
        """The oldest block queued with ``run_sampled_async`` that was not fetched yet (waits for ITS
        download only)."""
        npend, ns, flags = self.pending_samples()
        if npend == 0:
            raise EngineError("no samples recorded: call run_sampled_async first")
        H = np.empty((ns, self.R))
        feat = np.empty((ns, self.R, self.F))
        acc = np.empty((ns, self.R), dtype=np.uint8)
        with_occ = bool(flags & capi.SAMPLE_OCCUPANCY)
        extra = {}
        if flags & capi.SAMPLE_BIAS:
            extra["bias"] = np.empty((ns, self.R))
        if flags & capi.SAMPLE_WL:
            extra.update(entropy=np.empty((ns, self.R, self.L)), histogram=np.empty((ns, self.R, self.L), dtype=np.int64),
                         occurrences=np.empty((ns, self.R, self.L), dtype=np.int64),
                         mean_features=np.empty((ns, self.R, self.L, self.F)), mod_factor=np.empty((ns, self.R)))
        if packed or extra:
            occ = np.empty((ns, self.R, self.N), dtype=np.uint8) if with_occ else None
            self._chk(self._lib.smolmc_get_samples_ex(
                self._h, _p(H, C.c_double), _p(feat, C.c_double), _p(acc, C.c_uint8), _p(occ, C.c_uint8),
                _p(extra.get("bias"), C.c_double), _p(extra.get("entropy"), C.c_double),
                _p(extra.get("histogram"), C.c_int64), _p(extra.get("occurrences"), C.c_int64),
                _p(extra.get("mean_features"), C.c_double), _p(extra.get("mod_factor"), C.c_double)))
            if occ is not None and not packed:
                occ = occ.astype(np.int32)
        else:
            occ = np.empty((ns, self.R, self.N), dtype=np.int32) if with_occ else None
            self._chk(self._lib.smolmc_get_samples(self._h, _p(H, C.c_double), _p(feat, C.c_double),
                                                   _p(acc, C.c_uint8), _p(occ, C.c_int32)))
        return dict(enthalpy=H, features=feat, accepted=acc.astype(bool), occupancy=occ, **extra)

    def last_kernel_ms(self):
        ms = C.c_float()
        self._chk(self._lib.smolmc_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def replay(self, steps, uniforms, log_priori=None, with_priori=False):
        """steps (R, n, 2k) int32 with k <= 8 (site, code) pairs per record (padded to
        SMOLMC_STEP_ROW with -1), uniforms (R, n) float64, log_priori (R, n) or None (see
        smolmc_replay) -> (accepted (R,n) bool, H (R,n)[, log_priori used (R,n)])."""
        uniforms = np.ascontiguousarray(uniforms, dtype=np.float64).reshape(self.R, -1)
        n = uniforms.shape[1]
        steps = capi.step_rows(steps, self.R, n)
        lp = None if log_priori is None else np.ascontiguousarray(log_priori, dtype=np.float64).reshape(self.R, n)
        acc = np.zeros((self.R, n), dtype=np.uint8)
        H = np.zeros((self.R, n))
        lpo = np.zeros((self.R, n)) if with_priori else None
        self._chk(
            self._lib.smolmc_replay(
                self._h, n, _p(steps, C.c_int32), _p(uniforms, C.c_double), _p(lp, C.c_double),
                _p(acc, C.c_uint8), _p(H, C.c_double), _p(lpo, C.c_double),
            )
        )
        return (acc.astype(bool), H, lpo) if with_priori else (acc.astype(bool), H)

    # ---- evaluator level --------------------------------------------------------
    def eval_full(self, occupancies):
        occ = self._occ32(occupancies, (-1, self.N))
        out = np.zeros((len(occ), self.F))
        self._chk(self._lib.smolmc_eval_full(self._h, _p(occ, C.c_int32), len(occ), _p(out, C.c_double)))
        return out

    def eval_delta(self, occupancy, steps, single_step=None):
        """Feature changes of steps applied to ``occupancy`` (each step on its own, not chained), (n, F).

        ``steps`` as records: an ndarray of (n, 2k) int32 rows (site_0, code_0, ..., site_{k-1}, code_{k-1}),
        k <= 8, -1 = absent -- n steps, exactly what capi.step_rows makes of it; a flat record is one step.
        ``steps`` the way the reference's ushers return ONE step: a Python list of (site, code) tuples
        (mcusher.py:104-116).  An (n, 2) ndarray with n > 1 could be read either way -- n single flips or one step
        of n flips, and it silently changed meaning between rounds -- so it is refused unless ``single_step`` says
        which: True = one step of n flips, False = n steps of one flip.  ``single_step`` also overrides the default
        reading of the other spellings."""
        occ = self._occ32(occupancy, (self.N,))
        as_pairs = isinstance(steps, (list, tuple)) and len(steps) > 0 and all(
            isinstance(f, (list, tuple)) and len(f) == 2 for f in steps)
        a = np.asarray(steps, dtype=np.int32)
        if single_step is None:
            if isinstance(steps, np.ndarray) and a.ndim == 2 and a.shape[1] == 2 and a.shape[0] > 1:
                raise ValueError(
                    f"an ({a.shape[0]}, 2) array is ambiguous: {a.shape[0]} single-flip steps (single_step=False) or "
                    f"one step of {a.shape[0]} flips (single_step=True)?")
            single_step = a.ndim == 1 or as_pairs
        if single_step:
            a = a.reshape(1, -1)  # one step: flat record, a list of (site, code) tuples, or pair rows
        steps = capi.step_rows(a)
        out = np.zeros((len(steps), self.F))
        self._chk(
            self._lib.smolmc_eval_delta(
                self._h, _p(occ, C.c_int32), _p(steps, C.c_int32), len(steps), _p(out, C.c_double)
            )
        )
        return out

    # ---- device plumbing ----------------------------------------------------------
    def set_stream(self, stream_ptr):
        self._chk(self._lib.smolmc_set_stream(self._h, C.c_void_p(int(stream_ptr))))

    def export_enthalpy(self, dev_ptr):
        self._chk(self._lib.smolmc_export_enthalpy_dev(self._h, C.c_void_p(int(dev_ptr))))

    def import_temperature(self, dev_ptr):
        self._chk(self._lib.smolmc_import_temperature_dev(self._h, C.c_void_p(int(dev_ptr))))

    def exchange_dev(self, n_total, first, parity, enthalpy_all_ptr, ladder_ptr, log_u_ptr, rung_of_ptr, stats_ptr=0):
        """One exchange attempt of a temperature ladder decided on the device (smolmc_exchange_dev): device
        pointers in, this handle's temperatures follow the new rung assignment; asynchronous."""
        self._chk(self._lib.smolmc_exchange_dev(
            self._h, int(n_total), int(first), int(parity), C.c_void_p(int(enthalpy_all_ptr)), C.c_void_p(int(ladder_ptr)),
            C.c_void_p(int(log_u_ptr)), C.c_void_p(int(rung_of_ptr)), C.c_void_p(int(stats_ptr)) if stats_ptr else None))
