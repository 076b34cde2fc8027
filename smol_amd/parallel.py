"""Multi-GPU layer: one process per GPU, replicas sharded across ranks.

The path shards into independent units: walkers never interact while sampling (the
reference's ``nwalkers`` are independent kernels, smol/moca/sampler/sampler.py:111-116,
:436-440), so there is NO data-path collective.  RCCL (torch.distributed backend "nccl"
on ROCm; "gloo" in the CPU tests) is used only for

  * ``global_sums``  -- all-reduce of O(F) float64 running sums at reporting time;
  * ``ReplicaExchange`` -- the temperature-ladder swap step of BASELINE config 5, which
    has no counterpart in the reference (NEW functionality; validated by invariants,
    not parity): all-gather of one float64 enthalpy per walker, then every rank takes
    the same swap decisions from a shared counter-based random stream and only the
    *temperature assignment* moves -- occupancies never leave their GPU.
"""

from __future__ import annotations

import numpy as np

kB = 8.617333262145e-5  # smol/constants.py:4


def shard(total, rank, world):
    """Contiguous block of walkers owned by ``rank``: (first, count)."""
    base, rem = divmod(int(total), int(world))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


class _NoDist:
    """torch is optional for single-GPU use (engine.load_library treats it the same way)."""

    @staticmethod
    def is_available():
        return False

    @staticmethod
    def is_initialized():
        return False


def _dist():
    try:
        import torch.distributed as dist
    except ImportError:
        return _NoDist
    return dist


def rank_and_world():
    """(rank, world) of the initialised torch.distributed group, else (0, 1)."""
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def collective_device():
    """Where tensors handed to a collective must live: "cuda" under RCCL ("nccl"), else "cpu"."""
    dist = _dist()
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        return "cuda"
    return "cpu"


def local_device(rank):
    """HIP device ordinal of a rank: LOCAL_RANK when the launcher exports it (one process per
    GPU; with per-rank device masking only device 0 is visible), else rank modulo the visible
    devices, else 0."""
    import os

    try:
        import torch

        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        n = 0
    lr = int(os.environ.get("LOCAL_RANK", rank))
    return lr % n if n else 0


def global_sums(local, group=None):
    """Sum a float64 tensor over all ranks (RCCL all-reduce); identity for one rank."""
    dist = _dist()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(local, op=dist.ReduceOp.SUM, group=group)
    return local


def _philox_uniforms(seed, counter, n):
    """n uniforms in [0,1) from NumPy's counter-based Philox (Philox4x64, NOT the engine's
    Philox4x32-10) keyed by ``seed`` at ``counter``: a pure function of (seed, counter), so
    every rank draws the identical numbers without communicating."""
    bitgen = np.random.Philox(key=np.uint64(seed), counter=[0, 0, 0, np.uint64(counter)])
    return np.random.Generator(bitgen).random(n)


class ReplicaExchange:
    """Parallel-tempering bookkeeping over a global ladder of walkers.

    Global walker g lives on rank ``g // per_rank`` at local slot ``g % per_rank`` (equal
    shards).  ``ladder`` holds the temperatures of the rungs; ``rung_of[g]`` is the rung
    walker g currently samples at.  One ``exchange`` attempts swaps between walkers on
    neighbouring rungs (even pairs on even calls, odd pairs on odd calls), accepting
    with min(1, exp((beta_a - beta_b) (H_a - H_b))) -- the standard detailed-balance rule
    for exchanging temperatures between two canonical (or semigrand, H = E - mu N) chains.
    """

    def __init__(self, ladder, per_rank, rank=0, world=1, seed=0, group=None):
        self.ladder = np.asarray(ladder, dtype=np.float64)
        self.n = len(self.ladder)
        self.per_rank, self.rank, self.world, self.group = int(per_rank), int(rank), int(world), group
        if self.per_rank * self.world != self.n:
            raise ValueError("ladder length must equal per_rank * world")
        self.seed = int(seed)
        self.rung_of = np.arange(self.n)  # walker -> rung
        self.calls = 0
        self.exchange_seconds = 0.0  # host wall time spent in exchange steps (run_replica_exchange), walkers idle
        self.exchange_timed = 0
        self.attempted = np.zeros(self.n - 1, dtype=np.int64)
        self.accepted = np.zeros(self.n - 1, dtype=np.int64)

    # rung_of / attempted / accepted: host arrays; after `decide_on_device` attempts the current values live on the
    # GPU and every read brings them back first (sync_from_device), so no reader ever sees a stale ladder
    def _synced(name):
        def get(self):
            self.sync_from_device()
            return getattr(self, "_" + name)

        def put(self, value):
            setattr(self, "_" + name, value)

        return property(get, put)

    rung_of, attempted, accepted = _synced("rung_of"), _synced("attempted"), _synced("accepted")
    del _synced

    # ------------------------------------------------------------------------------
    @property
    def temperatures(self):
        """Temperature of every global walker."""
        self.sync_from_device()
        return self.ladder[self.rung_of]

    def local_temperatures(self):
        a = self.rank * self.per_rank
        return self.temperatures[a:a + self.per_rank].copy()

    def gather(self, local_enthalpy, force_collective=False):
        """All-gather the per-walker enthalpies (torch tensor, float64, len per_rank) into a
        NumPy array of all walkers.  8 bytes per walker: latency-bound over xGMI.
        ``force_collective`` runs the all-gather even for a single rank (a world-size-1 process
        group), so that the multi-GPU code path can be exercised on one GPU."""
        import torch

        dist = _dist()
        ready = dist.is_available() and dist.is_initialized()
        if not ready or (self.world == 1 and not force_collective):
            return local_enthalpy.detach().cpu().numpy().astype(np.float64)
        out = torch.empty(self.n, dtype=torch.float64, device=local_enthalpy.device)
        dist.all_gather_into_tensor(out, local_enthalpy.contiguous(), group=self.group)
        return out.cpu().numpy()

    def decide(self, enthalpy):
        """Swap decisions from the gathered enthalpies: pure function of (state, enthalpy),
        identical on every rank.  Returns the list of accepted rung pairs (k, k+1)."""
        self.sync_from_device()
        if getattr(self, "_dev", None) is not None:
            self._dev = None  # (host decisions from here on: the device copies would go stale)
        parity = self.calls & 1
        walker_at = np.empty(self.n, dtype=np.int64)  # rung -> walker
        walker_at[self.rung_of] = np.arange(self.n)
        pairs = np.arange(parity, self.n - 1, 2)
        u = _philox_uniforms(self.seed, self.calls, max(len(pairs), 1))
        beta = 1.0 / (kB * self.ladder)
        # the pairs of one parity are disjoint: all decisions of an attempt at once
        enthalpy = np.asarray(enthalpy, dtype=np.float64)
        a, b = walker_at[pairs], walker_at[pairs + 1]
        expo = (beta[pairs] - beta[pairs + 1]) * (enthalpy[a] - enthalpy[b])
        with np.errstate(divide="ignore"):
            acc = (expo >= 0) | (np.log(u[: len(pairs)]) < expo)
        self.attempted[pairs] += 1
        won = pairs[acc]
        self.rung_of[a[acc]] = won + 1
        self.rung_of[b[acc]] = won
        self.accepted[won] += 1
        self.calls += 1
        return [(int(k), int(k + 1)) for k in won]

    # ---- decisions on the device (smolmc_exchange_dev) ---------------------------------------------
    # The same attempt without the host in the loop: the all-gathered enthalpies stay on the GPU, one small kernel
    # takes the decisions of `decide` -- same arithmetic, same counter-based uniforms (their logs are made on the
    # host, a block of attempts ahead of time, and uploaded once per block) -- moves the rung assignment and sets the
    # engine's temperatures.  rung_of / attempted / accepted live in device tensors meanwhile; `sync_from_device`
    # (called by every reader below) brings them back.
    LOG_U_BLOCK = 64  # attempts whose log-uniforms are uploaded together

    def _device_state(self, device):
        import torch

        if getattr(self, "_dev", None) is None or self._dev["device"] != device:
            self._dev = dict(
                device=device,
                ladder=torch.from_numpy(self.ladder).to(device),
                rung_of=torch.from_numpy(self.rung_of.astype(np.int32)).to(device),
                stats=torch.from_numpy(np.concatenate([self.attempted, self.accepted]).astype(np.int64)).to(device),
                log_u=None, log_u_first=-1,
            )
            # (the uploads ran on torch's stream, the decision kernel runs on the engine's: they must have landed)
            torch.cuda.synchronize(device)
            self._dev_dirty = False
        return self._dev

    def _log_u_row(self, st):
        """Device row of log(u) for attempt `self.calls` (uploaded LOG_U_BLOCK attempts at a time)."""
        import torch

        if st["log_u"] is None or not (st["log_u_first"] <= self.calls < st["log_u_first"] + self.LOG_U_BLOCK):
            half = max(self.n // 2, 1)
            rows = np.zeros((self.LOG_U_BLOCK, half))
            for i in range(self.LOG_U_BLOCK):
                c = self.calls + i
                npairs = len(range(c & 1, self.n - 1, 2))
                with np.errstate(divide="ignore"):  # (the draws `decide` makes for attempt c, in its order)
                    rows[i, :npairs] = np.log(_philox_uniforms(self.seed, c, max(npairs, 1))[:npairs])
            st["log_u"] = torch.from_numpy(rows).to(st["device"])
            torch.cuda.synchronize(st["device"])  # (as above: another stream reads it)
            st["log_u_first"] = self.calls
        return st["log_u"][self.calls - st["log_u_first"]]

    def decide_on_device(self, engine, enthalpy_all_dev):
        """One attempt decided by `engine`'s GPU from the device tensor of ALL walkers' enthalpies (float64, global
        order); the engine's temperatures are set by the same kernel.  Identical decisions to `decide`."""
        st = self._device_state(enthalpy_all_dev.device)
        lu = self._log_u_row(st)
        engine.exchange_dev(self.n, self.rank * self.per_rank, self.calls & 1, enthalpy_all_dev.data_ptr(),
                            st["ladder"].data_ptr(), lu.data_ptr(), st["rung_of"].data_ptr(), st["stats"].data_ptr())
        self.calls += 1
        self._dev_dirty = True

    def sync_from_device(self):
        """Bring rung_of / attempted / accepted back from the device after `decide_on_device` attempts."""
        if getattr(self, "_dev_dirty", False):
            import torch

            self._dev_dirty = False  # (first: the assignments below go through the syncing properties)
            torch.cuda.synchronize(self._dev["device"])
            self.rung_of = self._dev["rung_of"].cpu().numpy().astype(np.int64)
            stats = self._dev["stats"].cpu().numpy()
            self.attempted, self.accepted = stats[: self.n - 1].copy(), stats[self.n - 1:].copy()
        return self

    def exchange(self, local_enthalpy, force_collective=False):
        """gather + decide; returns this rank's new temperatures (NumPy, len per_rank)."""
        self.decide(self.gather(local_enthalpy, force_collective))
        return self.local_temperatures()

    @property
    def acceptance(self):
        self.sync_from_device()
        return self.accepted / np.maximum(self.attempted, 1)


def geometric_ladder(t_min, t_max, n):
    """Geometric temperature ladder (400-2000 K)."""
    return np.geomspace(t_min, t_max, n)


def run_replica_exchange(engine, rex, n_exchanges, steps_between, device=None, collective=None, device_decide=None):
    """Alternate ``steps_between`` MC steps on every walker with one exchange attempt.

    ``engine`` is a smol_amd.engine.Engine holding this rank's ``rex.per_rank`` walkers.
    Collective path (several ranks, or ``collective=True``) on RCCL: the enthalpies are exported
    device-to-device into a torch tensor (smolmc_export_enthalpy_dev), all-gathered
    without staging through the host, and the new temperatures go back through a device
    tensor (smolmc_import_temperature_dev); initialise torch.cuda / the process group BEFORE
    creating the engine, as bench.py does.  Collective path on a CPU backend (gloo: the CPU
    tests, ``bench.py --oversubscribe`` / ``--dry-run``): the same all-gather over host tensors.
    Single rank (default): no collective is needed, the enthalpies are read back directly and
    the temperatures uploaded with set_temperature.  All paths take the same decisions
    (tests/test_gpu_device_plumbing.py, tests/test_parallel_gloo.py).
    ``device_decide`` (default: SMOLMC_REX_DEVICE_DECIDE=1 in the environment; collective device path only): the swap
    decisions are taken by a kernel on the all-gathered device tensor (smolmc_exchange_dev) instead of NumPy on a
    host copy -- no device-to-host copy, no host arithmetic and no temperature upload per attempt; `rex.rung_of` /
    `attempted` / `accepted` are brought back when read (`rex.sync_from_device()`)."""
    import os

    if device_decide is None:
        device_decide = os.environ.get("SMOLMC_REX_DEVICE_DECIDE") == "1"
    dist = _dist()
    ready = dist.is_available() and dist.is_initialized()
    multi = ready and (rex.world > 1 if collective is None else bool(collective))
    if collective and not ready:
        raise RuntimeError("collective=True needs an initialised torch.distributed process group")
    on_device = multi and (device is not None or collective_device() == "cuda")
    buf = tbuf = allbuf = None
    if multi:
        import torch
    if on_device:
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        buf = torch.empty(rex.per_rank, dtype=torch.float64, device=dev)
        tbuf = torch.empty(rex.per_rank, dtype=torch.float64, device=dev)
    import time

    engine.set_temperature(rex.local_temperatures())
    for _ in range(n_exchanges):
        engine.run(steps_between)
        if hasattr(engine, "sync"):
            engine.sync()  # (the exchange needs the launch's enthalpies anyway; timed from here: its own latency)
        t_ex = time.perf_counter()
        if on_device and device_decide:
            engine.export_enthalpy(buf.data_ptr())
            if allbuf is None:
                allbuf = torch.empty(rex.n, dtype=torch.float64, device=dev)
            if rex.world > 1 or collective:
                dist.all_gather_into_tensor(allbuf, buf, group=rex.group)
            else:
                allbuf.copy_(buf)
            torch.cuda.current_stream().synchronize()  # (the kernel runs on the engine's stream)
            rex.decide_on_device(engine, allbuf)
        elif on_device:
            engine.export_enthalpy(buf.data_ptr())
            new_t = rex.exchange(buf, force_collective=True)
            tbuf.copy_(torch.from_numpy(new_t))
            torch.cuda.current_stream().synchronize()
            engine.import_temperature(tbuf.data_ptr())
        elif multi:
            mine = torch.from_numpy(np.ascontiguousarray(engine.get_enthalpy(), dtype=np.float64))
            engine.set_temperature(rex.exchange(mine, force_collective=True))
        else:
            rex.decide(engine.get_enthalpy())
            engine.set_temperature(rex.local_temperatures())
        rex.exchange_seconds += time.perf_counter() - t_ex
        rex.exchange_timed += 1
    if on_device and device_decide and hasattr(engine, "sync"):
        engine.sync()  # (the last attempt's temperatures are in place when this returns)
    return rex
