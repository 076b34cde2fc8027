"""The multi-GPU device plumbing on ONE GPU: the three C-ABI entry points that only
the multi-rank path uses -- smolmc_set_stream, smolmc_export_enthalpy_dev,
smolmc_import_temperature_dev -- and the collective branch of run_replica_exchange driven
through a world-size-1 RCCL ("nccl") process group.  Independent walkers are the reference's
semantics (smol/moca/sampler/sampler.py:436-440); the exchange ladder is new functionality and
is validated by equality of the two code paths (collective vs. direct) on identical inputs."""

import os
import socket

import numpy as np
import pytest

from smol_amd import capi, parallel, synth
from smol_amd.engine import Engine

pytestmark = pytest.mark.gpu


def _engine(R=64, step=capi.STEP_SWAP):
    model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [6, 6, 6])
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=3, scale=0.02))
    rng = np.random.default_rng(0)
    occ = (rng.random((R, sc.num_sites)) < 0.5).astype(np.int32)
    seeds = np.arange(R, dtype=np.uint64) + np.uint64(17)
    eng = Engine(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, step))
    return eng, occ, seeds


@pytest.fixture(scope="module")
def nccl_world1():
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_export_enthalpy_and_import_temperature_device_pointers():
    import torch

    eng, occ, seeds = _engine()
    eng.set_state(occ, seeds, 1500.0)
    eng.run(500)
    buf = torch.full((eng.R,), float("nan"), dtype=torch.float64, device="cuda")
    eng.export_enthalpy(buf.data_ptr())
    np.testing.assert_array_equal(buf.cpu().numpy(), eng.get_enthalpy())  # same bits
    # temperatures through a device array == temperatures through the host entry point
    temps = np.linspace(300.0, 3000.0, eng.R)
    ref, occ2, _ = _engine()
    ref.set_state(occ, seeds, 1500.0)
    ref.run(500)
    ref.set_temperature(temps)
    eng.import_temperature(torch.from_numpy(temps).cuda().data_ptr())
    eng.run(2000)
    ref.run(2000)
    a, b = eng.get_state(), ref.get_state()
    assert np.array_equal(a["occupancy"], b["occupancy"])
    assert np.array_equal(a["n_accepted"], b["n_accepted"])
    np.testing.assert_array_equal(a["enthalpy"], b["enthalpy"])
    # the temperatures took effect: hot walkers accept more than cold ones
    acc = a["n_accepted"].astype(float)
    assert acc[-8:].mean() > acc[:8].mean()


def test_set_stream_runs_on_a_torch_stream():
    import torch

    eng, occ, seeds = _engine()
    ref, _, _ = _engine()
    side = torch.cuda.Stream()
    eng.set_stream(side.cuda_stream)
    for e in (eng, ref):
        e.set_state(occ, seeds, 2000.0)
    # work queued on the torch stream before the launch is ordered before it: the engine's
    # kernels and torch's share one queue, so a tensor written by torch is visible to a
    # following import_temperature without a host synchronisation
    with torch.cuda.stream(side):
        t = torch.full((eng.R,), 2000.0, dtype=torch.float64, device="cuda")
        t[::2] = 500.0
        eng.import_temperature(t.data_ptr())
        eng.run(1500)
        h = torch.empty(eng.R, dtype=torch.float64, device="cuda")
        eng.export_enthalpy(h.data_ptr())
    side.synchronize()
    temps = np.full(eng.R, 2000.0)
    temps[::2] = 500.0
    ref.set_temperature(temps)
    ref.run(1500)
    a, b = eng.get_state(), ref.get_state()
    assert np.array_equal(a["occupancy"], b["occupancy"])
    np.testing.assert_array_equal(h.cpu().numpy(), b["enthalpy"])
    # HIP-event timing follows the launch stream
    assert eng.last_kernel_ms() > 0.0
    # handing the stream back: a null stream is legal and the handle keeps working
    eng.set_stream(0)
    eng.run(10, sync=True)


def test_collective_exchange_branch_equals_direct_branch(nccl_world1):
    """run_replica_exchange(collective=True) goes export_enthalpy_dev -> all_gather_into_tensor
    (RCCL, world size 1) -> decide -> import_temperature_dev; it must take exactly the decisions
    of the direct single-rank branch and leave identical walkers."""
    R = 64
    ladder = parallel.geometric_ladder(400.0, 2400.0, R)
    out = []
    for collective in (True, False):
        eng, occ, seeds = _engine(R)
        eng.set_state(occ, seeds, ladder)
        rex = parallel.ReplicaExchange(ladder, R, rank=0, world=1, seed=5)
        parallel.run_replica_exchange(eng, rex, 12, 216, collective=collective)
        out.append((eng.get_state(), rex.rung_of.copy(), rex.accepted.copy()))
    (sa, ra, aa), (sb, rb, ab) = out
    assert np.array_equal(ra, rb) and np.array_equal(aa, ab)
    assert aa.sum() > 0 and sorted(ra) == list(range(R))
    assert np.array_equal(sa["occupancy"], sb["occupancy"])
    np.testing.assert_array_equal(sa["enthalpy"], sb["enthalpy"])


@pytest.mark.parametrize("R", [64, 65, 2], ids=["even", "odd", "two"])
def test_exchange_decided_on_the_device_equals_the_numpy_decisions(nccl_world1, R):
    """smolmc_exchange_dev (round 6): export_enthalpy_dev -> all_gather_into_tensor (RCCL, world size 1) -> ONE kernel
    that takes the swap decisions from the device tensor, moves the rung assignment and sets the temperatures --
    against the host path (NumPy over a host copy, upload of the new temperatures): the same rung assignment after
    every block of attempts, the same acceptance counters, identical walkers.  Ladders of even and odd length (the
    odd-parity attempt has one pair fewer) and of two walkers (the odd attempt has none); more attempts than one
    uploaded block of log-uniforms; a host-side attempt in between (`decide`) drops the device copies and a later
    device attempt starts from the host state again."""
    ladder = parallel.geometric_ladder(400.0, 2400.0, R)
    out = []
    for dev in (True, False):
        eng, occ, seeds = _engine(R)
        eng.set_state(occ, seeds, ladder)
        rex = parallel.ReplicaExchange(ladder, R, rank=0, world=1, seed=5)
        trace = []
        for n_ex in (1, 7, parallel.ReplicaExchange.LOG_U_BLOCK + 3):
            parallel.run_replica_exchange(eng, rex, n_ex, 54, collective=True, device_decide=dev)
            trace.append((rex.temperatures.copy(), rex.attempted.copy(), rex.accepted.copy()))
        rex.decide(eng.get_enthalpy())  # (a host attempt in the middle of either run)
        eng.set_temperature(rex.local_temperatures())
        parallel.run_replica_exchange(eng, rex, 5, 54, collective=True, device_decide=dev)
        out.append((eng.get_state(), rex.rung_of.copy(), rex.accepted.copy(), rex.attempted.copy(), trace, rex.calls))
    (sa, ra, aa, ta, tra, ca), (sb, rb, ab, tb, trb, cb) = out
    assert ca == cb == 1 + 7 + parallel.ReplicaExchange.LOG_U_BLOCK + 3 + 1 + 5
    for (t1, at1, ac1), (t2, at2, ac2) in zip(tra, trb):
        np.testing.assert_array_equal(t1, t2)
        assert np.array_equal(at1, at2) and np.array_equal(ac1, ac2)
    assert np.array_equal(ra, rb) and np.array_equal(aa, ab) and np.array_equal(ta, tb)
    assert sorted(ra) == list(range(R)) and ta.sum() > 0
    if R > 2:
        assert aa.sum() > 0
    assert np.array_equal(sa["occupancy"], sb["occupancy"])
    np.testing.assert_array_equal(sa["enthalpy"], sb["enthalpy"])


def test_exchange_dev_argument_checks(nccl_world1):
    import torch

    from smol_amd.engine import EngineError

    eng, occ, seeds = _engine(8)
    eng.set_state(occ, seeds, 1000.0)
    H = torch.zeros(8, dtype=torch.float64, device="cuda")
    lad = torch.full((8,), 1000.0, dtype=torch.float64, device="cuda")
    lu = torch.zeros(4, dtype=torch.float64, device="cuda")
    ro = torch.arange(8, dtype=torch.int32, device="cuda")
    with pytest.raises((EngineError, ValueError), match="out of range"):
        eng.exchange_dev(8, 1, 0, H.data_ptr(), lad.data_ptr(), lu.data_ptr(), ro.data_ptr())
    with pytest.raises((EngineError, ValueError), match="parity"):
        eng.exchange_dev(8, 0, 2, H.data_ptr(), lad.data_ptr(), lu.data_ptr(), ro.data_ptr())
    with pytest.raises((EngineError, ValueError), match="16384"):
        eng.exchange_dev(20000, 0, 0, H.data_ptr(), lad.data_ptr(), lu.data_ptr(), ro.data_ptr())
    eng.exchange_dev(8, 0, 0, H.data_ptr(), lad.data_ptr(), lu.data_ptr(), ro.data_ptr())  # (stats may be omitted)
    eng.sync()
    # equal enthalpies: exponent 0 >= 0, every pair of the even attempt swaps
    assert ro.cpu().tolist() == [1, 0, 3, 2, 5, 4, 7, 6]
    eng.close()


def test_global_sums_over_nccl(nccl_world1):
    import torch

    t = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, device="cuda")
    got = parallel.global_sums(t.clone())
    assert torch.equal(got, t)  # world size 1: identity, but through the RCCL all-reduce path
    nccl_world1.all_reduce(t)
    assert torch.equal(got, t)


def test_sampler_shards_walkers_by_rank():
    """The smol-shaped Sampler owns only its rank's block of the global walkers and walker g has
    the same seed / start / trajectory whatever the world size."""
    from smol_amd import moca

    model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [6, 6, 6])
    ens = moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model, seed=3, scale=0.02))
    nw = 12
    rng = np.random.default_rng(1)
    occ = (rng.random((nw, sc.num_sites)) < 0.5).astype(np.int32)
    seeds = list(range(100, 100 + nw))
    whole = moca.Sampler.from_ensemble(ens, temperature=1800.0, nwalkers=nw, seeds=seeds,
                                       rank=0, world_size=1)
    whole.run(600, occ, thin_by=200)
    ref = whole.samples.get_occupancies(flat=False)
    parts = []
    for r in range(3):
        s = moca.Sampler.from_ensemble(ens, temperature=1800.0, nwalkers=nw, seeds=seeds,
                                       rank=r, world_size=3)
        assert s.walker_range == (4 * r, 4)
        s.run(600, occ, thin_by=200)  # global occupancies in, the rank takes its block
        parts.append(s.samples.get_occupancies(flat=False))
        tot = s.global_statistics()
        assert tot["walkers"] == 4  # no process group here: local sums
    assert np.array_equal(np.concatenate(parts, axis=1), ref)
