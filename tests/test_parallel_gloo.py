"""The N>1 path on CPU: world_size-2 gloo process groups exercising the sharding,
the global-average all-reduce and the replica-exchange protocol (the only places the
multi-GPU run communicates).  Replica exchange is NEW functionality without a reference
counterpart: it is validated by invariants (rank agreement, permutation, detailed
balance), not parity."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from smol_amd import parallel


def test_shard_covers_everything():
    for total in (1, 7, 4096, 4099):
        for world in (1, 2, 3, 8):
            got = [parallel.shard(total, r, world) for r in range(world)]
            assert sum(c for _, c in got) == total
            assert got[0][0] == 0
            for (a, c), (b, _) in zip(got, got[1:]):
                assert a + c == b


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # ---- global sums ------------------------------------------------------------
        local = torch.tensor([1.0 + rank, 10.0 * (rank + 1), 1.0], dtype=torch.float64)
        tot = parallel.global_sums(local.clone()).numpy()
        # ---- replica exchange over 2 ranks x 4 walkers ------------------------------
        per = 4
        ladder = parallel.geometric_ladder(400.0, 2000.0, per * world)
        rex = parallel.ReplicaExchange(ladder, per, rank, world, seed=99)
        rng = np.random.default_rng(5)  # same stream on both ranks -> a shared "truth"
        history = []
        for it in range(40):
            H_all = rng.normal(0.0, 0.5, per * world) - 3.0 / rex.temperatures * 1000.0
            mine = torch.tensor(H_all[rank * per:(rank + 1) * per], dtype=torch.float64)
            gathered = rex.gather(mine)
            assert np.array_equal(gathered, H_all)
            rex.decide(gathered)
            history.append(rex.rung_of.copy())
        q.put((rank, tot, np.array(history), rex.local_temperatures(), rex.accepted.copy(),
               rex.attempted.copy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_protocol():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, tot0, h0, t0, acc0, att0), (r1, tot1, h1, t1, acc1, att1) = res
    np.testing.assert_allclose(tot0, [3.0, 30.0, 2.0])
    np.testing.assert_allclose(tot0, tot1)
    assert np.array_equal(h0, h1)  # every rank took identical decisions
    for row in h0:
        assert sorted(row) == list(range(8))  # rungs stay a permutation of the walkers
    assert np.array_equal(acc0, acc1) and att0.sum() > 0
    ladder = parallel.geometric_ladder(400.0, 2000.0, 8)
    np.testing.assert_allclose(np.concatenate([t0, t1]), ladder[h0[-1]])
    assert acc0.sum() > 0  # swaps do happen


def test_exchange_rule_is_metropolis_for_two_rungs():
    """Acceptance frequency of a fixed pair equals min(1, exp((b0-b1)(H_a-H_b)))."""
    ladder = np.array([500.0, 1000.0])
    beta = 1.0 / (parallel.kB * ladder)
    dH = 0.05
    expect = min(1.0, np.exp((beta[0] - beta[1]) * (-dH)))
    hits, n = 0, 4000
    for s in range(n):
        rex = parallel.ReplicaExchange(ladder, 2, seed=s)
        hits += len(rex.decide(np.array([0.0, dH])))  # walker 0 (cold) lower by dH: unfavourable
    assert abs(hits / n - expect) < 4 * np.sqrt(expect * (1 - expect) / n)
    rex = parallel.ReplicaExchange(ladder, 2, seed=1)
    assert rex.decide(np.array([dH, 0.0])) == [(0, 1)]  # favourable swaps are always taken


def test_identical_ladder_leaves_temperatures_unchanged():
    """SURVEY 8e invariant: with equal temperatures every swap is accepted and changes
    nothing observable."""
    rex = parallel.ReplicaExchange(np.full(6, 800.0), 6, seed=3)
    for _ in range(10):
        rex.decide(np.random.default_rng(0).normal(size=6))
    np.testing.assert_allclose(rex.local_temperatures(), 800.0)
    assert rex.accepted.sum() == rex.attempted.sum()


def test_ladder_length_mismatch():
    with pytest.raises(ValueError):
        parallel.ReplicaExchange(np.ones(5), 2, 0, 2)


def test_vectorised_exchange_decisions_equal_the_pairwise_loop():
    """ReplicaExchange.decide takes all disjoint pairs of one parity at once; the pair-by-pair
    statement of the same rule (exchange of neighbouring rungs with probability
    min(1, exp((beta_k - beta_k+1) (H_a - H_b))), alternating parity) must give the same rungs."""
    rng = np.random.default_rng(4)
    n = 37
    ladder = parallel.geometric_ladder(300.0, 3000.0, n)
    rex = parallel.ReplicaExchange(ladder, n, seed=9)
    rung_of = np.arange(n)
    beta = 1.0 / (parallel.kB * ladder)
    for call in range(12):
        H = rng.normal(0.0, 2.0, n)
        u = parallel._philox_uniforms(9, call, max(len(range(call & 1, n - 1, 2)), 1))
        walker_at = np.empty(n, dtype=np.int64)
        walker_at[rung_of] = np.arange(n)
        want = []
        for j, k in enumerate(range(call & 1, n - 1, 2)):
            a, b = walker_at[k], walker_at[k + 1]
            expo = (beta[k] - beta[k + 1]) * (H[a] - H[b])
            if expo >= 0 or np.log(u[j]) < expo:
                rung_of[a], rung_of[b] = k + 1, k
                want.append((k, k + 1))
        assert rex.decide(H) == want
        assert np.array_equal(rex.rung_of, rung_of)
    assert rex.attempted.sum() == sum(len(range(c & 1, n - 1, 2)) for c in range(12))


def _sampler_worker(rank, world, port, q):
    """Two ranks of the smol-shaped Sampler: each owns its block of the global walkers; the
    recorded samples are filled in by hand (no GPU here) and reduced with global_statistics."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from smol_amd import moca, synth

        model = synth.build_cluster_model(synth.fcc_prim(), {2: 4.5})
        sc = synth.build_supercell(model, [3, 3, 3])
        ens = moca.Ensemble.from_cluster_expansion(sc, synth.random_coefs(model, seed=1))
        nw = 6
        s = moca.Sampler.from_ensemble(ens, temperature=900.0, nwalkers=nw, seeds=list(range(50, 50 + nw)))
        first, count = s.walker_range  # picked up from the process group
        F = len(ens.natural_parameters)
        glob = np.random.default_rng(0).normal(size=(4, nw))  # "enthalpies" of all walkers, same on both ranks
        acc = np.random.default_rng(1).random((4, nw)) < 0.5
        s.samples.append_block(dict(
            occupancy=np.zeros((4, count, sc.num_sites), np.int32), features=np.zeros((4, count, F)),
            enthalpy=glob[:, first:first + count, None], temperature=np.full((4, count, 1), 900.0),
            accepted=acc[:, first:first + count, None]), thinned_by=10)
        st = s.global_statistics()
        q.put((rank, first, count, s.seeds, st, float(glob.mean()), float(glob.var()), float(acc.mean())))
    finally:
        dist.destroy_process_group()


def test_sampler_shards_and_reduces_over_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sampler_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, f0, c0, seeds0, st0, mean, var, accm), (_, f1, c1, seeds1, st1, _, _, _) = res
    assert (f0, c0, f1, c1) == (0, 3, 3, 3)
    assert seeds0 == [50, 51, 52] and seeds1 == [53, 54, 55]  # seeds follow the GLOBAL walker index
    for st in (st0, st1):  # every rank holds the statistics of ALL walkers
        assert st["walkers"] == 6 and st["samples"] == 24
        np.testing.assert_allclose(st["mean_enthalpy"], mean, rtol=1e-12)
        np.testing.assert_allclose(st["enthalpy_variance"], var, rtol=1e-10)
        np.testing.assert_allclose(st["acceptance"], accm, rtol=1e-12)


class _StubEngine:
    """What run_replica_exchange needs from an Engine, without a GPU: enthalpies that depend on
    the walker's current temperature through a fixed per-walker offset (so that the exchange
    decisions feed back into later ones, as they do in a real run)."""

    def __init__(self, first, n, total):
        self.offset = np.random.default_rng(17).normal(0.0, 0.3, total)[first:first + n]
        self.T = np.full(n, 1000.0)
        self.steps = 0

    def set_temperature(self, t):
        self.T = np.broadcast_to(np.asarray(t, dtype=np.float64), self.T.shape).copy()

    def run(self, nsteps, sync=False):
        self.steps += int(nsteps)

    def get_enthalpy(self):
        return self.offset - 2.0e3 / self.T + 1e-3 * self.steps


def _rex_driver_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        per = 5
        ladder = parallel.geometric_ladder(400.0, 2000.0, per * world)
        rex = parallel.ReplicaExchange(ladder, per, rank, world, seed=3)
        eng = _StubEngine(rank * per, per, per * world)
        parallel.run_replica_exchange(eng, rex, 25, 100)  # host-staged all-gather (gloo) inside
        q.put((rank, rex.rung_of.copy(), eng.T.copy(), eng.steps, rex.accepted.copy()))
    finally:
        dist.destroy_process_group()


def test_run_replica_exchange_two_ranks_equals_one_rank():
    """The N-rank driver loop of config 5 (bench.py at world > 1): two ranks holding five walkers
    each must walk the global ladder exactly as one rank holding all ten does."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rex_driver_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ladder = parallel.geometric_ladder(400.0, 2000.0, 10)
    rex1 = parallel.ReplicaExchange(ladder, 10, seed=3)
    eng1 = _StubEngine(0, 10, 10)
    parallel.run_replica_exchange(eng1, rex1, 25, 100)
    (_, rung0, T0, steps0, acc0), (_, rung1, T1, steps1, acc1) = res
    assert np.array_equal(rung0, rung1) and np.array_equal(rung0, rex1.rung_of)
    assert np.array_equal(acc0, rex1.accepted) and rex1.accepted.sum() > 0
    np.testing.assert_array_equal(np.concatenate([T0, T1]), eng1.T)
    assert steps0 == steps1 == eng1.steps == 2500
