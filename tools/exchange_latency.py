#!/usr/bin/env python3
"""Host wall time of one replica-exchange attempt over 2048 walkers on one GPU (walkers idle meanwhile), three ways:
the single-rank read-back (get_enthalpy -> NumPy decisions -> set_temperature), the collective path with host decisions
(export_enthalpy_dev -> RCCL all_gather_into_tensor at world size 1 -> device-to-host copy -> NumPy -> upload ->
import_temperature_dev) and the collective path with the decisions taken by smolmc_exchange_dev on the device tensor
(round 6).  One JSON line each.   python tools/exchange_latency.py [--walkers 2048] [--attempts 200]"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from smol_amd import capi, parallel, synth  # noqa: E402
from smol_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--walkers", type=int, default=2048)
    ap.add_argument("--attempts", type=int, default=200)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    model = synth.build_cluster_model(synth.fcc_prim(), {2: 6.0, 3: 5.0})
    sc = synth.build_supercell(model, [6, 6, 6])
    tab = capi.TableSet.from_synth(sc, synth.random_coefs(model, seed=3, scale=0.02))
    R = a.walkers
    occ = (np.random.default_rng(0).random((R, sc.num_sites)) < 0.5).astype(np.int32)
    ladder = parallel.geometric_ladder(400.0, 2400.0, R)
    for name, kw in (("single rank: read-back + NumPy + set_temperature", dict()),
                     ("collective, host decisions (NumPy on a host copy)", dict(collective=True, device_decide=False)),
                     ("collective, device decisions (smolmc_exchange_dev)", dict(collective=True, device_decide=True))):
        eng = Engine(tab, capi.make_config(R, capi.KERNEL_METROPOLIS, capi.STEP_SWAP))
        eng.set_state(occ, np.arange(R, dtype=np.uint64), ladder)
        rex = parallel.ReplicaExchange(ladder, R, seed=5)
        parallel.run_replica_exchange(eng, rex, 10, 20, **kw)  # warm-up (buffers, first upload of the log-uniforms)
        rex.exchange_seconds, rex.exchange_timed = 0.0, 0
        t0 = time.perf_counter()
        parallel.run_replica_exchange(eng, rex, a.attempts, 20, **kw)
        wall = time.perf_counter() - t0
        print(json.dumps(dict(path=name, walkers=R, attempts=a.attempts,
                              exchange_ms_per_attempt=rex.exchange_seconds / rex.exchange_timed * 1e3,
                              loop_ms_per_attempt=wall / a.attempts * 1e3, exchange_acceptance=float(rex.acceptance.mean()))))
        eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
