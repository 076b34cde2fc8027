"""bench.py must be launchable for N ranks by ONE command (`python bench.py --gpus N`) and by
torch.distributed.run; without GPUs the launch + rendezvous + reductions are exercised over gloo
(--dry-run) and the real run ends with a clean "needs N GPUs" message, not a launcher error."""

import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(cmd, **kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300, **kw)


def _line(out):
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_self_spawn_dry_run_weak_and_strong():
    out = _run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "2"])
    assert out.returncode == 0, out.stderr
    rec = _line(out)
    assert rec["dry_run"] and rec["n_gpus"] == 2 and rec["scaling"] == "weak"
    assert rec["walkers_total"] == 8192 and rec["walkers_rank0"] == [0, 4096]
    assert rec["rccl_ranks"] == 2 and rec["collective_backend"].startswith("gloo")  # the rank census of the bench line
    # configs 4 and 5 as N-rank workloads: independent shards / one ladder over all ranks with the
    # exchange (host-staged all-gather here) inside the timed region
    c4, c5 = rec["other_configs"]
    assert c4["n_gpus"] == 2 and c4["replicas"] == 2048
    assert c5["replicas"] == 4096 and c5["exchanges"] == 6 and c5["exchange_attempts"] > 0
    assert c5["rungs_are_a_permutation"] and c5["ranks_agree"] and "all-gather over 2 ranks" in c5["exchange_path"]
    out = _run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "2", "--scaling", "strong"])
    assert out.returncode == 0, out.stderr
    rec = _line(out)
    assert rec["scaling"] == "strong" and rec["walkers_total"] == 4096 and rec["walkers_rank0"] == [0, 2048]


def test_torchrun_form_still_works():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2",
                "--dry-run", "--steps", "2"])
    assert out.returncode == 0, out.stderr[-2000:]
    assert _line(out)["walkers_total"] == 8192


def test_without_gpus_the_error_is_about_gpus_not_the_launcher():
    import torch

    if torch.cuda.is_available():
        return  # on a GPU box this command would really run the benchmark
    out = _run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"])
    assert out.returncode == 3
    assert "needs 2 AMD GPU" in out.stderr and "rendezvous work" in out.stderr
    out = _run([sys.executable, BENCH, "--gpus", "3"], )
    assert out.returncode == 3 and "needs 3 AMD GPU" in out.stderr


def test_gpus_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-run"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr
