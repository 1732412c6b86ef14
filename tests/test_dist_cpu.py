"""world_size-2 gloo test (CPU) of the N>1 host path used by bench.py: env init, contiguous utterance shards,
embedding gather (equal and ragged shards), max-over-ranks timing. The per-utterance 'extractor' here is a CPU stand-in
(a fixed random projection) because there is no GPU on this box; the collective plumbing is what is under test."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kaldi-tflite_amd"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_extract(x):
    g = torch.Generator().manual_seed(0)
    P = torch.randn(x.shape[1], 8, generator=g)
    return x @ P


def _worker(rank, world, port, total, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from kaldi_tflite_amd import parallel as P
    r, lr, w = P.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(1234)
    full = torch.randn(total, 16, generator=g)
    lo, hi = P.shard_range(total, rank, world)
    local = _fake_extract(full[lo:hi])
    P.barrier(world)
    if total % world == 0:
        allv = P.gather_embeddings(local, world)
    else:
        allv = P.gather_ragged_embeddings(local, world)
    # row-sharded trial matrix (plda_trials): a bilinear stand-in scorer, every rank ends with the full matrix
    enroll = torch.randn(5, 16, generator=g)
    trials = P.plda_trials(lambda a, b: a @ b.T, full, enroll, rank, world)
    assert trials.shape == (total, 5) and torch.equal(trials, full @ enroll.T)
    t = P.max_over_ranks(1.0 + rank, world, torch.device("cpu"))
    np.save(os.path.join(out_dir, f"r{rank}.npy"), allv.numpy())
    assert t == float(world)
    torch.distributed.destroy_process_group()


def _run(total, tmp_path, world=2):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(1234)
    want = _fake_extract(torch.randn(total, 16, generator=g)).numpy()
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"r{r}.npy"))
        assert got.shape == want.shape and np.array_equal(got, want)


def test_two_rank_equal_shards(tmp_path):
    _run(8, tmp_path)


def test_two_rank_ragged_shards(tmp_path):
    _run(7, tmp_path)


def test_four_rank_shards(tmp_path):
    """The same host path at world size 4 (equal and ragged shards): what `bench.py --gpus 4` runs per rank."""
    _run(8, tmp_path, world=4)
    _run(10, tmp_path, world=4)


def test_shard_range_partitions():
    from kaldi_tflite_amd.parallel import shard_range
    for total in [0, 1, 7, 8, 8192, 8191]:
        for world in [1, 2, 3, 8]:
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ----------------------------------------------------------------------------- bench.py's own launcher (python bench.py --gpus N)
_RANK_SCRIPT = """
import os, sys, json
sys.path.insert(0, {pkg!r})
import torch
from kaldi_tflite_amd import parallel as P
rank, local_rank, world = P.init_from_env(backend="gloo")
t = torch.tensor([float(rank + 1)])
torch.distributed.all_reduce(t)
P.barrier(world)
secs = P.max_over_ranks(0.5 * (rank + 1), world, torch.device("cpu"))
if rank == 0:
    print(json.dumps({{"ranks_seen": torch.distributed.get_world_size(), "sum": float(t.item()), "max_s": secs,
                      "argv": sys.argv[1:], "master": os.environ["MASTER_ADDR"]}}))
torch.distributed.destroy_process_group()
"""


def test_bench_self_launcher_starts_the_ranks(tmp_path, monkeypatch):
    """`python bench.py --gpus N` with no launcher around it: bench.py starts N ranks itself through
    torch.distributed.run on 127.0.0.1 before it touches the GPU. The same command line is run here on a 2-rank gloo
    stand-in script, and main() is checked to take that route exactly when WORLD_SIZE is unset."""
    import json
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    script = tmp_path / "ranks.py"
    script.write_text(_RANK_SCRIPT.format(pkg=os.path.join(ROOT, "kaldi-tflite_amd")))
    cmd = bench.launch_command(2, ["--gpus", "2", "--steps", "3"], script=str(script))
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    got = json.loads(line)
    assert got == {"ranks_seen": 2, "sum": 3.0, "max_s": 1.0, "argv": ["--gpus", "2", "--steps", "3"], "master": "127.0.0.1"}
    # main(): --gpus 2 without WORLD_SIZE -> self_launch (no GPU call before it); with WORLD_SIZE set -> the rank path
    calls = []
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.subprocess, "call", lambda c, env=None: calls.append((c, env)) or 0)
    with pytest.raises(SystemExit) as ex:
        bench.main(["--gpus", "2", "--steps", "1"])
    assert ex.value.code == 0 and len(calls) == 1
    c, e = calls[0]
    assert c[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"] and "--nproc-per-node=2" in c
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
