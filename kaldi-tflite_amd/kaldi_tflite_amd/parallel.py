"""
Multi-GPU driver helpers: one process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm; "gloo" on CPU
for tests). Utterances are independent end to end, so the compute path has NO collective: each rank extracts a
contiguous shard of the batch. The only communication is the optional all-gather of the (B_local, dim) embeddings.
"""

import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_from_env(backend=None):
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("KTF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(total, rank, world):
    """Contiguous [lo, hi) block of `total` utterances owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_embeddings(local, world):
    """All-gather equal-sized (B_local, dim) embeddings into (world*B_local, dim) on every rank (rank-major order)."""
    if world == 1:
        return local
    out = torch.empty((world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def gather_ragged_embeddings(local, world):
    """Shards of different sizes: pad to the largest shard, all-gather, and drop the padding."""
    if world == 1:
        return local
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = torch.zeros((m, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], 0)


def plda_trials(score_block, test, enroll, rank, world, gather=True):
    """Row-sharded N x M trial matrix: rank r scores rows shard_range(N, r, world) of `test` against ALL of `enroll`
    with `score_block(test_rows, enroll) -> (n_r, M)` (e.g. `PLDA.score` on transformed vectors). No collective on the
    scoring path; `gather=True` all-gathers the row blocks so every rank ends with the full (N, M) matrix."""
    lo, hi = shard_range(test.shape[0], rank, world)
    block = score_block(test[lo:hi], enroll)
    if not gather or world == 1:
        return block
    return gather_ragged_embeddings(block, world)


def max_over_ranks(seconds, world, device):
    if world == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value, world, device):
    """[value of rank 0, ..., value of rank world - 1] on every rank (per-rank timings of bench.py)."""
    if world == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def barrier(world):
    if world > 1:
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])     # RCCL: name the device, do not let it guess from the rank
        else:
            dist.barrier()
