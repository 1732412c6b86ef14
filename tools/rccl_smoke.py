#!/usr/bin/env python3
"""RCCL smoke on one GPU: a world-size-1 nccl group through the exact calls bench.py makes at N > 1 (barrier with
device_ids, all_gather_into_tensor of embeddings, fp64 MAX all_reduce)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kaldi-tflite_amd"))
import torch
import torch.distributed as dist
from kaldi_tflite_amd import parallel

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
parallel.barrier(2)                                        # world argument > 1 forces the collective path
y = torch.randn((1024, 128), device="cuda")
out = torch.empty((1024, 128), device="cuda")
dist.all_gather_into_tensor(out, y)
assert torch.equal(out, y)
t = parallel.max_over_ranks(1.25, 2, torch.device("cuda", 0))
assert t == 1.25
dist.destroy_process_group()
print("rccl smoke ok")
