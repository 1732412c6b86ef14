"""Round-4 GPU tests (run with `-m gpu` on an MI355X): the N > 1 bench path on ONE GPU, and the fused-tail guard.

* `bench.py --gpus 2` is what the driver runs on an 8-GPU node (one rank per GPU over RCCL). No such node is available to the test
  run, so two ranks share the one GPU (`KTF_SHARE_GPU=1`) over the gloo backend (`KTF_DIST_BACKEND=gloo`): fresh child processes
  started by bench.py's own self-launch before they touch the GPU. Checked: two ranks seen by the collective backend, the gather ran,
  and the gathered x-vectors are rank 0's and rank 1's own, bit for bit (BASELINE.json config 4: batch-sharded utterances, gather of
  the embeddings).
* an extractor whose LDA keeps more than 256 dimensions takes the three-launch tail (ktf_xvec_tail_f32 serves out_dim <= 256).
"""

import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import synth
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import ops
from oracle import ktf_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu_gathers_both_shards(tmp_path):
    B, sec = 48, 3.0
    dump = str(tmp_path / "xv.npy")
    env = dict(os.environ, KTF_SHARE_GPU="1", KTF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline",
           "--no-parity", "--batch", str(B), "--seconds", str(sec), "--gemm", "f32", "--dump-xvectors", dump]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["config"]["ranks_seen_by_collective_backend"] == 2 and line["config"]["gather"] is True
    assert line["config"]["collective_backend"] == "gloo"
    assert line["value"] > 0 and abs(line["value"] - 2 * B * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
    got = np.load(dump)
    assert got.shape == (2 * B, 128)
    # the same extraction in this process: rank r's waveforms come from generator seed 1234 + r (bench.py)
    cfg = synth.extractor_cfg(dither=0.0)
    mdl = synth.build_extractor(ktf, cfg, synth.make_weights(seed=4321, narrow=False), gemm="f32")
    for r in range(2):
        g = torch.Generator(device="cuda").manual_seed(1234 + r)
        wav = torch.clamp(torch.round(1000.0 * torch.randn((B, int(sec * 16000)), generator=g, device="cuda")), -32767, 32767)
        want = mdl(wav).cpu().numpy()
        assert np.array_equal(got[r * B:(r + 1) * B], want), f"rank {r}'s shard of the gathered embeddings"


def test_wide_lda_takes_the_three_launch_tail():
    """LDA output wider than the fused tail serves (300 > 256): finalize + GEMM + ktf_xvec_post_f32, same x-vectors as the oracle."""
    cfg = synth.extractor_cfg()
    w = synth.make_weights(seed=5)
    rng = np.random.default_rng(17)
    w["lda"] = (rng.standard_normal((300, 513)) / np.sqrt(512)).astype(np.float32)
    wav = synth.make_wav(3, 40000, seed=9, ragged=True)
    want = O.xvector_forward(wav, cfg, synth.oracle_layers(w), w["mean"], w["lda"], dtype=np.float64)
    for gemm in ("f32", "f16mx"):
        mdl = synth.build_extractor(ktf, cfg, w, gemm=gemm)
        mdl.xvec.min_tiles, mdl.xvec.min_frames = {}, {}
        assert not mdl._tail_fusable()
        got = mdl(torch.as_tensor(wav, device="cuda"))
        assert ops.last_kernel().startswith("tdnn_f32"), ops.last_kernel()          # tdnn6 ran as a GEMM launch of its own
        assert got.shape == (3, 300)
        assert np.abs(got.cpu().numpy() - want).max() <= 1e-4
    narrow = synth.build_extractor(ktf, cfg, synth.make_weights(seed=5), gemm="f32")
    assert narrow._tail_fusable()
