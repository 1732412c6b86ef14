"""GPU tests added in round 5 (run with `-m gpu` on an MI355X): one model object under concurrent host threads, the library's build
id and clock probe."""

import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import synth
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import ops

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_model_object_under_concurrent_host_threads():
    """tools/thread_probe.py --shared-model: FOUR host threads drive ONE XvectorExtractor at once -- two on streams of their own, two on
    the default stream; batches of different shapes, ragged ones with utterances below MIN_FRAMES (the per-utterance second pass and
    its pinned flag) -- and every x-vector equals the single-threaded result bit for bit. The reference's layers are stateless after
    build (kaldi_tflite/lib/layers/tdnn/tdnn.py:251-280, normalization/cmvn.py:186-250): nothing a call sets may live on
    the model object."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "thread_probe.py"), "25", "--shared-model"], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and "on ONE shared model: every x-vector equals the single-threaded result" in out.stdout, \
        out.stdout[-3000:] + out.stderr[-2000:]


def test_last_lens_and_short_count_are_those_of_the_calling_thread():
    import threading
    cfg, w = synth.extractor_cfg(), synth.make_weights(seed=3)
    m = synth.build_extractor(ktf, cfg, w, gemm="f16mx")
    a = torch.as_tensor(synth.make_wav(2, 160000, seed=1), device="cuda")
    b = torch.as_tensor(synth.make_wav(5, 96000, seed=2), device="cuda")
    m(a)
    seen = {}

    def other():
        assert m.last_lens is None and m.last_short_count == 0          # this thread has not called yet
        m(b)
        seen["shape"] = tuple(m.last_lens.shape)

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen["shape"] == (5,) and tuple(m.last_lens.shape) == (2,)


def test_build_id_and_clock_probe():
    bid = ops.build_id()
    assert len(bid) == 16 and int(bid, 16) >= 0
    out = torch.zeros(4, dtype=torch.int64, device="cuda")
    st = torch.cuda.Stream()
    ops.clock_probe(out, 20000, st)                      # 20 ms on an otherwise idle chip
    torch.cuda.synchronize()
    clk, ticks, lo, hi = [int(v) for v in out.cpu()]
    assert 1_900_000 <= ticks <= 2_600_000               # 100 MHz ticks of 20 ms (+ the loop's last sleep)
    mhz = 100.0 * clk / ticks
    assert 400.0 <= mhz <= 2500.0 and 0 < lo <= hi <= 2_600_000, (mhz, lo, hi)
