import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    # the C-ABI library is a build artefact (git-ignored): build it once if a fresh checkout is tested before
    # __graft_entry__.build() ran (hipcc cross-compiles gfx950 without a GPU); a failed build surfaces in the tests
    so = os.path.join(ROOT, "kaldi-tflite_amd", "kaldi_tflite_amd", "libktf_hip.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "kaldi-tflite_amd", "csrc")], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
