#!/usr/bin/env python3
"""Phase stamps of the fused VAD + CMVN kernel on one 10 s utterance, per split workgroup (measurement tool).

    python tools/vc_phase_probe.py build     here: patches a scratch copy of csrc/vad_cmvn.hip (VC_PROBE -> a 100 MHz wall-clock stamp of
                                             thread 0 of every workgroup of utterance 0) and links libktf_abl_vcprobe.so
    python tools/vc_phase_probe.py           on the GPU box: microseconds per phase and split
"""
import ctypes, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "kaldi-tflite_amd", "csrc")
OUT = os.path.join(ROOT, "kaldi-tflite_amd", "kaldi_tflite_amd", "libktf_abl_vcprobe.so")

if sys.argv[1:] == ["build"]:
    s = open(os.path.join(CS, "vad_cmvn.hip")).read()
    a = "#define VC_PROBE(k)\n"
    assert a in s
    s = s.replace(a, """__device__ long long vc_stamps[8 * 16];
#define VC_PROBE(k) if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y < 16) vc_stamps[(k) * 16 + blockIdx.y] = wall_clock64();
extern "C" int ktf_probe_vc_stamps(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vc_stamps), sizeof(vc_stamps)); }
""")
    # an early return of a split leaves its stamps at the previous call's values: stamp the exits too
    scratch = os.path.join(CS, "_vcprobe.hip")
    open(scratch, "w").write(s)
    srcs = re.search(r"^SRCS := (.*)$", open(os.path.join(CS, "Makefile")).read(), re.M).group(1).split()
    objs = [os.path.join(CS, f[:-4] + ".o") for f in srcs if f != "vad_cmvn.hip"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"),
                           "-c", scratch, "-o", os.path.join(CS, "_vcprobe.o")])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [os.path.join(CS, "_vcprobe.o"), "-o", OUT])
    os.remove(scratch); os.remove(os.path.join(CS, "_vcprobe.o"))
    print("built", OUT)
    sys.exit(0)

os.environ["KTF_ALLOW_LIBRARY_OVERRIDE"] = "1"
os.environ["KTF_LIBRARY"] = OUT
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, synth
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import _lib as L
dev = torch.device("cuda", 0)
lib = L.load()
mdl = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm="f32")
B = int(os.environ.get("B", "1"))
wav = torch.as_tensor(synth.make_wav(B, 160000, seed=3), device=dev)
host = (ctypes.c_longlong * 128)()
acc, n = None, 0
for it in range(30):
    mdl(wav)
    torch.cuda.synchronize()
    assert lib.ktf_probe_vc_stamps(host) == 0
    d = np.array(host[:], dtype=np.float64).reshape(8, 16) * 0.01          # microseconds
    if it >= 10:
        d = d - d[0][d[0] > 0].min()
        acc = d if acc is None else acc + d
        n += 1
acc /= n
names = ["start", "C0 mean", "vote + scan", "rows staged", "block sums", "windows + stores", "edge frames", "end"]
live = [j for j in range(16) if acc[7, j] > 0]
print("split            " + "".join(f"{j:8d}" for j in live))
for k, nm in enumerate(names):
    print(f"{nm:17s}" + "".join(f"{acc[k, j]:8.2f}" for j in live))
