#!/usr/bin/env python3
"""Timing-only ablation builds of csrc/frontend512.hip (WRONG results by design): which part of a frame costs what.
Builds libktf_abl_fe_<name>.so beside the product library; on the GPU box:
  for l in kaldi-tflite_amd/kaldi_tflite_amd/libktf_abl_fe_*.so; do KTF_ALLOW_LIBRARY_OVERRIDE=1 KTF_LIBRARY=$PWD/$l python tools/fe_time.py; done"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "kaldi-tflite_amd", "csrc")
src = open(os.path.join(CS, "frontend512.hip")).read()


def rep(s, a, b):
    assert s.count(a) >= 1, a
    return s.replace(a, b)


V = {}
V["base"] = lambda s: s
V["no_dct"] = lambda s: rep(s, "        for (int m4 = 0; m4 < F5_MAXMEL / 4; ++m4) {\n            const f32x4 f =", "        for (int m4 = 0; m4 < 1; ++m4) {\n            const f32x4 f =")
V["no_mel"] = lambda s: rep(s, "        for (int j4 = 0; j4 < F5_MAXW / 4; ++j4) {\n            if (4 * j4 < maxw) {", "        for (int j4 = 0; j4 < 1; ++j4) {\n            if (4 * j4 < maxw) {")
V["no_transposes"] = lambda s: rep(rep(rep(s, "        transpose4<0, 5>(z, lane);", ""), "        transpose4<4, 3>(z, lane);", ""), "        transpose4<2, 1>(z, lane);", "")
V["no_fft"] = lambda s: rep(rep(rep(rep(V["no_transposes"](s), "        bfly4(z);                                   // over k (stride 64)", ""), "        bfly4(z);\n#pragma unroll\n        for (int r = 1; r < 4; ++r) z[r] = cmulf(z[r], tw2[r - 1]);", ""),
                            "        bfly4(z);\n#pragma unroll\n        for (int r = 1; r < 4; ++r) z[r] = cmulf(z[r], tw3[r - 1]);", ""), "        bfly4(z);\n        // natural order through LDS", "        // natural order through LDS")
V["no_preemph"] = lambda s: rep(s, "const bool c_pre = STD ? true : cfg.preemph > 0.0f;", "const bool c_pre = false;")
V["no_dc_energy"] = lambda s: rep(rep(s, "const bool c_dc = STD ? true : cfg.remove_dc != 0;", "const bool c_dc = false;"), "const bool c_raw_e = STD ? true : (cfg.use_energy && cfg.raw_energy);", "const bool c_raw_e = false;")
V["no_split"] = lambda s: rep(s, "            pw[j] = 0.25f * fmaf(x2.x, x2.x, x2.y * x2.y);", "            pw[j] = zk.x;")
names = sys.argv[1:] or list(V)
srcs = re.search(r"^SRCS := (.*)$", open(os.path.join(CS, "Makefile")).read(), re.M).group(1).split()
objs = [f[:-4] + ".o" for f in srcs if f != "frontend512.hip"]
subprocess.check_call(["make", "-j8"], cwd=CS, stdout=subprocess.DEVNULL)
for n in names:
    path = f"/tmp/fe512_{n}.hip"
    open(path, "w").write(V[n](src))
    out = os.path.join(ROOT, "kaldi-tflite_amd", "kaldi_tflite_amd", f"libktf_abl_fe_{n}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-I" + CS, "-c", path, "-o", f"/tmp/fe512_{n}.o"], cwd=CS)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(CS, o) for o in objs] + [f"/tmp/fe512_{n}.o", "-o", out])
    print("built", out)
