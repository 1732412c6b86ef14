#!/usr/bin/env python3
"""Timing-only ablation builds of csrc/tdnn_mx.hip (measurement tool; results of these builds are WRONG by design). The product
source carries no ablation switches: this script patches a scratch copy, builds libktf_abl_<name>.so beside the product
library and prints the `KTF_LIBRARY=... KTF_ALLOW_LIBRARY_OVERRIDE=1 python bench.py --gemm f16mx --no-extra --no-cpu-baseline`
lines to run on the GPU box (tools/mx/run_ablations.sh does)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "kaldi-tflite_amd", "csrc")
src = open(os.path.join(CS, "tdnn_mx.hip")).read()

def rep(s, a, b, count=1):
    assert s.count(a) >= 1, a
    return s.replace(a, b) if count == 0 else s.replace(a, b, count)

V = {}
V["no_encode"] = lambda s: rep(rep(rep(rep(s,
    "                        mx_encode32(v, hp, l4, h4, sw_);",
    "                        for (int k = 0; k < 4; ++k) hp[k] = u32x4{__float_as_uint(v[8*k]), __float_as_uint(v[8*k+2]), __float_as_uint(v[8*k+4]), __float_as_uint(v[8*k+6])}; l4 = hp[0]; h4 = hp[1]; sw_ = 0;"),
    "                        __builtin_nontemporal_store(l4, reinterpret_cast<u32x4*>(p.yl4 + rec * 16));", "(void)rec;"),
    "                        __builtin_nontemporal_store(h4, reinterpret_cast<u32x4*>(p.y4 + rec * 16));", ""),
    "                        __builtin_nontemporal_store(sw_, reinterpret_cast<unsigned*>(p.ys + rec * 4));", "")
V["no_mx_mfma"] = lambda s: rep(s, "            for (int jh = 0; jh < 2; ++jh) {", "            for (int jh = 0; jh < 0; ++jh) {")
V["no_side_dma"] = lambda s: rep(rep(s, "                if (j == 0) {\n                    if (i == 4)", "                if (false) {\n                    if (i == 4)"),
                                 "                if (j == 1) {\n                    if (i == 4)", "                if (false) {\n                    if (i == 4)")
V["no_f16_dma"] = lambda s: rep(s, "                if (next) {\n                    if (i == 0) MX_DMA_F16", "                if (false) {\n                    if (i == 0) MX_DMA_F16")
V["no_dma"] = lambda s: V["no_f16_dma"](V["no_side_dma"](s))
V["no_mfma"] = lambda s: rep(V["no_mx_mfma"](s), "                if (live) {\n                    hfrag8 a_nxt", "                if (false) {\n                    hfrag8 a_nxt")
V["no_epilogue"] = lambda s: rep(s, "        for (int pass = 0; pass < 2; ++pass) {           // rows wm*128", "        for (int pass = 0; pass < 0; ++pass) {           // rows wm*128")

V["hot_a"] = lambda s: rep(rep(rep(s, "const int64_t ub = (int64_t)b * p.nch_in * p.T;", "const int64_t ub = 0;"),
                                "        a_row[i] = t0 + row;", "        a_row[i] = row;"),
                            "        int r_ = t0 + rg_ * 64 + lane + off__;", "        int r_ = rg_ * 64 + lane + off__;")
V["no_stores"] = lambda s: rep(rep(rep(rep(rep(s,
    "                        __builtin_nontemporal_store(l4, reinterpret_cast<u32x4*>(p.yl4 + rec * 16));", "                        if (l4.x == 0x12345678u) __builtin_nontemporal_store(l4, reinterpret_cast<u32x4*>(p.yl4 + rec * 16));"),
    "                        __builtin_nontemporal_store(h4, reinterpret_cast<u32x4*>(p.y4 + rec * 16));", "                        if (h4.x == 0x12345678u) __builtin_nontemporal_store(h4, reinterpret_cast<u32x4*>(p.y4 + rec * 16));"),
    "                        __builtin_nontemporal_store(sw_, reinterpret_cast<unsigned*>(p.ys + rec * 4));", "                        if (sw_ == 0x12345678u) __builtin_nontemporal_store(sw_, reinterpret_cast<unsigned*>(p.ys + rec * 4));"),
    "                            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p.yh + (rec0 + m) * 64 + (lane & 3) * 16));", "                            if (v.x == 0x12345678u) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p.yh + (rec0 + m) * 64 + (lane & 3) * 16));"),
    "xxxx", "xxxx") if False else rep(rep(rep(rep(s,
    "                        __builtin_nontemporal_store(l4, reinterpret_cast<u32x4*>(p.yl4 + rec * 16));", "                        if (l4.x == 0x12345678u) __builtin_nontemporal_store(l4, reinterpret_cast<u32x4*>(p.yl4 + rec * 16));"),
    "                        __builtin_nontemporal_store(h4, reinterpret_cast<u32x4*>(p.y4 + rec * 16));", "                        if (h4.x == 0x12345678u) __builtin_nontemporal_store(h4, reinterpret_cast<u32x4*>(p.y4 + rec * 16));"),
    "                        __builtin_nontemporal_store(sw_, reinterpret_cast<unsigned*>(p.ys + rec * 4));", "                        if (sw_ == 0x12345678u) __builtin_nontemporal_store(sw_, reinterpret_cast<unsigned*>(p.ys + rec * 4));"),
    "                            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p.yh + (rec0 + m) * 64 + (lane & 3) * 16));", "                            if (v.x == 0x12345678u) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p.yh + (rec0 + m) * 64 + (lane & 3) * 16));")

V["prio_hi_half"] = lambda s: rep(s, "    for (int ss = 0; ss < p.nss; ++ss) {", "    if (wave >= 4) __builtin_amdgcn_s_setprio(1);\n    for (int ss = 0; ss < p.nss; ++ss) {")
V["prio_lo_half"] = lambda s: rep(s, "    for (int ss = 0; ss < p.nss; ++ss) {", "    if (wave < 4) __builtin_amdgcn_s_setprio(1);\n    for (int ss = 0; ss < p.nss; ++ss) {")

names = sys.argv[1:] or list(V)
objs = [o for o in ("api.o", "frontend.o", "frontend512.o", "vad_cmvn.o", "tdnn_gemm.o", "tdnn_f32.o", "tdnn_bf16.o", "tdnn_split.o", "pool_post.o")]
for n in names:
    path = f"/tmp/tdnn_mx_{n}.hip"
    open(path, "w").write(V[n](src))
    out = os.path.join(ROOT, "kaldi-tflite_amd", "kaldi_tflite_amd", f"libktf_abl_{n}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-I" + CS,
                           "-c", path, "-o", f"/tmp/tdnn_mx_{n}.o"], cwd=CS)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(CS, o) for o in objs]
                          + [f"/tmp/tdnn_mx_{n}.o", "-o", out])
    print("built", out)
