#!/usr/bin/env python3
"""Timing-only ablation builds of csrc/tdnn_mxp.hip (measurement tool; results of most of these builds are WRONG by design). The product
source carries no ablation switches: this script patches a scratch copy and builds libktf_abl_<name>.so beside the product library;
on the GPU box `ABL_ARGS=--mx-persist tools/mx/run_ablations.sh` times every one of them (per-layer GEMM ms of bench.py)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "kaldi-tflite_amd", "csrc")
src = open(os.path.join(CS, "tdnn_mxp.hip")).read()


def rep(s, a, b, count=1):
    assert s.count(a) >= 1, a
    return s.replace(a, b) if count == 0 else s.replace(a, b, count)


V = {}
V["p_base"] = lambda s: s
# operands not exchanged (the epilogues then store transposed garbage): what does the exchange cost in the K-loop?
V["p_noswap"] = lambda s: rep(s, "constexpr bool SWAP = OUT != MX_OUT_STATS;", "constexpr bool SWAP = false;")
# no epilogue work at all (nothing is stored): the K-loop, the hoisted DMAs and the tile switch alone
V["p_noepi"] = lambda s: rep(s, "        if (e_id >= 0) {\n            // ====", "        if (false) {\n            // ====")
# the epilogue's arithmetic without its stores
V["p_nostores"] = lambda s: re.sub(r"__builtin_nontemporal_store\(([^;]+?), (reinterpret_cast<[^;]+)\);", r"if (rows_valid == 0x12345678) __builtin_nontemporal_store(\1, \2);", s)
# no cross-tile stage 0 / no hoisting: every tile starts cold (stage 0 issued at the tile's top, behind the epilogue)
# (not built: needs the prologue back)
# A/B (correct results): ordinary instead of non-temporal plane stores (acknowledged by the L2 instead of by the memory: what K-step 2 of
# the next tile waits for, vmcnt being in order)
V["p_plain"] = lambda s: re.sub(r"__builtin_nontemporal_store\(([^;]+?), (reinterpret_cast<[^;]+)\);", r"*(\2) = \1;", s)
# no DMA at all
V["p_nodma"] = lambda s: rep(s, "__builtin_amdgcn_global_load_lds(", "if (tid == 0x12345) __builtin_amdgcn_global_load_lds(", 0)
# no MFMA
V["p_nomfma"] = lambda s: re.sub(r"acc\[i\]\[jj\] = SWAP \? __builtin_amdgcn_mfma[^;]+;", "acc[i][jj][0] += 1.0f;", s)

names = sys.argv[1:] or list(V)
srcs = re.search(r"^SRCS := (.*)$", open(os.path.join(CS, "Makefile")).read(), re.M).group(1).split()
objs = [f[:-4] + ".o" for f in srcs if f != "tdnn_mxp.hip"]
subprocess.check_call(["make", "-j8"], cwd=CS, stdout=subprocess.DEVNULL)
FLAGS = {"p_iterilp": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"], "p_maxilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
for k in FLAGS:
    V[k] = lambda s: s
for n in names:
    path = f"/tmp/tdnn_mxp_{n}.hip"
    open(path, "w").write(V[n](src))
    out = os.path.join(ROOT, "kaldi-tflite_amd", "kaldi_tflite_amd", f"libktf_abl_{n}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-I" + CS]
                          + FLAGS.get(n, []) + ["-c", path, "-o", f"/tmp/tdnn_mxp_{n}.o"], cwd=CS)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(CS, o) for o in objs]
                          + [f"/tmp/tdnn_mxp_{n}.o", "-o", out])
    print("built", out)
