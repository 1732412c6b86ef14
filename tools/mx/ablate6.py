#!/usr/bin/env python3
"""Round 6: timing-only variants of csrc/tdnn_mx.hip, one feature of the K-loop removed at a time (measurement tool; the results of these
builds are WRONG by design -- run bench.py with --no-parity). Patches a scratch copy, builds _ab/libktf_abl6_<name>.so through
tools/build_variant.sh; tools/ab_libs.sh times them beside the product build (docs/lab_notes_r6.md section 2).

    python tools/mx/ablate6.py [name ...]      (no argument: all)
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "kaldi-tflite_amd", "csrc")
src = open(os.path.join(CS, "tdnn_mx.hip")).read()


def rep(s, a, b, count=1):
    assert s.count(a) >= 1, a
    return s.replace(a, b) if count == 0 else s.replace(a, b, count)


def all_waits_zero(s):     # a variant that issues fewer DMAs must not leave the counted waits counting DMAs that no longer exist
    return rep(s, 'if (j == 1 || j == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");', 'if (false) {}')


V = {}
V["force_edge"] = lambda s: rep(s, "        if (interior) kloop(std::true_type{});", "        if (false) kloop(std::true_type{});")
V["no_side_dma"] = lambda s: all_waits_zero(rep(rep(s, "#define MX_DMA_SA(n_)  ", "#define MX_DMA_SA(n_) {}\n#define MX_DMA_SA_UNUSED(n_)  "),
                                                    "#define MX_DMA_SW(ss_, n_)  ", "#define MX_DMA_SW(ss_, n_) {}\n#define MX_DMA_SW_UNUSED(ss_, n_)  "))
V["no_m_mfma"] = lambda s: rep(s, "                for (int i = 0; i < 8; ++i) {\n                    const u32x4 l = l_n, h = h_n;", "                for (int i = 0; i < 0; ++i) {\n                    const u32x4 l = l_n, h = h_n;")
V["no_a_dma"] = lambda s: rep(s, "if (MX_NEXT()) { MX_DMA_A(ks + 1, j + 1, 0) MX_DMA_A(ks + 1, j + 1, 1) }", "if (false) { MX_DMA_A(ks + 1, j + 1, 0) MX_DMA_A(ks + 1, j + 1, 1) }")
# one A-row DMA per wave and K-step instead of two (rows 128-255 of the stage are never refreshed): what a shared A image per chunk -- the
# three context offsets of a chunk read nearly the same rows -- could give at most on the multi-context layers (tdnn4 / tdnn5 lose a DMA they need)
V["half_a_dma"] = lambda s: rep(s, "if (MX_NEXT()) { MX_DMA_A(ks + 1, j + 1, 0) MX_DMA_A(ks + 1, j + 1, 1) }", "if (MX_NEXT()) { MX_DMA_A(ks + 1, j + 1, 0) }")
V["no_w_dma"] = lambda s: rep(rep(s, "if (i == 0) { if (MX_NEXT()) MX_DMA_W(ks + 1, 0) }", "if (i == 0) { if (false) MX_DMA_W(ks + 1, 0) }"),
                              "if (i == 1) { if (MX_NEXT()) MX_DMA_W(ks + 1, 1) }", "if (i == 1) { if (false) MX_DMA_W(ks + 1, 1) }")
V["no_stage_dma"] = lambda s: V["no_w_dma"](V["no_a_dma"](s))
V["no_dma"] = lambda s: V["no_stage_dma"](V["no_side_dma"](s))
V["no_f16_mfma"] = lambda s: rep(rep(s, "                        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur, bh[0], acc[i][0], 0, 0, 0);", "                        acc[i][0][0] += (float)a_cur[0] + (float)bh[0][0];"),
                                 "for (int jj = 1; jj < 4; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur, bh[jj], acc[i][jj], 0, 0, 0);", "for (int jj = 1; jj < 4; ++jj) acc[i][jj][0] += (float)bh[jj][0];")
# every tile reads (not writes) the rows of the first 8 tiles: no A operand comes from HBM -- the upper bound of any deeper A prefetch
V["hot_a"] = lambda s: rep(s, "    const int R0 = mt * 256;                              // FLAT: first flat row of the tile",
                           "    const int R0 = (mt & 7) * 256;                        // ABLATION: hot rows")
# ... the rows READ only (dense batches): the plane stores go where they belong -- separates the first touch of the A rows from the faster stores
V["hot_reads"] = lambda s: rep(s, "            for (int i = 0; i < 3; ++i) er[i] = flat_row(mr[i]);",
                               "            for (int i = 0; i < 3; ++i) { const int R = (mt & 7) * 256 + mr[i]; const int bb = (int)(__umulhi((unsigned)R, p.t_div_m) >> p.t_div_s); er[i] = i32x4{R, R - bb * len, len, bb}; }")
V["no_table"] = lambda s: rep(s, "                    if (j == 2 && i == 3) {               // the table entries of K-steps 4 ss + 5 .. 4 ss + 8\n                        tb_n = *reinterpret_cast<const i32x4*>(tkb + 4 * ss + 4);\n                        to_n = *reinterpret_cast<const i32x4*>(tko + 4 * ss + 4);",
                              "                    if (j == 2 && i == 3) {\n                        tb_n = i32x4{kb[1], kb[2], kb[3], kb[4]};\n                        to_n = i32x4{ko[1], ko[2], ko[3], ko[4]};")

if __name__ == "__main__":
    names = sys.argv[1:] or list(V)
    for n in names:
        out = os.path.join(CS, f"_abl6_{n}.hip")
        open(out, "w").write(V[n](src))
        try:
            subprocess.run([os.path.join(ROOT, "tools", "build_variant.sh"), f"abl6_{n}", os.path.basename(out)], check=True,
                           env=dict(os.environ, REPLACES="tdnn_mx.hip"))
        finally:
            os.remove(out)
