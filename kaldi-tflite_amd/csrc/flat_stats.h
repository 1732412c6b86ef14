// Fused pooling on flat row tiles, shared by the split-bf16 and the f16mx 256 x 256 kernels (tdnn_split.hip, tdnn_mx.hip): a wave's 128-row
// block of a tile over the batch's valid rows laid end to end holds rows of several utterances, each a run of consecutive rows.
#pragma once
#include "common.h"

// acc[i][j][r]: the FINISHED value (activation, BatchNorm applied) of tile row 128 wm + 16 i + 4 (lane >> 4) + r, column 16 j + (lane & 15) of
// the wave's 64 columns. Per run (wave-uniform loop) the wave sums its columns over the run's rows -- in fp32 relative to the run's first
// row, so that a constant column sums to exactly (n v, n v^2) -- and hands the fp64 result to `out(b, slot, j, s, q)` on lanes 0..15,
// slot = (flat 128-row block of the run) - (flat 128-row block of the utterance's first row): every (utterance, slot) has one writer, an
// utterance of len rows starting at flat row s uses slots 0 .. ((s + len - 1) >> 7) - (s >> 7) (ktf_stats_finalize_flat adds exactly
// those, in order; ktf_flat_stats_slots(T) are allocated per utterance).
// `run_of(m)`: (frame t, utterance length, utterance b) of tile row m (wave-uniform m; valid rows only). R0: first flat row of the tile.
template <typename RunFn, typename OutFn>
__device__ __forceinline__ void flat_stats_runs(f32x4 (&acc)[8][4], int R0, int rows_valid, int wm, int lane, RunFn run_of, OutFn out) {
    const int g4 = lane >> 4;
    const int blk0 = wm * 128;
    const int blk_end = min(blk0 + 128, rows_valid);
    int m = blk0;
    while (m < blk_end) {
        int t_m, len_m, b;
        run_of(m, t_m, len_m, b);
        const int seg_end = min(blk_end, m + (len_m - t_m));
        const int lm = m - blk0, le = seg_end - blk0;                 // the run's rows inside the block: [lm, le)
        const int slot = ((R0 + blk0) >> 7) - ((R0 + m - t_m) >> 7);
        const int rb = g4 * 4 - lm;                                    // lane's row (i, r) relative to the run's first: rb + 16 i + r
        const unsigned span = (unsigned)(le - lm);
        // (lm, le are the same on every lane: as scalars they steer wave-uniform branches)
        const int lmu = __builtin_amdgcn_readfirstlane(lm), leu = __builtin_amdgcn_readfirstlane(le);
        // the run's first row lmu = 16 i0 + 4 g + r0 of the block is held by the lanes of quarter g in acc[i0][.][r0]
        float pv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i == (lmu >> 4)) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) pv[j] = r == (lmu & 3) ? acc[i][j][r] : pv[j];
            }
        const int src = ((lmu >> 2) & 3) * 16 + (lane & 15);          // the lane that holds the run's first row of this column
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[j] = __shfl(pv[j], src, 64);
        float s32[4] = {0.0f, 0.0f, 0.0f, 0.0f}, q32[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        int cnt = 0;
        // per 16-row group of the block (wave-uniform tests): outside the run -- nothing (its terms were + 0.0f: the sums are the same bits);
        // inside -- the plain sums, 12 vector instructions per row of four columns; across an end of the run -- per-row tests, 22 per row.
        // A 998-frame utterance: seven blocks in eight are one run over all eight groups.
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (16 * i + 16 <= lmu || 16 * i >= leu) continue;
            if (16 * i >= lmu && 16 * i + 16 <= leu) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float u = acc[i][j][r] - pv[j];
                        s32[j] += u;
                        q32[j] = fmaf(u, u, q32[j]);
                    }
                cnt += 4;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool in = (unsigned)(rb + 16 * i + r) < span;
                    cnt += in ? 1 : 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float u = in ? acc[i][j][r] - pv[j] : 0.0f;
                        s32[j] += u;
                        q32[j] = fmaf(u, u, q32[j]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double pd = (double)pv[j], sd = (double)s32[j], nd = (double)cnt;
            double s = sd + nd * pd;
            double q = (double)q32[j] + 2.0 * pd * sd + nd * pd * pd;
            s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
            s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
            if (lane < 16) out(b, slot, j, s, q);
        }
        m = seg_end;
    }
}
