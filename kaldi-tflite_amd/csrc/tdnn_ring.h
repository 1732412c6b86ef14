// The 256 x 256 ring tile shared by the 16-bit kernel families: tile constants and the epilogues of the 32x32 and the 16x16
// accumulator layouts (bias -> activation -> BatchNorm affine, then LDS-staged coalesced stores or fused pooling sums).
#pragma once
#include "tdnn_common.h"

#define R_BM 256
#define R_BN 256
#define R_BK 32
#define R_NSTAGE 4
#define R_TILE_BYTES (256 * R_BK * 2)           // 16 KiB per operand
#define R_STAGE_BYTES (2 * R_TILE_BYTES)        // 32 KiB
#define R_EPI_PITCH 260
#define R_LDS_BYTES (R_NSTAGE * R_STAGE_BYTES)  // 131,072 B (epilogue staging needs 64*260*4 = 66,560 B)

// Epilogue shared by the 256x256 kernels: bias -> activation -> BatchNorm affine on the 4x2 accumulator tiles of each wave,
// then either (STATS) fp64 column sums / sums of squares into stats[b][0|1][unit], or four passes of LDS-staged,
// fully coalesced row stores (one 256-column row per wave-instruction).
template <int ACT, bool STATS>
__device__ __forceinline__ void ring_epilogue(f32x16 (&acc)[4][2], const TdnnParams& p, double* __restrict__ stats,
                                              unsigned char* rsm, int b, int t0, int n0, int out_len, int wm, int wn,
                                              int wave, int lane) {
    // ---- epilogue: four passes of 64 staged rows (wave (wm, wn) contributes its 32 x 64 block of pass i)
    float* et = reinterpret_cast<float*>(rsm);
    float bias[2], sc[2], sh[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        const bool nv = n < p.units;
        bias[j] = (nv && p.bias) ? p.bias[n] : 0.0f;
        sc[j] = (nv && p.scale) ? p.scale[n] : 1.0f;
        sh[j] = (nv && p.shift) ? p.shift[n] : 0.0f;
    }
    const int rows_valid = out_len - t0;
    if (STATS) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float v = acc[i][j][r] + bias[j];
                    if (ACT == KTF_ACT_RELU) v = fmaxf(v, 0.0f);
                    else if (ACT != KTF_ACT_NONE) v = apply_act(v, ACT);
                    v = v * sc[j] + sh[j];
                    if (m < rows_valid) {
                        s += (double)v;
                        q += (double)v * (double)v;
                    }
                }
            }
            s += __shfl_xor(s, 32, 64);      // the two half-waves hold the same column
            q += __shfl_xor(q, 32, 64);
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            if (lane < 32 && n < p.units) stats_out(stats, p, b, (t0 >> 7) + wm, n, s, q);
        }
        return;
    }
    const int64_t out_row0 = (int64_t)b * p.Tout + t0;
    const int nl = lane * 4;                      // this lane's 4 columns of the 256-wide staged row
    const int n = n0 + nl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int srow = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[i][j][r] + bias[j];
                if (ACT == KTF_ACT_RELU) v = fmaxf(v, 0.0f);
                else if (ACT != KTF_ACT_NONE) v = apply_act(v, ACT);
                v = v * sc[j] + sh[j];
                et[srow * R_EPI_PITCH + col] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) {
            const int srow = sp * 8 + wave;          // one staged row per wave: 256 contiguous columns
            const int m = (srow >> 5) * 128 + i * 32 + (srow & 31);
            if (m < rows_valid) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(et + srow * R_EPI_PITCH + nl);
                const int64_t off = (out_row0 + m) * p.ldy + n;
                if (n + 4 <= p.units) {
                    if (p.y_dtype == KTF_F32) {
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + off) = v;
                    } else {
                        const unsigned short h0 = f2bf(v.x), h1 = f2bf(v.y), h2 = f2bf(v.z), h3 = f2bf(v.w);
                        uint2 pk;
                        pk.x = (unsigned)h0 | ((unsigned)h1 << 16);
                        pk.y = (unsigned)h2 | ((unsigned)h3 << 16);
                        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.y) + off) = pk;
                        if (p.y_lo) {            // split-bf16 output: the residual plane, the next layer's lo operand
                            uint2 pl;
                            pl.x = (unsigned)f2bf(v.x - bf2f(h0)) | ((unsigned)f2bf(v.y - bf2f(h1)) << 16);
                            pl.y = (unsigned)f2bf(v.z - bf2f(h2)) | ((unsigned)f2bf(v.w - bf2f(h3)) << 16);
                            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.y_lo) + off) = pl;
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e < p.units) {
                            if (p.y_dtype == KTF_F32) {
                                reinterpret_cast<float*>(p.y)[off + e] = v[e];
                            } else {
                                const unsigned short h = f2bf(v[e]);
                                reinterpret_cast<unsigned short*>(p.y)[off + e] = h;
                                if (p.y_lo) reinterpret_cast<unsigned short*>(p.y_lo)[off + e] = f2bf(v[e] - bf2f(h));
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

// STATS: instead of storing y, the epilogue adds every column's sum and sum of squares over the tile's valid rows (fp64)
// into stats[b][0|1][unit] — statistics pooling fused into the producing GEMM, the (B,T,units) activation never exists.
// ------------------------------------------------------------------------------------ BF16, 256x256 tile, 16x16x32 MFMA
// Same ring / DMA / tile order as tdnn_bf16r_kernel, but the wave's 128 x 64 block is 8 x 4 tiles of
// v_mfma_f32_16x16x32_bf16: one MFMA consumes the whole 32-deep K-step, and the chip holds a higher clock on this
// shape under load (MI355X_MICROARCH.md, DVFS item 7). Fragment lane map: row = lane&15, 16-B chunk = lane>>4, so the
// conflict-free chunk permutation is c ^ ((4 - (row>>2)) & 3) (each ds_read_b128 lane group then covers all 16 slots).
typedef __attribute__((ext_vector_type(4))) float f32x4v;
typedef __attribute__((ext_vector_type(8))) _Float16 hfrag8;
__device__ __forceinline__ f32x4v mfma16x16x32(const bfrag8& a, const bfrag8& b, const f32x4v& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// per-lane epilogue constants of the 16x16 accumulator layout: bias / BatchNorm scale / shift of the lane's four columns
struct Epi16Prm { float bias[4], sc[4], sh[4]; };
__device__ __forceinline__ Epi16Prm epi16_load(const TdnnParams& p, int n0, int wn, int lane) {
    Epi16Prm e;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + (lane & 15);
        const bool nv = n < p.units;
        e.bias[j] = (nv && p.bias) ? p.bias[n] : 0.0f;
        e.sc[j] = (nv && p.scale) ? p.scale[n] : 1.0f;
        e.sh[j] = (nv && p.shift) ? p.shift[n] : 0.0f;
    }
    return e;
}

// 16-byte store of a piece of a 16-bit activation plane. The plane (1 GB per layer at 1024 utterances) is read by the NEXT
// launch only: written non-temporally it does not push the weights and the activation tiles two workgroups share out of the
// XCD's L2 (+1.2 % on the whole step; non-temporal operand LOADS cost 3-6 %).
__device__ __forceinline__ void st16(u32x4* dst, const u32x4& v) {
    if (KTF_X3_Y_NT) __builtin_nontemporal_store(v, dst);
    else *dst = v;
}

// FLAT (tdnn_x3s_kernel's flat row tiling; row-major outputs only): tile row m is output row rowmap[m] of the (B * Tout)-row output
// (an LDS table behind the staging image), t0 = 0 and out_len = the tile's valid rows.
template <int ACT, bool STATS, bool FLAT = false>
__device__ __forceinline__ void ring_epilogue16(f32x4v (&acc)[8][4], const TdnnParams& p, double* __restrict__ stats,
                                                unsigned char* rsm, int b, int t0, int n0, int out_len, int wm, int wn,
                                                int wave, int lane, const Epi16Prm& prm, const int* rowmap = nullptr) {
    float* et = reinterpret_cast<float*>(rsm);
    const float (&bias)[4] = prm.bias;
    const float (&sc)[4] = prm.sc;
    const float (&sh)[4] = prm.sh;
    const int rows_valid = out_len - t0;
    if (STATS) {
        // A lane holds 32 rows of each of its 4 columns. Their sum and sum of squares are taken in fp32 RELATIVE TO A PIVOT
        // (row 0 of the wave's 128-row block: a constant column -- a dead ReLU unit -- gives exactly 0 and 0, not fp32
        // cancellation noise) and only the per-lane results go to fp64: 32 x 3 fp32 operations per column instead of 32 x 3
        // fp64 ones (the fp64 form was 4.3 us per tile, a fifth of a K = 512 tile's K-loop).
        const int rv = rows_valid - wm * 128;                  // valid rows of this wave's block (may be <= 0)
        const int g4 = lane >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v0 = acc[0][j][0] + bias[j];
            if (ACT == KTF_ACT_RELU) v0 = fmaxf(v0, 0.0f);
            else if (ACT != KTF_ACT_NONE) v0 = apply_act(v0, ACT);
            v0 = v0 * sc[j] + sh[j];
            const float pv = __shfl(v0, lane & 15, 64);          // row 0 of the block lives in the g4 == 0 lane of this column
            float s32 = 0.0f, q32 = 0.0f;
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r] + bias[j];
                    if (ACT == KTF_ACT_RELU) v = fmaxf(v, 0.0f);
                    else if (ACT != KTF_ACT_NONE) v = apply_act(v, ACT);
                    v = v * sc[j] + sh[j];
                    if (rv >= 128 || i * 16 + g4 * 4 + r < rv) {       // first term wave-uniform: full blocks carry no row predicate
                        const float u = v - pv;
                        s32 += u;
                        q32 = fmaf(u, u, q32);
                        ++cnt;
                    }
                }
            }
            const double pd = (double)pv, sd = (double)s32, nd = (double)cnt;
            double s = sd + nd * pd;
            double q = (double)q32 + 2.0 * pd * sd + nd * pd * pd;
            s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);      // the four 16-lane groups hold the same column
            s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
            const int n = n0 + wn * 64 + j * 16 + (lane & 15);
            if (lane < 16 && n < p.units) stats_out(stats, p, b, (t0 >> 7) + wm, n, s, q);
        }
        return;
    }
    const int64_t out_row0 = (int64_t)b * p.Tout + t0;
    const int nl = lane * 4;
    const int n = n0 + nl;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {          // rows wm*128 + pass*32 .. +31 of both wave rows -> 64 staged rows
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
            const int i = pass * 2 + ih;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = wn * 64 + j * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int srow = wm * 32 + ih * 16 + (lane >> 4) * 4 + r;
                    float v = acc[i][j][r] + bias[j];
                    if (ACT == KTF_ACT_RELU) v = fmaxf(v, 0.0f);
                    else if (ACT != KTF_ACT_NONE) v = apply_act(v, ACT);
                    v = v * sc[j] + sh[j];
                    et[srow * R_EPI_PITCH + col] = v;
                }
            }
        }
        __syncthreads();
        if (p.y_dtype != KTF_F32) {
            // bf16 output: 16-byte stores (8 columns per lane, two staged rows per wave instruction)
            const int n8 = n0 + (lane & 31) * 8;
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                const int srow = sp * 16 + wave * 2 + (lane >> 5);
                const int m = (srow >> 5) * 128 + pass * 32 + (srow & 31);
                if (m < rows_valid) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(et + srow * R_EPI_PITCH + (lane & 31) * 8);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(et + srow * R_EPI_PITCH + (lane & 31) * 8 + 4);
                    const int64_t off = (FLAT ? (int64_t)rowmap[m] : out_row0 + m) * p.ldy + n8;
                    unsigned short* yp = reinterpret_cast<unsigned short*>(p.y) + off;
                    const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    unsigned short hh[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) hh[e] = f2bf(vv[e]);
                    if (n8 + 8 <= p.units) {
                        u32x4 pk;
                        pk.x = (unsigned)hh[0] | ((unsigned)hh[1] << 16);
                        pk.y = (unsigned)hh[2] | ((unsigned)hh[3] << 16);
                        pk.z = (unsigned)hh[4] | ((unsigned)hh[5] << 16);
                        pk.w = (unsigned)hh[6] | ((unsigned)hh[7] << 16);
                        st16(reinterpret_cast<u32x4*>(yp), pk);
                        if (p.y_lo) {            // split-bf16 output: the residual plane, the next layer's lo operand
                            u32x4 pl;
                            pl.x = (unsigned)f2bf(vv[0] - bf2f(hh[0])) | ((unsigned)f2bf(vv[1] - bf2f(hh[1])) << 16);
                            pl.y = (unsigned)f2bf(vv[2] - bf2f(hh[2])) | ((unsigned)f2bf(vv[3] - bf2f(hh[3])) << 16);
                            pl.z = (unsigned)f2bf(vv[4] - bf2f(hh[4])) | ((unsigned)f2bf(vv[5] - bf2f(hh[5])) << 16);
                            pl.w = (unsigned)f2bf(vv[6] - bf2f(hh[6])) | ((unsigned)f2bf(vv[7] - bf2f(hh[7])) << 16);
                            st16(reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(p.y_lo) + off), pl);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (n8 + e < p.units) {
                                yp[e] = hh[e];
                                if (p.y_lo) reinterpret_cast<unsigned short*>(p.y_lo)[off + e] = f2bf(vv[e] - bf2f(hh[e]));
                            }
                    }
                }
            }
        } else
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) {
            const int srow = sp * 8 + wave;
            const int m = (srow >> 5) * 128 + pass * 32 + (srow & 31);
            if (m < rows_valid) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(et + srow * R_EPI_PITCH + nl);
                const int64_t off = (FLAT ? (int64_t)rowmap[m] : out_row0 + m) * p.ldy + n;
                if (n + 4 <= p.units) {
                    if (p.y_dtype == KTF_F32) {
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + off) = v;
                    } else {
                        uint2 pk;
                        pk.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
                        pk.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
                        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.y) + off) = pk;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e < p.units) {
                            if (p.y_dtype == KTF_F32) reinterpret_cast<float*>(p.y)[off + e] = v[e];
                            else reinterpret_cast<unsigned short*>(p.y)[off + e] = f2bf(v[e]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

