// lfss_mfma.hip.h - the LFSSBlock glue of lfss.hip.h for C = 32 (D = 64) with the 1x1 projections on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32).
//
// The thread-per-position kernels of lfss.hip.h stream their weights through SGPRs: 4096 FMAs per position need
// ~480 s_load per wave, and PMC shows the waves waiting on them (VALU issue 28 % of a wave's life at two waves per
// SIMD, `SQ_WAIT_INST_ANY` 31 %).  Here the weight matrices sit in LDS as MFMA A operands in fetch order (in 64 VGPRs
// in the first version: two waves per SIMD instead of four), a wave owns `gpw` groups of 64 positions, and a group
// is two 32-column MFMA tiles:
//   * thread-per-position phase (LayerNorm over channels in registers, gates), then one v_permlane32_swap per
//     channel pair turns [positions 0-63 of channel 2j], [.. of 2j+1] into the B operands of both tiles;
//   * a 32x32 accumulator tile holds, per lane, position (lane & 31) and rows 8g + 4(lane >> 5) + i: 16 of the 32
//     channels, the other 16 in lane ^ 32 - LayerNorm statistics are an in-lane sum and one cross-half exchange;
//   * the accumulator registers are used AS the next projection's B operands without any movement: register j
//     holds channel (j&3) + 8(j>>2) in lanes 0-31 and that + 4 in lanes 32-63, so the A operand of step j simply
//     carries those two weight columns (the K order of a GEMM is free).
// ln_2's affine is folded into conv1 (W1' = W1 diag(w), b1' = b1 + W1 b): exact algebra, fp32 rounding differs at
// 1e-7.  fp32 MFMA keeps fp32 products (no operand splitting).  The kernels are latency-bound rather than HBM-bound
// (DESIGN.md 4): loads are issued in explicit batches ahead of their use and ahead of the group's stores.
// Reference: basicsr/archs/wavemamba_arch.py :491-494 (SS2D tail), :525-526 (LFSSBlock), :226-230 (ffn).
#pragma once
#include <hip/hip_runtime.h>
#include "haar.hip.h"          // bf16_t, ld1 / st1 (fp32 and bf16 overloads)

namespace wm {

typedef float lfss_v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float silu_fast(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}
__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32); }

// waves each take `gpw` consecutive groups of 64 positions; chosen so that the waves fill whole rounds of the
// `slots` resident waves (1024 SIMDs x the kernel's waves per SIMD), at most 8 groups per wave
inline int lfss_groups_per_wave(long long ngroups, int slots) {
    const long long rounds = (ngroups + 8LL * slots - 1) / (8LL * slots);
    long long gpw = (ngroups + rounds * slots - 1) / (rounds * slots);
    return (int)(gpw < 1 ? 1 : gpw);
}

// A 32-position tile of a (B, L, 32) token array or a (B, 32, L) plane stack in accumulator layout: register 4g + i of
// lane (n, h) = channel 8g + 4h + i of position pq (16-byte accesses on tokens, 128-byte runs per half-wave on planes).
__device__ __forceinline__ void load_tile32(const float* __restrict__ src, bool nchw, long long b, long long pq,
                                            long long L, int h, float (&v)[16]) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int row0 = 8 * gq + 4 * h;
        if (nchw) {
            const float* tp = src + (b * 32 + row0) * L + pq;
            v[4 * gq] = tp[0]; v[4 * gq + 1] = tp[L]; v[4 * gq + 2] = tp[2 * L]; v[4 * gq + 3] = tp[3 * L];
        } else {
            const float4 t = *reinterpret_cast<const float4*>(src + (b * L + pq) * 32 + row0);
            v[4 * gq] = t.x; v[4 * gq + 1] = t.y; v[4 * gq + 2] = t.z; v[4 * gq + 3] = t.w;
        }
    }
}
__device__ __forceinline__ void store_tile32(float* __restrict__ dst, bool nchw, long long b, long long pos,
                                             long long L, int h, const float (&v)[16]) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int row0 = 8 * gq + 4 * h;
        if (nchw) {
            float* tp = dst + (b * 32 + row0) * L + pos;
            tp[0] = v[4 * gq]; tp[L] = v[4 * gq + 1]; tp[2 * L] = v[4 * gq + 2]; tp[3 * L] = v[4 * gq + 3];
        } else {
            *reinterpret_cast<float4*>(dst + (b * L + pos) * 32 + row0) =
                make_float4(v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]);
        }
    }
}
// (v - mean) * rstd over the 32 channels of a position held as 16 + 16 registers in lanes n and n + 32
__device__ __forceinline__ void tile_normalise(float (&v)[16], float eps) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    const float m = xhalf_sum(s) * (1.0f / 32);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] -= m; q = fmaf(v[i], v[i], q); }
    const float r = rsqrtf(xhalf_sum(q) * (1.0f / 32) + eps);
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] *= r;
}

// A operands live in LDS in fetch order: operand j of lane l at s[((j >> 2) * 64 + l) * 4 + (j & 3)] - one
// conflict-free ds_read_b128 per four MFMAs (256 B per MFMA against 64 cycles of matrix-core time).
__device__ __forceinline__ int aop_slot(int j, int l) { return ((j >> 2) * 64 + l) * 4 + (j & 3); }
// channel held by accumulator register j in lanes of half h
__device__ __forceinline__ int acc_chan(int j, int h) { return (j & 3) + 8 * (j >> 2) + 4 * h; }

// waves per SIMD the middle kernel is compiled for.  Four (128 registers) spilled 46 vector registers of the ny = 4 fp32
// form to scratch; three (163 registers, no spills) is 5 % faster on the kernel (6.92 -> 6.60 ms per UHD step for the
// block glue, tools: build_variant.sh mid3, bench.py twice each, gpurun_out r3z).
#ifndef WM_LFSS_MID_WAVES
#define WM_LFSS_MID_WAVES 3
#endif
#ifndef WM_LFSS_IN_WAVES
#define WM_LFSS_IN_WAVES 4
#endif

// two tiles' accumulators of one output row block -> 256-byte runs: after the swap, register i of `lo` is channel
// row(i) at positions p0 .. p0 + 63 and register i of `hi` is channel row(i) + 4
template <typename TP>
__device__ __forceinline__ void store_rows64(TP* __restrict__ plane0 /* channel 0 of the 32-row block, + p0 + lane */,
                                             long long L, bool ok, lfss_v16f lo, lfss_v16f hi) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo[i]), __float_as_uint(hi[i]), false, false);
        if (ok) {
            const long long ro = (long long)(8 * (i >> 2) + (i & 3)) * L;
            st1(plane0 + ro, __uint_as_float(r[0]));
            st1(plane0 + ro + 4 * L, __uint_as_float(r[1]));
        }
    }
}

// ---- lfss_mid: ysum, z, tok -> tok1 (B, L, C), f (B, D, L) -------------------------------------------
// NY = 1: merged core output; NY = 4: the four directions' outputs, added here (:490).  TP: storage type of the y / z / f
// planes (float, or bf16_t in the bf16-storage mode)
// RZ: the gate z = in_proj(ln_1(tok))[D:2D] (:485-486) is RECOMPUTED here from the token tile this kernel reads anyway for
// the skip connection, instead of being written by lfss_in and read back (512 B per position of the block's 3456): the same
// A operands (W_in rows D.. x ln_1.weight), the same normalised tile as B operands, the same matrix-instruction order and bias
// start as lfss_in_mfma_kernel - bit-identical z in fp32 storage - with the output ROWS permuted so that accumulator
// register i of row block mt IS channel 2 (16 mt + i) + h: the layout the out_norm'ed y has after its half-wave swap.
// (the recomputing form holds the normalised token tile and the gate accumulators beside y: at three waves per SIMD it spills 19
// registers; at two - 256 registers, none spilled - it is 3-6 % faster per call, tools/bench_lfss_rz.py with -DWM_LFSS_MID_WAVES=2;
// the reading form is 25 % slower at two)
#ifndef WM_LFSS_MID_RZ_WAVES
#define WM_LFSS_MID_RZ_WAVES 2
#endif
template <int NY, typename TP = float, bool RZ = false>
__global__ __launch_bounds__(256, RZ ? WM_LFSS_MID_RZ_WAVES : WM_LFSS_MID_WAVES) void lfss_mid_mfma_kernel(
    const TP* __restrict__ ysum, long long ystride, const TP* __restrict__ z, const float* __restrict__ tok, int tok_nchw,
    const float* __restrict__ on_w, const float* __restrict__ on_b, float on_eps,
    const float* __restrict__ W_out /*(C, D)*/, const float* __restrict__ skip1,
    const float* __restrict__ ln2_w, const float* __restrict__ ln2_b, float ln2_eps,
    const float* __restrict__ W1 /*(D, C)*/, const float* __restrict__ b1,
    float* __restrict__ tok1, TP* __restrict__ f, int B, long long L, int ngl, long long ngroups, int gpw,
    const float* __restrict__ ln1_w = nullptr, const float* __restrict__ ln1_b = nullptr, float ln1_eps = 0.0f,
    const float* __restrict__ W_in = nullptr /*(2D, C)*/) {
    constexpr int C = 32, D = 64;
    __shared__ __attribute__((aligned(16))) float s_skip[C];
    __shared__ __attribute__((aligned(16))) float s_b1[D];
    __shared__ __attribute__((aligned(16))) float s_Aout[(D / 2) * 64];          // 32 operands x 64 lanes
    __shared__ __attribute__((aligned(16))) float s_A1[2 * (C / 2) * 64];        // 2 row blocks x 16 operands
    __shared__ __attribute__((aligned(16))) float s_Az[RZ ? 2 * (C / 2) * 64 : 4];   // RZ: the gate's 2 row blocks x 16 operands
    __shared__ __attribute__((aligned(16))) float s_bz[RZ ? D : 4];             // RZ: [half h][row block][register]
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (RZ) {
        for (int e = threadIdx.x; e < 2 * (C / 2) * 64; e += 256) {
            const int mt = e >> 10, j = (e >> 6) & 15, l = e & 63;
            const int k = acc_chan(j, l >> 5), r = l & 31;
            const int c = 2 * (16 * mt + (r & 3) + 4 * (r >> 3)) + ((r >> 2) & 1);   // the gate channel MFMA row r of block mt computes
            s_Az[aop_slot(mt * 16 + j, l)] = W_in[(D + c) * C + k] * ln1_w[k];
        }
        if (threadIdx.x >= 128 && threadIdx.x < 128 + D) {
            const int m = threadIdx.x - 128, hh = m >> 5, mt = (m >> 4) & 1, i = m & 15;
            const int c = 2 * (16 * mt + i) + hh;
            float acc = 0.0f;
            for (int k = 0; k < C; ++k) acc = fmaf(W_in[(D + c) * C + k], ln1_b[k], acc);
            s_bz[m] = acc;
        }
    }
    if (threadIdx.x < C) s_skip[threadIdx.x] = skip1[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + D) {
        const int m = threadIdx.x - 64;
        float acc = b1[m];
        for (int k = 0; k < C; ++k) acc = fmaf(W1[m * C + k], ln2_b[k], acc);
        s_b1[m] = acc;
    }
    for (int e = threadIdx.x; e < (D / 2) * 64; e += 256) {
        const int j = e >> 6, l = e & 63;
        s_Aout[aop_slot(j, l)] = W_out[(l & 31) * D + 2 * j + (l >> 5)];
    }
    for (int e = threadIdx.x; e < 2 * (C / 2) * 64; e += 256) {
        const int mt = e >> 10, j = (e >> 6) & 15, l = e & 63;
        const int k = acc_chan(j, l >> 5);
        s_A1[aop_slot(mt * 16 + j, l)] = W1[(32 * mt + (l & 31)) * C + k] * ln2_w[k];
    }
    __syncthreads();

    const long long g0 = ((long long)blockIdx.x * 4 + wv) * gpw;
    for (int gi = 0; gi < gpw; ++gi) {
        const long long g = g0 + gi;
        if (g >= ngroups) break;
        const long long b = g / ngl;
        const long long p0 = (g - b * ngl) * 64;
        // ---- thread-per-position: out_norm, gate ----
        const bool okl = p0 + lane < L;
        const long long pc = min(p0 + lane, L - 1);
        const TP* yp = ysum + b * D * L + pc;
        const TP* zp = z + b * D * L + pc;
        float y[D];
        if constexpr (NY == 1) {
#pragma unroll
            for (int d = 0; d < D; ++d) y[d] = ld1(yp + (long long)d * L);
        } else {
            // the four directions' outputs, added in the reference's order y1 + y2 + y3 + y4 (:490) =
            // [row fwd] + [row rev] + [col fwd] + [col rev]; explicit batches of 4 channels x 4 buffers in flight (8 x 4 spills at 128 registers)
            // channels per batch of y loads (x 4 direction buffers in flight per lane).  The recomputing form runs two waves per SIMD
            // with registers to spare: 16 channels = 64 loads in flight per lane, 0.71 -> 0.65 ms at UHD level 1 (4.9 TB/s on its
            // 1536 B per position); 32 gains nothing more (tools/bench_lfss_rz.py with -DWM_LFSS_MID_YB_RZ=..)
#ifndef WM_LFSS_MID_YB_RZ
#define WM_LFSS_MID_YB_RZ 16
#endif
            // NY = 2 (paired core output): [row fwd + row rev] + [col fwd + col rev], twice the channels per batch = the same loads in flight
            constexpr int YB = (RZ ? WM_LFSS_MID_YB_RZ : 4) * (NY == 2 ? 2 : 1);
#pragma unroll
            for (int d0 = 0; d0 < D; d0 += YB) {
                float t[NY][YB];
#pragma unroll
                for (int q = 0; q < NY; ++q)
#pragma unroll
                    for (int i = 0; i < YB; ++i) t[q][i] = ld1(yp + q * ystride + (long long)(d0 + i) * L);
#pragma unroll
                for (int i = 0; i < YB; ++i) {
                    if constexpr (NY == 2) y[d0 + i] = t[0][i] + t[1][i];
                    else y[d0 + i] = ((t[0][i] + t[1][i]) + t[2][i]) + t[3][i];
                }
                __builtin_amdgcn_sched_barrier(0);       // one batch of loads in flight, not all 256 (spills)
            }
        }
        float mean = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) mean += y[d];
        mean *= (1.0f / D);
        float var = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) { const float q = y[d] - mean; var = fmaf(q, q, var); }
        const float rstd = rsqrtf(var * (1.0f / D) + on_eps);
        if constexpr (RZ) {
#pragma unroll
            for (int d0 = 0; d0 < D; d0 += 8) {          // (eight channels' scalar weights at a time: all 128 at once spill)
#pragma unroll
                for (int i = 0; i < 8; ++i) y[d0 + i] = fmaf((y[d0 + i] - mean) * rstd, on_w[d0 + i], on_b[d0 + i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        // z in explicit double-buffered batches: left to itself the compiler issues the 64 loads one at a time, each
        // followed by its wait (64 serialised round trips per group)
        constexpr int ZB = 8;
        float zb[2][ZB];
#pragma unroll
        for (int i = 0; i < ZB; ++i) zb[0][i] = ld1(zp + (long long)i * L);
#pragma unroll
        for (int d0 = 0; d0 < D; d0 += ZB) {
            const int cur = (d0 / ZB) & 1;
            if (d0 + ZB < D) {
#pragma unroll
                for (int i = 0; i < ZB; ++i) zb[cur ^ 1][i] = ld1(zp + (long long)(d0 + ZB + i) * L);
            }
#pragma unroll
            for (int i = 0; i < ZB; ++i)
                y[d0 + i] = fmaf((y[d0 + i] - mean) * rstd, on_w[d0 + i], on_b[d0 + i]) * silu_fast(zb[cur][i]);
        }
        }
        // ---- B operands of the two tiles ----
#pragma unroll
        for (int j = 0; j < D / 2; ++j) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(y[2 * j]), __float_as_uint(y[2 * j + 1]),
                                                            false, false);
            y[2 * j] = __uint_as_float(r[0]);           // tile 0: positions p0 + n,      channels 2j + h
            y[2 * j + 1] = __uint_as_float(r[1]);       // tile 1: positions p0 + 32 + n, channels 2j + h
        }
        if constexpr (RZ) {
            // ---- the gate, recomputed in the layout y now has: y[2 j + t] *= silu(z[channel 2 j + h] at tile t's position).
            // One tile at a time (16 + 16 live registers beside y's 64; both tiles at once spilled 22)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                __builtin_amdgcn_sched_barrier(0);       // the tile's loads stay here (hoisted above out_norm they spill)
                float nt[16];
                load_tile32(tok, tok_nchw != 0, b, min(p0 + 32 * t + n, L - 1), L, h, nt);
                tile_normalise(nt, ln1_eps);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    lfss_v16f za;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const float4 bb = *reinterpret_cast<const float4*>(&s_bz[(h * 2 + mt) * 16 + 4 * gq]);
                        za[4 * gq] = bb.x; za[4 * gq + 1] = bb.y; za[4 * gq + 2] = bb.z; za[4 * gq + 3] = bb.w;
                    }
#pragma unroll
                    for (int j4 = 0; j4 < C / 8; ++j4) {
                        const float4 a4 = *reinterpret_cast<const float4*>(&s_Az[((mt * 4 + j4) * 64 + lane) * 4]);
                        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
                            za = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], nt[4 * j4 + jj], za, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) y[2 * (16 * mt + i) + t] *= silu_fast(za[i]);
                }
            }
        }
        lfss_v16f acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
#pragma unroll
        for (int j4 = 0; j4 < D / 8; ++j4) {
            const float4 a4 = *reinterpret_cast<const float4*>(&s_Aout[(j4 * 64 + lane) * 4]);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], y[2 * (4 * j4 + jj)], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], y[2 * (4 * j4 + jj) + 1], acc[1], 0, 0, 0);
            }
        }
        float tt[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long long pos = p0 + 32 * t + n;
            const long long pq = min(pos, L - 1);
            load_tile32(tok, tok_nchw != 0, b, pq, L, h, tt[t]);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 sk = *reinterpret_cast<const float4*>(&s_skip[8 * gq + 4 * h]);
                tt[t][4 * gq] = fmaf(tt[t][4 * gq], sk.x, acc[t][4 * gq]);
                tt[t][4 * gq + 1] = fmaf(tt[t][4 * gq + 1], sk.y, acc[t][4 * gq + 1]);
                tt[t][4 * gq + 2] = fmaf(tt[t][4 * gq + 2], sk.z, acc[t][4 * gq + 2]);
                tt[t][4 * gq + 3] = fmaf(tt[t][4 * gq + 3], sk.w, acc[t][4 * gq + 3]);
            }
            if (pos < L) store_tile32(tok1, false, b, pos, L, h, tt[t]);
            tile_normalise(tt[t], ln2_eps);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            lfss_v16f a[2];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 bb = *reinterpret_cast<const float4*>(&s_b1[32 * mt + 8 * gq + 4 * h]);
                a[0][4 * gq] = bb.x; a[0][4 * gq + 1] = bb.y; a[0][4 * gq + 2] = bb.z; a[0][4 * gq + 3] = bb.w;
            }
            a[1] = a[0];
#pragma unroll
            for (int j4 = 0; j4 < C / 8; ++j4) {
                const float4 a4 = *reinterpret_cast<const float4*>(&s_A1[((mt * 4 + j4) * 64 + lane) * 4]);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    a[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], tt[0][4 * j4 + jj], a[0], 0, 0, 0);
                    a[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], tt[1][4 * j4 + jj], a[1], 0, 0, 0);
                }
            }
            store_rows64(f + (b * D + 32 * mt) * L + p0 + lane, L, okl, a[0], a[1]);
        }
    }
}

// ---- lfss_in: tok -> x (B, D, L), z (B, D, L) ------------------------------------------------------
// ln_1's affine is folded into in_proj (W' = W diag(w), bias' = W b).
template <typename TP = float>
__global__ __launch_bounds__(256, WM_LFSS_IN_WAVES) void lfss_in_mfma_kernel(
    const float* __restrict__ tok, int tok_nchw, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
    const float* __restrict__ W_in /*(2D, C)*/, TP* __restrict__ x, TP* __restrict__ z, int B, long long L, int ngl,
    long long ngroups, int gpw) {
    constexpr int C = 32, D = 64;
    __shared__ __attribute__((aligned(16))) float s_bias[2 * D];
    __shared__ __attribute__((aligned(16))) float s_A[4 * (C / 2) * 64];         // 4 row blocks x 16 operands
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 2 * D) {
        float acc = 0.0f;
        for (int k = 0; k < C; ++k) acc = fmaf(W_in[threadIdx.x * C + k], ln_b[k], acc);
        s_bias[threadIdx.x] = acc;
    }
    for (int e = threadIdx.x; e < 4 * (C / 2) * 64; e += 256) {
        const int mt = e >> 10, j = (e >> 6) & 15, l = e & 63;
        const int k = acc_chan(j, l >> 5);
        s_A[aop_slot(mt * 16 + j, l)] = W_in[(32 * mt + (l & 31)) * C + k] * ln_w[k];
    }
    __syncthreads();
    const long long g0 = (long long)blockIdx.x * 4 * gpw + wv;     // the block's waves walk adjacent groups together
    float nx[2][16];
    auto load_tok = [&](long long g) {
        const long long b = g / ngl;
        const long long p0 = (g - b * ngl) * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t) load_tile32(tok, tok_nchw != 0, b, min(p0 + 32 * t + n, L - 1), L, h, nx[t]);
    };
    if (g0 < ngroups) load_tok(g0);
    for (int gi = 0; gi < gpw; ++gi) {
        const long long g = g0 + 4 * gi;
        if (g >= ngroups) break;
        const long long b = g / ngl;
        const long long p0 = (g - b * ngl) * 64;
        const bool okl = p0 + lane < L;
        float a[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[t][i] = nx[t][i];
            tile_normalise(a[t], eps);
        }
        if (gi + 1 < gpw && g + 4 < ngroups) load_tok(g + 4);       // ahead of this group's 128 stores
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (mt >= 2 && !z) break;                    // z == nullptr: the consumer recomputes the gate (lfss_mid, RZ)
            lfss_v16f acc[2];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 bb = *reinterpret_cast<const float4*>(&s_bias[32 * mt + 8 * gq + 4 * h]);
                acc[0][4 * gq] = bb.x; acc[0][4 * gq + 1] = bb.y; acc[0][4 * gq + 2] = bb.z; acc[0][4 * gq + 3] = bb.w;
            }
            acc[1] = acc[0];
#pragma unroll
            for (int j4 = 0; j4 < C / 8; ++j4) {
                const float4 a4 = *reinterpret_cast<const float4*>(&s_A[((mt * 4 + j4) * 64 + lane) * 4]);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], a[0][4 * j4 + jj], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], a[1][4 * j4 + jj], acc[1], 0, 0, 0);
                }
            }
            TP* dp = (mt < 2 ? x + (b * D + 32 * mt) * L : z + (b * D + 32 * (mt - 2)) * L) + p0 + lane;
            store_rows64(dp, L, okl, acc[0], acc[1]);
        }
    }
}

// ---- lfss_out: fc (B, D, L), tok1 -> tok2 --------------------------------------------------------------
template <typename TP = float>
__global__ __launch_bounds__(256, 2) void lfss_out_mfma_kernel(const TP* __restrict__ fc, const float* __restrict__ tok1,
                                                              const float* __restrict__ W3 /*(C, C)*/,
                                                              const float* __restrict__ b3, const float* __restrict__ skip2,
                                                              float* __restrict__ out, int out_nchw, int B, long long L,
                                                              int ngl, long long ngroups, int gpw) {
    constexpr int C = 32, D = 64;
    __shared__ __attribute__((aligned(16))) float s_b3[C];
    __shared__ __attribute__((aligned(16))) float s_skip[C];
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < C) { s_b3[threadIdx.x] = b3[threadIdx.x]; s_skip[threadIdx.x] = skip2[threadIdx.x]; }
    float A[C / 2];
#pragma unroll
    for (int j = 0; j < C / 2; ++j) A[j] = W3[n * C + (j & 3) + 8 * (j >> 2) + 4 * h];
    __syncthreads();
    float sk[16];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const float4 s4 = *reinterpret_cast<const float4*>(&s_skip[8 * gq + 4 * h]);
        sk[4 * gq] = s4.x; sk[4 * gq + 1] = s4.y; sk[4 * gq + 2] = s4.z; sk[4 * gq + 3] = s4.w;
    }
    const long long g0 = (long long)blockIdx.x * 4 * gpw + wv;     // the block's waves walk adjacent groups together
    for (int gi = 0; gi < gpw; ++gi) {
        const long long g = g0 + 4 * gi;
        if (g >= ngroups) break;
        const long long b = g / ngl;
        const long long p0 = (g - b * ngl) * 64;
        float gate[2][16], val[2][16], tk[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long long pq = min(p0 + 32 * t + n, L - 1);
            const TP* fp = fc + (b * D + 4 * h) * L + pq;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long long ro = (long long)(8 * (i >> 2) + (i & 3)) * L;
                gate[t][i] = ld1(fp + ro);
                val[t][i] = ld1(fp + ro + 32 * L);
            }
            load_tile32(tok1, false, b, pq, L, h, tk[t]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long long pos = p0 + 32 * t + n;
            lfss_v16f acc;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 bb = *reinterpret_cast<const float4*>(&s_b3[8 * gq + 4 * h]);
                acc[4 * gq] = bb.x; acc[4 * gq + 1] = bb.y; acc[4 * gq + 2] = bb.z; acc[4 * gq + 3] = bb.w;
            }
#pragma unroll
            for (int j = 0; j < C / 2; ++j)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j], gelu_erf(gate[t][j]) * val[t][j], acc, 0, 0, 0);
            float o[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] = fmaf(tk[t][i], sk[i], acc[i]);
            if (pos < L) store_tile32(out, out_nchw != 0, b, pos, L, h, o);
        }
    }
}

// ---- lfss_out with the gated ffn's depth-wise 3x3 folded in: f (B, D, H, W), tok1 -> tok2 -------------------------
// SURVEY.md 8f rank 2 / reference :226-230: fc = dwconv3x3(f) + bias;  gelu(fc[:C]) * fc[C:] -> conv3 -> * skip.  The
// unfused pair wrote fc (256 B per position) and read it back in the next launch.  Here a wave takes 64 consecutive
// positions, one per lane: a depth-wise tap needs no other channel, so the lane loads the nine taps of each of its 64
// channel values itself - 256-byte coalesced runs per (channel, row, column shift), the shifted copies are first-level
// cache hits - and reduces them in the depth-wise kernel's own order (bias, then the taps row by row).  The 32 products
// gelu(gate) * value go to a per-wave LDS tile [channel][position], from which the two 32-position MFMA tiles read their
// B operands in lfss_out's K order: fused and unfused results are BIT-IDENTICAL on fp32 planes.
// Groups in which no lane sits in the first or last image column take the unmasked path; zero padding above / below the
// image comes from the buffer range check.
// One plane (channel) of f as a raw buffer (a descriptor per channel: four SGPRs from scalar arithmetic): a 32-bit lane
// offset + immediate column shift per load, and the range check (offset >= plane bytes -> 0; a negative offset wraps to a
// huge one) supplies the zero padding ABOVE the first and BELOW the last image row for free.  (The scalar offset operand
// takes part in the range check on gfx950 - one descriptor over all planes + a per-channel soffset read zeros.)
template <typename TP> __device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, int voff);
template <> __device__ __forceinline__ float buf_ld<float>(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
template <> __device__ __forceinline__ float buf_ld<bf16_t>(__amdgpu_buffer_rsrc_t r, int voff) {
    return __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, voff, 0, 0) << 16);
}

// EDGE: some lane of the group sits in the first or last image column (its left / right taps are zero padding, not the
// neighbouring row's end).  Rows need no care (range check above).
template <bool EDGE, typename TP>
__device__ __forceinline__ void dwconv_gate_positions(const TP* __restrict__ fb /* + b D L */, const float* __restrict__ s_cw,
                                                      long long L, int W, long long p /* the lane's position, < L */,
                                                      float* __restrict__ sg /* the wave's [32 channels][64 positions] + lane */) {
    constexpr int E = (int)sizeof(TP);
    bool ml = true, mr = true;
    if constexpr (EDGE) {
        const int col = (int)((unsigned)p % (unsigned)W);            // p < L < 2^31
        ml = col > 0; mr = col < W - 1;
    }
    int off[3];
#pragma unroll
    for (int dr = 0; dr < 3; ++dr) off[dr] = (int)((p + (long long)(dr - 1) * W) * E);
    // A ROLLED loop over groups of four channel pairs (72 loads in flight): unrolled, the compiler hoists all 576 loads of
    // the lane above their first use and spills a thousand registers.
#pragma unroll 1
    for (int c0 = 0; c0 < 32; c0 += 4) {
        float t[4][2][9];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(               // wave-uniform
                    const_cast<TP*>(fb + (long long)(c0 + cc + 32 * v) * L), 0, (int)(L * E), 0x00020000);
#pragma unroll
                for (int dr = 0; dr < 3; ++dr) {
                    const float a = buf_ld<TP>(rs, off[dr] - E), b = buf_ld<TP>(rs, off[dr]), e = buf_ld<TP>(rs, off[dr] + E);
                    t[cc][v][3 * dr] = (!EDGE || ml) ? a : 0.0f; t[cc][v][3 * dr + 1] = b; t[cc][v][3 * dr + 2] = (!EDGE || mr) ? e : 0.0f;
                }
            }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            float fc[2];
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const float* wk = s_cw + (c0 + cc + 32 * v) * 12;                                    // [9 taps | bias | 0 0]
                const float4 w0 = *reinterpret_cast<const float4*>(wk), w1 = *reinterpret_cast<const float4*>(wk + 4),
                             w2 = *reinterpret_cast<const float4*>(wk + 8);
                float a = w2.y;                                      // the depth-wise kernel's order: bias, then taps row by row
                a = fmaf(w0.x, t[cc][v][0], a); a = fmaf(w0.y, t[cc][v][1], a); a = fmaf(w0.z, t[cc][v][2], a);
                a = fmaf(w0.w, t[cc][v][3], a); a = fmaf(w1.x, t[cc][v][4], a); a = fmaf(w1.y, t[cc][v][5], a);
                a = fmaf(w1.z, t[cc][v][6], a); a = fmaf(w1.w, t[cc][v][7], a); a = fmaf(w2.x, t[cc][v][8], a);
                fc[v] = a;
            }
            sg[(c0 + cc) * 64] = gelu_erf(fc[0]) * fc[1];
        }
    }
}

#ifndef WM_LFSS_OUT_TILE
#define WM_LFSS_OUT_TILE 8        // image rows per band of the column-major group order (0 rows = linear order: pass gpr = 0)
#endif
template <typename TP = float>
__global__ __launch_bounds__(256, 4) void lfss_out_conv_mfma_kernel(
    const TP* __restrict__ f, const float* __restrict__ cw /*(D, 3, 3)*/, const float* __restrict__ cbias /*(D) or null*/,
    const float* __restrict__ tok1, const float* __restrict__ W3 /*(C, C)*/, const float* __restrict__ b3,
    const float* __restrict__ skip2, float* __restrict__ out, int out_nchw, int B, int H, int W, int ngl, long long ngroups,
    int gpw, int gpr /* groups per image row when W % 64 == 0, else 0 */) {
    constexpr int C = 32, D = 64;
    __shared__ __attribute__((aligned(16))) float s_b3[C];
    __shared__ __attribute__((aligned(16))) float s_skip[C];
    __shared__ __attribute__((aligned(16))) float s_A3[(C / 2) * 64];            // 16 A operands x 64 lanes, fetch order
    __shared__ __attribute__((aligned(16))) float s_cw[D * 12];                  // [channel][9 taps | bias | 0 0]
    __shared__ __attribute__((aligned(16))) float s_g[4 * C * 64];               // per wave: gelu(gate) * value, [channel][position]
    const long long L = (long long)H * W;
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < C) { s_b3[threadIdx.x] = b3[threadIdx.x]; s_skip[threadIdx.x] = skip2[threadIdx.x]; }
    for (int e = threadIdx.x; e < (C / 2) * 64; e += 256) {
        const int j = e >> 6, l = e & 63;
        s_A3[aop_slot(j, l)] = W3[(l & 31) * C + acc_chan(j, l >> 5)];    // lfss_out's K order: bit-identical sums
    }
    for (int e = threadIdx.x; e < D * 12; e += 256) {
        const int c = e / 12, q = e - 12 * c;
        s_cw[e] = q < 9 ? cw[c * 9 + q] : (q == 9 && cbias ? cbias[c] : 0.0f);
    }
    __syncthreads();
    const long long g0 = (long long)blockIdx.x * 4 * gpw + wv;     // the block's waves walk adjacent groups together
    for (int gi = 0; gi < gpw; ++gi) {
        const long long g = g0 + 4 * gi;
        if (g >= ngroups) break;
        const long long b = g / ngl;
        long long gl = g - b * ngl;
        // Banded group order (round 4).  In linear order a workgroup's 4 x gpw groups are ~2 image rows, and an f row is fetched by
        // every workgroup whose output rows touch it (workgroup i runs on XCD i % 8: no shared L2).  Within bands of
        // WM_LFSS_OUT_TILE image rows the order is column-major instead: the four waves of a workgroup sit on four vertically
        // adjacent groups of one 64-column strip at a time (then the next four rows, then the next strip), so a band's interior rows
        // come from the L2 its own waves just filled.  Same arithmetic per position: bit-identical outputs.  Measured
        // (tools/bench_lfss_out_conv.py against the linear order of round 3): 0.107 -> 0.088 ms at UHD level 2, 0.454 -> 0.468 at
        // level 1 (there the kernel is bound by its 576 tap loads per lane and group, not by HBM), level 3 (W % 64 != 0) keeps the
        // linear order: 0.05 ms per UHD image.
        if (gpr > 0) {
            const long long per = (long long)WM_LFSS_OUT_TILE * gpr, band = gl / per;
            if ((band + 1) * per <= ngl) {
                const int rem = (int)(gl - band * per);
                gl = (band * WM_LFSS_OUT_TILE + rem % WM_LFSS_OUT_TILE) * gpr + rem / WM_LFSS_OUT_TILE;
            }
        }
        const long long p0 = gl * 64;
        const long long pc = min(p0 + lane, L - 1);
        const int col0 = (int)((unsigned)p0 % (unsigned)W);          // wave-uniform; p0 < L < 2^31
        const bool edge = col0 == 0 || col0 + 64 >= W;               // a lane in the first / last column (or two rows in the group)
        float* sg = s_g + wv * (C * 64);
        const TP* fb = f + b * D * L;
        __builtin_amdgcn_wave_barrier();                             // the previous group's operand reads are done (in-order LDS)
        if (edge) dwconv_gate_positions<true, TP>(fb, s_cw, L, W, pc, sg + lane);
        else dwconv_gate_positions<false, TP>(fb, s_cw, L, W, pc, sg + lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        lfss_v16f acc[2];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 bb = *reinterpret_cast<const float4*>(&s_b3[8 * gq + 4 * h]);
            acc[0][4 * gq] = bb.x; acc[0][4 * gq + 1] = bb.y; acc[0][4 * gq + 2] = bb.z; acc[0][4 * gq + 3] = bb.w;
        }
        acc[1] = acc[0];
#pragma unroll
        for (int j4 = 0; j4 < C / 8; ++j4) {
            const float4 a4 = *reinterpret_cast<const float4*>(&s_A3[(j4 * 64 + lane) * 4]);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {            // K-step j = channels acc_chan(j, h); tile 0 = positions n, tile 1 = 32 + n
                const int ch = acc_chan(4 * j4 + jj, h);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], sg[ch * 64 + n], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], sg[ch * 64 + 32 + n], acc[1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long long pos = p0 + 32 * t + n;
            float tk[16], o[16];
            load_tile32(tok1, false, b, min(pos, L - 1), L, h, tk);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 s4 = *reinterpret_cast<const float4*>(&s_skip[8 * gq + 4 * h]);
                o[4 * gq] = fmaf(tk[4 * gq], s4.x, acc[t][4 * gq]); o[4 * gq + 1] = fmaf(tk[4 * gq + 1], s4.y, acc[t][4 * gq + 1]);
                o[4 * gq + 2] = fmaf(tk[4 * gq + 2], s4.z, acc[t][4 * gq + 2]); o[4 * gq + 3] = fmaf(tk[4 * gq + 3], s4.w, acc[t][4 * gq + 3]);
            }
            if (pos < L) store_tile32(out, out_nchw != 0, b, pos, L, h, o);
        }
    }
}

// ---- lfss_out_conv, accumulating row-window form (round 6) -------------------------------------------------------------------
// PMC over the UHD step (profiles/r04/pmc_step_traffic_per_kernel.txt) showed the one-row form HBM-bound on 2.3 x its algorithmic
// bytes: every wave fetches the three tap rows of its output row itself.  Rounds 4-5 kept R = 2 output rows' product tiles
// [32 gated channels][64 positions] in LDS (R x 8 KB per wave: R = 3 already left too few waves) and fetched every tap row three times (left / centre / right: 9 (R + 2) / R load instructions per output value and
// channel).  Here the 32 x 32 closing product ACCUMULATES over groups of eight gated channels - the K order of lfss_out, so the sums
// stay bit-identical - in R x 2 register tiles, which leaves 2 R KB of products in LDS per wave, and a tap row is ONE coalesced
// load: the left / right neighbours come from the adjacent lanes (DPP wave_shr:1 / wave_shl:1), the two values a strip lacks (the
// column before its first and after its last) from one more load that only touches two cache lines.  2 (R + 2) / R load
// instructions per output value and channel, (R + 2) / R of f; a wave walks `bpw` consecutive bands of ITS strip, so the two rows
// two bands share come from the cache the wave has just filled.  W % 64 == 0.
template <int R, typename TP>
__global__ __launch_bounds__(256, 2) void lfss_out_conv_acc_kernel(
    const TP* __restrict__ f, const float* __restrict__ cw /*(D, 3, 3)*/, const float* __restrict__ cbias /*(D) or null*/,
    const float* __restrict__ tok1, const float* __restrict__ W3 /*(C, C)*/, const float* __restrict__ b3,
    const float* __restrict__ skip2, float* __restrict__ out, int out_nchw, int B, int H, int W, int nstrips, int nbands,
    int bpw /* bands per walk */, int nchunks /* walks per strip */, long long nwalks) {
    constexpr int C = 32, D = 64, E = (int)sizeof(TP);
    __shared__ __attribute__((aligned(16))) float s_b3[C];
    __shared__ __attribute__((aligned(16))) float s_skip[C];
    __shared__ __attribute__((aligned(16))) float s_A3[(C / 2) * 64];
    __shared__ __attribute__((aligned(16))) float s_cw[D * 12];
    __shared__ __attribute__((aligned(16))) float s_g[4 * 8 * R * 64];           // per wave: [gated channel of the group][row][position]
    const long long L = (long long)H * W;
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < C) { s_b3[threadIdx.x] = b3[threadIdx.x]; s_skip[threadIdx.x] = skip2[threadIdx.x]; }
    for (int e = threadIdx.x; e < (C / 2) * 64; e += 256) {
        const int j = e >> 6, l = e & 63;
        s_A3[aop_slot(j, l)] = W3[(l & 31) * C + acc_chan(j, l >> 5)];
    }
    for (int e = threadIdx.x; e < D * 12; e += 256) {
        const int c = e / 12, q = e - 12 * c;
        s_cw[e] = q < 9 ? cw[c * 9 + q] : (q == 9 && cbias ? cbias[c] : 0.0f);
    }
    __syncthreads();
    const long long wk = (long long)blockIdx.x * 4 + wv;         // walk = (batch, chunk of bands, strip), strips fastest
    if (wk >= nwalks) return;
    const int strip = (int)(wk % nstrips);
    const long long wb = wk / nstrips;
    const int chunk = (int)(wb % nchunks);
    const long long b = wb / nchunks;
    float* sg = s_g + wv * (8 * R * 64);
    const TP* fb = f + b * D * L;
    const bool hok = lane < 32 ? strip != 0 : strip != nstrips - 1;      // the lane's halo element lies inside the image
    const int hcol = strip * 64 + (lane < 32 ? -1 : 64);
    const int band_end = min(nbands, (chunk + 1) * bpw);
#pragma unroll 1
    for (int band = chunk * bpw; band < band_end; ++band) {
        const int r0 = band * R;
        int off[R + 2], hoff[R + 2];
#pragma unroll
        for (int dr = 0; dr < R + 2; ++dr) {                      // rows r0 - 1 .. r0 + R (outside the image: range check = zeros)
            const long long rowp = (long long)(r0 + dr - 1) * W;
            off[dr] = (int)((rowp + strip * 64 + lane) * E);
            hoff[dr] = (int)((rowp + hcol) * E);
        }
        lfss_v16f acc[R][2];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 bb = *reinterpret_cast<const float4*>(&s_b3[8 * gq + 4 * h]);
            acc[0][0][4 * gq] = bb.x; acc[0][0][4 * gq + 1] = bb.y; acc[0][0][4 * gq + 2] = bb.z; acc[0][0][4 * gq + 3] = bb.w;
        }
        acc[0][1] = acc[0][0];
#pragma unroll
        for (int rr = 1; rr < R; ++rr) { acc[rr][0] = acc[0][0]; acc[rr][1] = acc[0][0]; }
        // (a double-buffered form - the next channel's loads in flight under this one's products - measured the same at UHD level 1
        // and 7 % slower at level 2, with four spilled registers: not kept)
#pragma unroll 1
        for (int j4 = 0; j4 < 4; ++j4) {
            __builtin_amdgcn_wave_barrier();                     // the previous group's operand reads are done (in-order LDS)
#pragma unroll 1
            for (int cs = 0; cs < 4; ++cs) {                      // two gated channels = four planes per batch of loads
                float ctr[2][2][R + 2], hal[2][2][R + 2];
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(               // wave-uniform
                            const_cast<TP*>(fb + (long long)(8 * j4 + 2 * cs + cc + 32 * v) * L), 0, (int)(L * E), 0x00020000);
#pragma unroll
                        for (int dr = 0; dr < R + 2; ++dr) {
                            ctr[cc][v][dr] = buf_ld<TP>(rs, off[dr]);
                            hal[cc][v][dr] = buf_ld<TP>(rs, hoff[dr]);
                        }
                    }
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    float w[2][12];
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        const float* wk9 = s_cw + (8 * j4 + 2 * cs + cc + 32 * v) * 12;                     // [9 taps | bias | 0 0]
                        const float4 w0 = *reinterpret_cast<const float4*>(wk9), w1 = *reinterpret_cast<const float4*>(wk9 + 4),
                                     w2 = *reinterpret_cast<const float4*>(wk9 + 8);
                        w[v][0] = w0.x; w[v][1] = w0.y; w[v][2] = w0.z; w[v][3] = w0.w; w[v][4] = w1.x; w[v][5] = w1.y;
                        w[v][6] = w1.z; w[v][7] = w1.w; w[v][8] = w2.x; w[v][9] = w2.y;
                    }
                    float t[2][R + 2][3];
#pragma unroll
                    for (int v = 0; v < 2; ++v)
#pragma unroll
                        for (int dr = 0; dr < R + 2; ++dr) {
                            const float hm = hok ? hal[cc][v][dr] : 0.0f;
                            t[v][dr][0] = dpp_from_lower_lane(hm, ctr[cc][v][dr]);
                            t[v][dr][1] = ctr[cc][v][dr];
                            t[v][dr][2] = dpp_from_upper_lane(hm, ctr[cc][v][dr]);
                        }
#pragma unroll
                    for (int rr = 0; rr < R; ++rr) {
                        float fc[2];
#pragma unroll
                        for (int v = 0; v < 2; ++v) {
                            float a = w[v][9];                       // the depth-wise kernel's order: bias, then taps row by row
#pragma unroll
                            for (int k = 0; k < 9; ++k) a = fmaf(w[v][k], t[v][rr + k / 3][k % 3], a);
                            fc[v] = a;
                        }
                        sg[((2 * cs + cc) * R + rr) * 64 + lane] = gelu_erf(fc[0]) * fc[1];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const float4 a4 = *reinterpret_cast<const float4*>(&s_A3[(j4 * 64 + lane) * 4]);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {                      // K-step j = 4 j4 + jj: channels acc_chan(j, h) = 8 j4 + jj + 4 h
                const float* sr = sg + ((jj + 4 * h) * R) * 64;
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    acc[rr][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], sr[rr * 64 + n], acc[rr][0], 0, 0, 0);
                    acc[rr][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], sr[rr * 64 + 32 + n], acc[rr][1], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            if (r0 + rr < H) {                                    // (wave-uniform) the last band may be short
                const long long p0 = (long long)(r0 + rr) * W + (long long)strip * 64;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const long long pos = p0 + 32 * tt + n;
                    float tk[16], o[16];
                    load_tile32(tok1, false, b, pos, L, h, tk);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const float4 s4 = *reinterpret_cast<const float4*>(&s_skip[8 * gq + 4 * h]);
                        o[4 * gq] = fmaf(tk[4 * gq], s4.x, acc[rr][tt][4 * gq]); o[4 * gq + 1] = fmaf(tk[4 * gq + 1], s4.y, acc[rr][tt][4 * gq + 1]);
                        o[4 * gq + 2] = fmaf(tk[4 * gq + 2], s4.z, acc[rr][tt][4 * gq + 2]); o[4 * gq + 3] = fmaf(tk[4 * gq + 3], s4.w, acc[rr][tt][4 * gq + 3]);
                    }
                    store_tile32(out, out_nchw != 0, b, pos, L, h, o);
                }
            }
        }
    }
}

}  // namespace wm
