// lfss_mfma.hip.h - the LFSSBlock glue of lfss.hip.h for C = 32 (D = 64) with the 1x1 projections on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32).
//
// The thread-per-position kernels of lfss.hip.h stream their weights through SGPRs: 4096 FMAs per position need
// ~480 s_load per wave, and PMC shows the waves waiting on them (VALU issue 28 % of a wave's life at two waves per
// SIMD, `SQ_WAIT_INST_ANY` 31 %).  Here a wave keeps both weight matrices in 64 VGPRs as MFMA A operands for all
// the position groups it owns, and a group of 64 positions is two 32-column MFMA tiles:
//   * thread-per-position phase (LayerNorm over channels in registers, gates), then one v_permlane32_swap per
//     channel pair turns [positions 0-63 of channel 2j], [.. of 2j+1] into the B operands of both tiles;
//   * a 32x32 accumulator tile holds, per lane, position (lane & 31) and rows 8g + 4(lane >> 5) + i: 16 of the 32
//     channels, the other 16 in lane ^ 32 - LayerNorm statistics are an in-lane sum and one cross-half exchange;
//   * the accumulator registers are used AS the next projection's B operands without any movement: register j
//     holds channel (j&3) + 8(j>>2) in lanes 0-31 and that + 4 in lanes 32-63, so the A operand of step j simply
//     carries those two weight columns (the K order of a GEMM is free).
// ln_2's affine is folded into conv1 (W1' = W1 diag(w), b1' = b1 + W1 b): exact algebra, fp32 rounding differs at
// 1e-7.  fp32 MFMA keeps fp32 products (no operand splitting); the kernels are HBM-bound.
// Reference: basicsr/archs/wavemamba_arch.py :491-494 (SS2D tail), :525-526 (LFSSBlock), :226-230 (ffn).
#pragma once
#include <hip/hip_runtime.h>

namespace wm {

typedef float lfss_v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float silu_fast(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}
__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32); }

constexpr int kLfssSlots = 2048;        // resident waves: 1024 SIMDs x 2

// waves each take `gpw` consecutive groups of 64 positions; chosen so that the waves fill whole rounds of the slots
inline int lfss_groups_per_wave(long long ngroups) {
    const long long rounds = (ngroups + 8LL * kLfssSlots - 1) / (8LL * kLfssSlots);
    long long gpw = (ngroups + rounds * kLfssSlots - 1) / (rounds * kLfssSlots);
    return (int)(gpw < 1 ? 1 : gpw);
}

// ---- lfss_mid: ysum, z, tok -> tok1 (B, L, C), f (B, D, L) -------------------------------------------
__global__ __launch_bounds__(256, 2) void lfss_mid_mfma_kernel(
    const float* __restrict__ ysum, const float* __restrict__ z, const float* __restrict__ tok, int tok_nchw,
    const float* __restrict__ on_w, const float* __restrict__ on_b, float on_eps,
    const float* __restrict__ W_out /*(C, D)*/, const float* __restrict__ skip1,
    const float* __restrict__ ln2_w, const float* __restrict__ ln2_b, float ln2_eps,
    const float* __restrict__ W1 /*(D, C)*/, const float* __restrict__ b1,
    float* __restrict__ tok1, float* __restrict__ f, int B, long long L, int ngl, long long ngroups, int gpw) {
    constexpr int C = 32, D = 64;
    __shared__ __attribute__((aligned(16))) float s_skip[C];
    __shared__ __attribute__((aligned(16))) float s_b1[D];
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < C) s_skip[threadIdx.x] = skip1[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + D) {
        const int m = threadIdx.x - 64;
        float acc = b1[m];
        for (int k = 0; k < C; ++k) acc = fmaf(W1[m * C + k], ln2_b[k], acc);
        s_b1[m] = acc;
    }
    // A operands: lane (m = n, k = h)
    float Aout[D / 2], A1[2][C / 2];
#pragma unroll
    for (int j = 0; j < D / 2; ++j) Aout[j] = W_out[n * D + 2 * j + h];
#pragma unroll
    for (int j = 0; j < C / 2; ++j) {
        const int k = (j & 3) + 8 * (j >> 2) + 4 * h;
        const float g = ln2_w[k];
        A1[0][j] = W1[n * C + k] * g;
        A1[1][j] = W1[(32 + n) * C + k] * g;
    }
    __syncthreads();

    const long long g0 = ((long long)blockIdx.x * 4 + wv) * gpw;
    for (int gi = 0; gi < gpw; ++gi) {
        const long long g = g0 + gi;
        if (g >= ngroups) break;
        const long long b = g / ngl;
        const long long p0 = (g - b * ngl) * 64;
        // ---- thread-per-position: out_norm, gate ----
        const long long pc = min(p0 + lane, L - 1);
        const float* yp = ysum + b * D * L + pc;
        const float* zp = z + b * D * L + pc;
        float y[D];
#pragma unroll
        for (int d = 0; d < D; ++d) y[d] = yp[(long long)d * L];
        float mean = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) mean += y[d];
        mean *= (1.0f / D);
        float var = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) { const float q = y[d] - mean; var = fmaf(q, q, var); }
        const float rstd = rsqrtf(var * (1.0f / D) + on_eps);
#pragma unroll
        for (int d = 0; d < D; ++d)
            y[d] = fmaf((y[d] - mean) * rstd, on_w[d], on_b[d]) * silu_fast(zp[(long long)d * L]);
        // ---- B operands of the two tiles ----
#pragma unroll
        for (int j = 0; j < D / 2; ++j) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(y[2 * j]), __float_as_uint(y[2 * j + 1]),
                                                            false, false);
            y[2 * j] = __uint_as_float(r[0]);           // tile 0: positions p0 + n,      channels 2j + h
            y[2 * j + 1] = __uint_as_float(r[1]);       // tile 1: positions p0 + 32 + n, channels 2j + h
        }
        lfss_v16f acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
#pragma unroll
        for (int j = 0; j < D / 2; ++j) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Aout[j], y[2 * j], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Aout[j], y[2 * j + 1], acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long long pos = p0 + 32 * t + n;
            const bool ok = pos < L;
            const long long pq = min(pos, L - 1);
            float tt[16];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int row0 = 8 * gq + 4 * h;
                float4 tk;
                if (tok_nchw) {
                    const float* tp = tok + (b * C + row0) * L + pq;
                    tk = make_float4(tp[0], tp[L], tp[2 * L], tp[3 * L]);
                } else {
                    tk = *reinterpret_cast<const float4*>(tok + (b * L + pq) * C + row0);
                }
                const float4 sk = *reinterpret_cast<const float4*>(&s_skip[row0]);
                tt[4 * gq] = fmaf(tk.x, sk.x, acc[t][4 * gq]);
                tt[4 * gq + 1] = fmaf(tk.y, sk.y, acc[t][4 * gq + 1]);
                tt[4 * gq + 2] = fmaf(tk.z, sk.z, acc[t][4 * gq + 2]);
                tt[4 * gq + 3] = fmaf(tk.w, sk.w, acc[t][4 * gq + 3]);
            }
            if (ok) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<float4*>(tok1 + (b * L + pos) * C + 8 * gq + 4 * h) =
                        make_float4(tt[4 * gq], tt[4 * gq + 1], tt[4 * gq + 2], tt[4 * gq + 3]);
            }
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s += tt[i];
            const float m2 = xhalf_sum(s) * (1.0f / C);
            float v2 = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { tt[i] -= m2; v2 = fmaf(tt[i], tt[i], v2); }
            const float r2 = rsqrtf(xhalf_sum(v2) * (1.0f / C) + ln2_eps);
#pragma unroll
            for (int i = 0; i < 16; ++i) tt[i] *= r2;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                lfss_v16f a;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 bb = *reinterpret_cast<const float4*>(&s_b1[32 * mt + 8 * gq + 4 * h]);
                    a[4 * gq] = bb.x; a[4 * gq + 1] = bb.y; a[4 * gq + 2] = bb.z; a[4 * gq + 3] = bb.w;
                }
#pragma unroll
                for (int j = 0; j < C / 2; ++j) a = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[mt][j], tt[j], a, 0, 0, 0);
                if (ok) {
                    float* fp = f + (b * D + 32 * mt + 4 * h) * L + pos;
#pragma unroll
                    for (int i = 0; i < 16; ++i) fp[(long long)(8 * (i >> 2) + (i & 3)) * L] = a[i];
                }
            }
        }
    }
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

// ---- lfss_in: tok -> x (B, D, L), z (B, D, L) ------------------------------------------------------
// ln_1's affine is folded into in_proj (W' = W diag(w), bias' = W b).
__global__ __launch_bounds__(256, 2) void lfss_in_mfma_kernel(const float* __restrict__ tok, int tok_nchw,
                                                             const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                             float eps, const float* __restrict__ W_in /*(2D, C)*/,
                                                             float* __restrict__ x, float* __restrict__ z, int B, long long L,
                                                             int ngl, long long ngroups, int gpw) {
    constexpr int C = 32, D = 64;
    __shared__ __attribute__((aligned(16))) float s_bias[2 * D];
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 2 * D) {
        float acc = 0.0f;
        for (int k = 0; k < C; ++k) acc = fmaf(W_in[threadIdx.x * C + k], ln_b[k], acc);
        s_bias[threadIdx.x] = acc;
    }
    float A[4][C / 2];
#pragma unroll
    for (int j = 0; j < C / 2; ++j) {
        const int k = (j & 3) + 8 * (j >> 2) + 4 * h;
        const float g = ln_w[k];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) A[mt][j] = W_in[(32 * mt + n) * C + k] * g;
    }
    __syncthreads();
    const long long g0 = ((long long)blockIdx.x * 4 + wv) * gpw;
    for (int gi = 0; gi < gpw; ++gi) {
        const long long g = g0 + gi;
        if (g >= ngroups) break;
        const long long b = g / ngl;
        const long long p0 = (g - b * ngl) * 64;
        float a[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t) load_tile32(tok, tok_nchw != 0, b, min(p0 + 32 * t + n, L - 1), L, h, a[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long long pos = p0 + 32 * t + n;
            const bool ok = pos < L;
            tile_normalise(a[t], eps);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                lfss_v16f acc;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 bb = *reinterpret_cast<const float4*>(&s_bias[32 * mt + 8 * gq + 4 * h]);
                    acc[4 * gq] = bb.x; acc[4 * gq + 1] = bb.y; acc[4 * gq + 2] = bb.z; acc[4 * gq + 3] = bb.w;
                }
#pragma unroll
                for (int j = 0; j < C / 2; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[mt][j], a[t][j], acc, 0, 0, 0);
                if (ok) {
                    float* dp = (mt < 2 ? x + (b * D + 32 * mt + 4 * h) * L : z + (b * D + 32 * (mt - 2) + 4 * h) * L) + pos;
#pragma unroll
                    for (int i = 0; i < 16; ++i) dp[(long long)(8 * (i >> 2) + (i & 3)) * L] = acc[i];
                }
            }
        }
    }
}

// ---- lfss_out: fc (B, D, L), tok1 -> tok2 --------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void lfss_out_mfma_kernel(const float* __restrict__ fc, const float* __restrict__ tok1,
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
    const long long g0 = ((long long)blockIdx.x * 4 + wv) * gpw;
    for (int gi = 0; gi < gpw; ++gi) {
        const long long g = g0 + gi;
        if (g >= ngroups) break;
        const long long b = g / ngl;
        const long long p0 = (g - b * ngl) * 64;
        float gate[2][16], val[2][16], tk[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long long pq = min(p0 + 32 * t + n, L - 1);
            const float* fp = fc + (b * D + 4 * h) * L + pq;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long long ro = (long long)(8 * (i >> 2) + (i & 3)) * L;
                gate[t][i] = fp[ro];
                val[t][i] = fp[ro + 32 * L];
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

}  // namespace wm
