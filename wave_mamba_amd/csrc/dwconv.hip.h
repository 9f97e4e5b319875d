// dwconv.hip.h - depth-wise 3x3 convolution (stride 1, zero padding 1) over NCHW fp32, with bias and
// an optional fused SiLU, for gfx950.
//
// Reference call sites: SS2D.conv2d + SiLU (/root/reference/basicsr/archs/wavemamba_arch.py:346-355,
// :487) and the gated ffn's conv2 (:220, :226) inside LFSSBlock - both `groups == channels`.  On
// ROCm these land on MIOpen's dense Winograd kernel (~30x the streaming time at UHD, see
// profiles/r01/bench_kernel_stats_step_tail.txt); the op is pure bandwidth: 9 FMA per 8 bytes moved.
//
// One thread walks a 4-column strip down RH output rows with a 3-row register window, so every
// input row is fetched once per strip (16-byte loads); the two halo columns come from the
// neighbouring lanes by a DPP wave shift (the strip edges of a wave read them from memory).
#pragma once
#include <hip/hip_runtime.h>
#include "haar.hip.h"          // bf16_t, load4 / store4 / ld1 (fp32 and bf16 overloads)

namespace wm {

// A value from the neighbouring lane of the 64-lane wave: one DPP move (v_mov_b32 wave_shr:1 / wave_shl:1) instead of a
// ds_bpermute through the LDS crossbar (what __shfl_up / __shfl_down compile to).  The lane without a neighbour keeps `own_if_...`.
__device__ __forceinline__ float dpp_from_lower_lane(float own_if_first, float v) {       // lane i <- lane i - 1; lane 0 keeps `own_if_first`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, own_if_first), __builtin_bit_cast(int, v),
                                                                 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_upper_lane(float own_if_last, float v) {         // lane i <- lane i + 1; lane 63 keeps `own_if_last`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, own_if_last), __builtin_bit_cast(int, v),
                                                                 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

constexpr int kDwRows = 16;     // output rows per thread strip (the kernels' `rows` argument: 16, or 8 / 4 where 16 leave the chip short of waves)

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

// exact (erf) GELU, nn.GELU() default: FeedForward.project_out[1] after its depth-wise conv (reference :739-741)
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// TP: storage type of the x / y planes (float, or bf16_t in the bf16-storage mode: fp32 arithmetic either way)
// LPR: lanes per strip row.  64: a wave is one strip of 256 columns.  32 / 16 (maps of <= 128 / <= 64 columns - the 128 x 128 and
// 64 x 64 maps of a BASELINE config-3 training step): a wave is 2 / 4 PLANES side by side, 32 / 16 lanes each - at 64 lanes per plane
// three quarters of a wave were idle on a 64-column map (24 us per launch for 16 MB).  The halo shuffles cross the planes' borders
// harmlessly: a plane's first / last lane takes its halo from memory, as the wave's first / last lane always did.
template <int ACT /*0 none, 1 silu, 2 gelu*/, bool VEC, typename TP = float, int LPR = 64>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const TP* __restrict__ x,
                                                        const float* __restrict__ wgt,
                                                        const float* __restrict__ bias,
                                                        TP* __restrict__ y, int C, int H, int W,
                                                        long long planes, int rows, int flip) {
    constexpr int PPW = 64 / LPR;                        // planes per wave
    const int lane = threadIdx.x;                        // LPR column groups = one strip row of one plane
    const int cl = lane & (LPR - 1);
    const int cg = blockIdx.x * LPR + cl;                // column group (4 columns)
    const int h0 = (blockIdx.y * 4 + threadIdx.y) * rows;
    for (long long pg = blockIdx.z; pg * PPW < planes; pg += gridDim.z) {
        const long long plane_raw = pg * PPW + (LPR == 64 ? 0 : lane / LPR);   // LPR = 64: wave-uniform (weights, bias and the plane's
                                                                                 // base address stay in scalar registers)
        const bool pok = plane_raw < planes;             // (a last, partly filled group of planes: its spare lanes are masked)
        const long long plane = pok ? plane_raw : planes - 1;
        const int c = (int)(plane % C);
        float k[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) k[i] = wgt[c * 9 + (flip ? 8 - i : i)];   // flip: the taps rotated by 180 degrees (input gradient)
        const float bv = bias ? bias[c] : 0.0f;
        const TP* xp = x + plane * (long long)H * W;
        TP* yp = y + plane * (long long)H * W;
        const int w0 = cg * 4;
        const bool colok = pok && w0 < W;                // whole quad in range when VEC (W % 4 == 0)

        // row(r): 6 values x[r][w0-1 .. w0+4], zero outside the image
        auto load_row = [&](int r, float (&v)[6]) {
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool rowok = r >= 0 && r < H;
            if (rowok && colok) {
                if constexpr (VEC) {
                    float t4[4];
                    load4(xp + (long long)r * W + w0, t4);
                    q = make_float4(t4[0], t4[1], t4[2], t4[3]);
                } else {
                    const TP* p = xp + (long long)r * W + w0;
                    q.x = ld1(p);
                    if (w0 + 1 < W) q.y = ld1(p + 1);
                    if (w0 + 2 < W) q.z = ld1(p + 2);
                    if (w0 + 3 < W) q.w = ld1(p + 3);
                }
            }
            float left = dpp_from_lower_lane(q.w, q.w), right = dpp_from_upper_lane(q.x, q.x);
            if (cl == 0) left = (rowok && w0 - 1 >= 0 && w0 - 1 < W) ? ld1(xp + (long long)r * W + w0 - 1) : 0.0f;
            if (cl == LPR - 1) right = (rowok && w0 + 4 < W) ? ld1(xp + (long long)r * W + w0 + 4) : 0.0f;
            v[0] = left; v[1] = q.x; v[2] = q.y; v[3] = q.z; v[4] = q.w; v[5] = right;
        };

        // 16-byte form: a row's loads are issued UNCONDITIONALLY from clamped addresses (the quad, and the one halo element
        // the wave's first / last lane needs from memory), two rows ahead of their use, and masked when consumed.  Written
        // as `ok ? load : 0` the loads sat behind branches with a full wait each, and the halo load was issued only after
        // the quad had arrived and been shuffled: two exposed memory latencies per output row (tools/isa_load_waits.py).
        auto fetch = [&](int r, float (&q)[4], float& e) {
            const int rc = min(max(r, 0), H - 1);
            const TP* rowp = xp + (long long)rc * W;
            load4(rowp + (colok ? w0 : 0), q);
            int we = cl == 0 ? w0 - 1 : w0 + 4;
            we = min(max(we, 0), W - 1);
            e = ld1(rowp + we);
        };
        auto finish = [&](int r, const float (&q)[4], float e, float (&v)[6]) {
            const bool rowok = r >= 0 && r < H, ok = rowok && colok;
            const float q0 = ok ? q[0] : 0.f, q1 = ok ? q[1] : 0.f, q2 = ok ? q[2] : 0.f, q3 = ok ? q[3] : 0.f;
            float left = dpp_from_lower_lane(q3, q3), right = dpp_from_upper_lane(q0, q0);
            if (cl == 0) left = (rowok && w0 - 1 >= 0 && w0 - 1 < W) ? e : 0.0f;
            if (cl == LPR - 1) right = (rowok && w0 + 4 < W) ? e : 0.0f;
            v[0] = left; v[1] = q0; v[2] = q1; v[3] = q2; v[4] = q3; v[5] = right;
        };

        if (h0 < H) {                                     // uniform per wave (threadIdx.y, blockIdx.y)
            float r0[6], r1[6], r2[6];
            const int hend = min(H, h0 + rows);
            auto body = [&](int h) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float acc = bv;
                    acc = fmaf(k[0], r0[j], acc); acc = fmaf(k[1], r0[j + 1], acc); acc = fmaf(k[2], r0[j + 2], acc);
                    acc = fmaf(k[3], r1[j], acc); acc = fmaf(k[4], r1[j + 1], acc); acc = fmaf(k[5], r1[j + 2], acc);
                    acc = fmaf(k[6], r2[j], acc); acc = fmaf(k[7], r2[j + 1], acc); acc = fmaf(k[8], r2[j + 2], acc);
                    o[j] = ACT == 1 ? silu_f(acc) : ACT == 2 ? gelu_f(acc) : acc;
                }
                if (colok) {
                    if constexpr (VEC) {
                        store4(yp + (long long)h * W + w0, o);
                    } else {
                        TP* p = yp + (long long)h * W + w0;
                        st1(p, o[0]);
                        if (w0 + 1 < W) st1(p + 1, o[1]);
                        if (w0 + 2 < W) st1(p + 2, o[2]);
                        if (w0 + 3 < W) st1(p + 3, o[3]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) { r0[j] = r1[j]; r1[j] = r2[j]; }
            };
            if constexpr (VEC) {
                float qa[4], qb[4], qc[4], qn[4], ea, eb, ec, en;
                fetch(h0 - 1, qa, ea); fetch(h0, qb, eb); fetch(h0 + 1, qc, ec);
                finish(h0 - 1, qa, ea, r0);
                finish(h0, qb, eb, r1);
                for (int h = h0; h < hend; ++h) {
                    fetch(h + 2, qn, en);                  // (past the strip's / the image's end: a clamped row, unused or masked)
                    finish(h + 1, qc, ec, r2);
                    body(h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) qc[j] = qn[j];
                    ec = en;
                }
            } else {
                load_row(h0 - 1, r0);
                load_row(h0, r1);
                for (int h = h0; h < hend; ++h) {
                    load_row(h + 1, r2);
                    body(h);
                }
            }
        }
    }
}

}  // namespace wm

namespace wm {

// ---- weight / bias gradient of the depth-wise 3x3 convolution ------------------------------------------
// dW[c][i][j] = sum_{b,h,w} gy[b,c,h,w] * x[b,c,h+i-1,w+j-1]   (zero padding),   db[c] = sum gy[b,c,h,w]
// Same strip walk as the forward (3-row register window of x, 4 columns per thread); 10 per-thread partial
// sums are reduced over the wave by shuffles and leave through one atomicAdd per wave and value.
template <bool VEC, int LPR = 64>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                              float* __restrict__ dW, float* __restrict__ db, int C,
                                                              int H, int W, long long planes, int rows) {
    constexpr int PPW = 64 / LPR;                        // planes per wave (dwconv3x3_kernel: narrow maps)
    const int lane = threadIdx.x;
    const int cl = lane & (LPR - 1);
    const int cg = blockIdx.x * LPR + cl;
    const int h0 = (blockIdx.y * 4 + threadIdx.y) * rows;
    for (long long pg = blockIdx.z; pg * PPW < planes; pg += gridDim.z) {
        const long long plane_raw = pg * PPW + (LPR == 64 ? 0 : lane / LPR);   // LPR = 64: wave-uniform (weights, bias and the plane's
                                                                                 // base address stay in scalar registers)
        const bool pok = plane_raw < planes;
        const long long plane = pok ? plane_raw : planes - 1;
        const int c = (int)(plane % C);
        const float* xp = x + plane * (long long)H * W;
        const float* gp = gy + plane * (long long)H * W;
        const int w0 = cg * 4;
        const bool colok = pok && w0 < W;
        auto load_row = [&](int r, float (&v)[6]) {
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool rowok = r >= 0 && r < H;
            if (rowok && colok) {
                if constexpr (VEC) q = *reinterpret_cast<const float4*>(xp + (long long)r * W + w0);
                else {
                    const float* p = xp + (long long)r * W + w0;
                    q.x = p[0];
                    if (w0 + 1 < W) q.y = p[1];
                    if (w0 + 2 < W) q.z = p[2];
                    if (w0 + 3 < W) q.w = p[3];
                }
            }
            float left = dpp_from_lower_lane(q.w, q.w), right = dpp_from_upper_lane(q.x, q.x);
            if (cl == 0) left = (rowok && w0 - 1 >= 0 && w0 - 1 < W) ? xp[(long long)r * W + w0 - 1] : 0.0f;
            if (cl == LPR - 1) right = (rowok && w0 + 4 < W) ? xp[(long long)r * W + w0 + 4] : 0.0f;
            v[0] = left; v[1] = q.x; v[2] = q.y; v[3] = q.z; v[4] = q.w; v[5] = right;
        };
        // 16-byte form: as in the forward kernel - unconditional loads from clamped addresses (x quad, its halo element,
        // the gy quad), one row ahead of their use, masked when consumed
        auto fetch = [&](int r, float4& q, float& e, float4& gq) {     // x row r (+ halo), gy row r - 1
            const int rc = min(max(r, 0), H - 1), gc = min(max(r - 1, 0), H - 1);
            const float* rowp = xp + (long long)rc * W;
            q = *reinterpret_cast<const float4*>(rowp + (colok ? w0 : 0));
            int we = cl == 0 ? w0 - 1 : w0 + 4;
            we = min(max(we, 0), W - 1);
            e = rowp[we];
            gq = *reinterpret_cast<const float4*>(gp + (long long)gc * W + (colok ? w0 : 0));
        };
        auto finish = [&](int r, const float4& q, float e, float (&v)[6]) {
            const bool rowok = r >= 0 && r < H, ok = rowok && colok;
            const float q0 = ok ? q.x : 0.f, q1 = ok ? q.y : 0.f, q2 = ok ? q.z : 0.f, q3 = ok ? q.w : 0.f;
            float left = dpp_from_lower_lane(q3, q3), right = dpp_from_upper_lane(q0, q0);
            if (cl == 0) left = (rowok && w0 - 1 >= 0 && w0 - 1 < W) ? e : 0.0f;
            if (cl == LPR - 1) right = (rowok && w0 + 4 < W) ? e : 0.0f;
            v[0] = left; v[1] = q0; v[2] = q1; v[3] = q2; v[4] = q3; v[5] = right;
        };
        float acc[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) acc[i] = 0.0f;
        if (h0 < H) {
            float r0[6], r1[6], r2[6];
            const int hend = min(H, h0 + rows);
            auto body = [&](const float (&g)[4]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        acc[j] = fmaf(g[q], r0[q + j], acc[j]);
                        acc[3 + j] = fmaf(g[q], r1[q + j], acc[3 + j]);
                        acc[6 + j] = fmaf(g[q], r2[q + j], acc[6 + j]);
                    }
                    acc[9] += g[q];
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) { r0[j] = r1[j]; r1[j] = r2[j]; }
            };
            if constexpr (VEC) {
                float4 qa, qb, qc, qn, ga, gb, gc4, gn;
                float ea, eb, ec, en;
                fetch(h0 - 1, qa, ea, ga); fetch(h0, qb, eb, gb); fetch(h0 + 1, qc, ec, gc4);   // gc4 = gy row h0
                finish(h0 - 1, qa, ea, r0);
                finish(h0, qb, eb, r1);
                for (int h = h0; h < hend; ++h) {
                    fetch(h + 2, qn, en, gn);              // x row h + 2, gy row h + 1
                    finish(h + 1, qc, ec, r2);
                    const float g[4] = {colok ? gc4.x : 0.f, colok ? gc4.y : 0.f, colok ? gc4.z : 0.f, colok ? gc4.w : 0.f};
                    body(g);
                    qc = qn; ec = en; gc4 = gn;
                }
            } else {
            load_row(h0 - 1, r0);
            load_row(h0, r1);
            for (int h = h0; h < hend; ++h) {
                load_row(h + 1, r2);
                float g[4] = {0.f, 0.f, 0.f, 0.f};
                if (colok) {
                    const float* p = gp + (long long)h * W + w0;
                    g[0] = p[0]; if (w0 + 1 < W) g[1] = p[1]; if (w0 + 2 < W) g[2] = p[2]; if (w0 + 3 < W) g[3] = p[3];
                }
                body(g);
            }
            }
        }
        // over the plane's lanes (shuffles), then over the workgroup's four strips (LDS, fixed order), then ONE atomic per value and
        // plane: the atomics are what a launch on a small map costs (round 5: four times the strips = twice the time)
        __shared__ float s_part[4][PPW][10];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            float v = acc[i];
#pragma unroll
            for (int off = LPR / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            if (cl == 0) s_part[threadIdx.y][LPR == 64 ? 0 : lane / LPR][i] = v;
        }
        __syncthreads();
        if (threadIdx.y == 0 && cl == 0) {
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int sp = LPR == 64 ? 0 : lane / LPR;
                const float v = (s_part[0][sp][i] + s_part[1][sp][i]) + (s_part[2][sp][i] + s_part[3][sp][i]);
                if (v != 0.0f) {
                    if (i < 9) atomicAdd(dW + c * 9 + i, v);
                    else if (db) atomicAdd(db + c, v);
                }
            }
        }
        __syncthreads();                                 // s_part is rewritten by the next group of planes
    }
}

}  // namespace wm
