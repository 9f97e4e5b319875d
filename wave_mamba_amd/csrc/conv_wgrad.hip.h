// conv_wgrad.hip.h - weight gradient of the dense 3x3 / 1x1 convolutions (stride 1, 'same' zero padding) for gfx950.
//
//   dW[co][ci][ky][kx] = sum over b, h, w of  gy[b][co][h][w] * x[b][ci][h + ky - PAD][w + kx - PAD]
//
// Autograd reaches it through every nn.Conv2d of the network in training (reference: femasr_model.py:181 ->
// wavemamba_arch.py: the plumbing / HFE convolutions, and in_proj / out_proj / the gated ffn's 1x1 of LFSSBlock on the NCHW
// training path).  On ROCm these went to MIOpen's implicit-GEMM weight-gradient kernels in NHWC - with three layout
// transposes around each - and, for 1x1, to hipBLASLt: ~19 ms of an 87-ms BASELINE config-3 training step on one MI355X
// (tools/train_breakdown.py).
//
// It is a GEMM whose K dimension is the positions: (Cout x K) . (K x Cin) per tap.  Same form as the x_proj weight gradient
// (ss2d_bwd.hip.h: projgrad_kernel): bf16 matrix cores with both operands split into two bf16 terms (hi + lo), the product
// accumulated in fp32 as hi.hi + hi.lo + lo.hi (~4e-6 relative per product; a leaf gradient: nothing propagates the error);
// K = 32 consecutive positions of one image row per v_mfma_f32_16x16x32_bf16: lane (row i16, kq) feeds positions
// 8 kq .. 8 kq + 7 of its channel plane - two 16-byte loads per operand tile, straight from NCHW.
//   * A workgroup = 4 waves on ONE 16-channel tile of the input (blockIdx.y), every output tile and every tap: a wave
//     walks its own run of (image, row, 32-column segment) units with OT x TAPS accumulator tiles in registers
//     (144 registers at 64 output channels x 9 taps); the operand split work is then unique per wave for x (the 3 x 3
//     shifted windows come from ONE 10-element load per row: columns -1 .. 8 of the lane's eight) and small for gy.
//   * Zero padding: a tap row outside the image is skipped (uniform); the two columns outside it are the first element of
//     the window in the row's first segment / the last in its last segment - masked per lane.
//   * The waves of a block add their tiles in LDS, the block writes ONE partial, a second small kernel adds the blocks'
//     partials into dW (no global atomics).
// Needs W % 32 == 0 (every map of the network: 512 / 256 / 128 / 64 wide at the training size) - else WM_EUNSUPPORTED and
// the caller stays on ATen.
#pragma once
#include <hip/hip_runtime.h>
#include "ss2d_core.hip.h"      // core_bf2 / core_bf8 / core_split2

namespace wm {

struct ConvWgradArgs {
    const float* gy;           // (B, Cout, H, W)
    const float* x;            // (B, Cin, H, W)
    float* part;               // [ci tile][block][OT * TAPS][16 (co)][16 (ci)]
    float* dW;                 // (Cout, Cin, KS, KS)
    int B, Cin, Cout, H, W;
    int upw;                   // units (32-column row segments) per wave
    long long nunits;          // B * H * (W / 32)
    int nblocks;               // position blocks (gridDim.x)
    int co0, nco;              // output channels co0 .. co0 + nco - 1 in this launch (96 = 64 + 32: the accumulators of six
                               // tiles x nine taps do not fit the registers)
};

typedef float cw_f4 __attribute__((ext_vector_type(4)));
constexpr int kCwWaves = 4;

template <int KS, int OT>
__global__ __launch_bounds__(64 * kCwWaves, 2) void conv_wgrad_kernel(const ConvWgradArgs a) {
    constexpr int TAPS = KS * KS, PAD = KS / 2;
    extern __shared__ __attribute__((aligned(16))) float cw_smem[];          // [OT * TAPS][256]
    for (int e = threadIdx.x; e < OT * TAPS * 256; e += 64 * kCwWaves) cw_smem[e] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
#ifndef WM_CW_ITFAST
#define WM_CW_ITFAST 1
#endif
    const int nit = (a.Cin + 15) >> 4;
    const int it = WM_CW_ITFAST ? (int)(blockIdx.x % nit) : (int)blockIdx.y;        // input-channel tile
    const int pblk = WM_CW_ITFAST ? (int)(blockIdx.x / nit) : (int)blockIdx.x;      // position block
    const int ci = 16 * it + i16;
    const bool ci_ok = ci < a.Cin;
    const int segs = a.W >> 5;
    const long long HW = (long long)a.H * a.W;
    cw_f4 acc[OT][TAPS];
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[o][t] = (cw_f4){0.f, 0.f, 0.f, 0.f};

    const int u_begin = (pblk * kCwWaves + wv) * a.upw;         // (nunits < 2^31: host check)
    const int u_end = (int)min(a.nunits, (long long)u_begin + a.upw);
    for (int u = u_begin; u < u_end; ++u) {
        const int seg = u % segs;
        const int bh = u / segs;
        const int h = bh % a.H, b = bh / a.H;
        const int w = 32 * seg + 8 * kq;                          // the lane's first column
        // ---- every load of the unit first (unconditional, clamped), then the splits and the products
        const float* gp = a.gy + ((long long)b * a.Cout) * HW + (long long)h * a.W + w;
        float4 gr[OT][2];
#pragma unroll
        for (int o = 0; o < OT; ++o) {
            const int co = min(a.co0 + 16 * o + i16, a.Cout - 1);
            gr[o][0] = *reinterpret_cast<const float4*>(gp + (long long)co * HW);
            gr[o][1] = *reinterpret_cast<const float4*>(gp + (long long)co * HW + 4);
        }
        const float* xp = a.x + ((long long)b * a.Cin + min(ci, a.Cin - 1)) * HW;
        float4 xr[KS][2];
        float xe[KS][2];                                          // columns w - 1 and w + 8 (3x3 only)
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int r = min(max(h + ky - PAD, 0), a.H - 1);
            const float* rp = xp + (long long)r * a.W;
            xr[ky][0] = *reinterpret_cast<const float4*>(rp + w);
            xr[ky][1] = *reinterpret_cast<const float4*>(rp + w + 4);
            if (KS == 3) { xe[ky][0] = rp[max(w - 1, 0)]; xe[ky][1] = rp[min(w + 8, a.W - 1)]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        // gy tiles -> bf16 hi / lo (rows beyond Cout: zero)
        core_bf8 gh[OT], gl[OT];
#pragma unroll
        for (int o = 0; o < OT; ++o) {
            const bool ok = 16 * o + i16 < a.nco;
            const float v[8] = {gr[o][0].x, gr[o][0].y, gr[o][0].z, gr[o][0].w, gr[o][1].x, gr[o][1].y, gr[o][1].z, gr[o][1].w};
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                core_bf2 h2, l2;
                core_split2(ok ? v[j] : 0.f, ok ? v[j + 1] : 0.f, h2, l2);
                gh[o][j] = h2[0]; gh[o][j + 1] = h2[1]; gl[o][j] = l2[0]; gl[o][j + 1] = l2[1];
            }
        }
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int r = h + ky - PAD;
            if (r < 0 || r >= a.H) continue;                      // uniform: a tap row outside the image contributes nothing
            // the ten columns w - 1 .. w + 8 of this row as bf16 hi / lo; e[j] = column w - 1 + j
            float e[10];
            e[0] = (KS == 3 && w > 0) ? xe[ky][0] : 0.0f;
            e[1] = xr[ky][0].x; e[2] = xr[ky][0].y; e[3] = xr[ky][0].z; e[4] = xr[ky][0].w;
            e[5] = xr[ky][1].x; e[6] = xr[ky][1].y; e[7] = xr[ky][1].z; e[8] = xr[ky][1].w;
            e[9] = (KS == 3 && w + 8 < a.W) ? xe[ky][1] : 0.0f;
            __bf16 eh[10], el[10];
#pragma unroll
            for (int j = 0; j < 10; j += 2) {
                core_bf2 h2, l2;
                core_split2(ci_ok ? e[j] : 0.f, ci_ok ? e[j + 1] : 0.f, h2, l2);
                eh[j] = h2[0]; eh[j + 1] = h2[1]; el[j] = l2[0]; el[j + 1] = l2[1];
            }
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int s0 = KS == 3 ? kx : 1;                  // window = columns w + kx - PAD .. + 7 = e[s0 .. s0 + 7]
                core_bf8 xh, xl;
#pragma unroll
                for (int j = 0; j < 8; ++j) { xh[j] = eh[s0 + j]; xl[j] = el[s0 + j]; }
#pragma unroll
                for (int o = 0; o < OT; ++o) {
                    cw_f4 c = acc[o][ky * KS + kx];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl[o], xh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[o], xl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[o], xh, c, 0, 0, 0);
                    acc[o][ky * KS + kx] = c;
                }
            }
        }
    }
    // D layout: lane holds rows 4 kq .. 4 kq + 3 (co within the tile) of column i16 (ci within the tile)
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(&cw_smem[(o * TAPS + t) * 256 + (4 * kq + r) * 16 + i16], acc[o][t][r]);
    __syncthreads();
    float* out = a.part + ((long long)it * a.nblocks + pblk) * (OT * TAPS * 256);
    for (int e = threadIdx.x; e < OT * TAPS * 256; e += 64 * kCwWaves) out[e] = cw_smem[e];
}

// dW[co][ci][tap] = sum over the position blocks' partials.  One thread per element of a partial, in the PARTIAL's order
// (input tile, output tile x tap, co row, ci column: consecutive threads read consecutive words of every block's partial;
// the scattered side is the 4-byte store into dW).  grid (ceil(OT * TAPS * 256 / 256), input tiles), block (256).
template <int KS, int OT>
__global__ __launch_bounds__(256) void conv_wgrad_finish_kernel(const ConvWgradArgs a) {
    constexpr int TAPS = KS * KS, PER = OT * TAPS * 256;
    const int f = blockIdx.x * 256 + threadIdx.x;                 // element of one partial
    if (f >= PER) return;
    const int it = blockIdx.y;
    const int c16 = f & 15, r16 = (f >> 4) & 15, tile = f >> 8, tap = tile % TAPS, o = tile / TAPS;
    const int co = 16 * o + r16, ci = 16 * it + c16;              // co: within this launch
    if (co >= a.nco || ci >= a.Cin) return;
    const float* p = a.part + ((long long)it * a.nblocks) * PER + f;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = 0;
    for (; i + 3 < a.nblocks; i += 4) {
        s0 += p[(long long)i * PER]; s1 += p[(long long)(i + 1) * PER];
        s2 += p[(long long)(i + 2) * PER]; s3 += p[(long long)(i + 3) * PER];
    }
    for (; i < a.nblocks; ++i) s0 += p[(long long)i * PER];
    a.dW[((long long)(a.co0 + co) * a.Cin + ci) * TAPS + tap] = (s0 + s1) + (s2 + s3);
}

}  // namespace wm
