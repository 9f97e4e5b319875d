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
//   * A wave owns ONE 16-channel tile of the input, every output tile and every tap: OT x TAPS accumulator tiles in
//     registers (144 at 64 output channels x 9 taps); the operand split work is then unique per wave for x (the 3 x 3
//     shifted windows come from ONE 10-element load per row: columns -1 .. 8 of the lane's eight) and small for gy.  The
//     4 waves of a workgroup are the (up to 4) input tiles over the SAME run of units - they read the same gy - and a
//     wave walks its units down a 32-column segment, keeping two of the three input rows from the unit before.
//   * Zero padding: a tap row outside the image is skipped (uniform); the two columns outside it are the first element of
//     the window in the row's first segment / the last in its last segment - masked per lane.
//   * Every wave writes its accumulators as one partial; a second small kernel adds the partials of an input tile into
//     dW (no atomics anywhere: bit-reproducible).
//   * The BIAS gradient db[co] = sum over b, h, w of gy rides along (round 5): the wave of input tile 0 has every gy value of its
//     position sub-range in registers anyway - eight adds per output tile and unit, one partial row per sub-range, summed by the
//     same finish kernel.  It was a separate pass over gy before (wm_plane_sums: 87 launches + 87 memsets, 1.4 ms of a
//     BASELINE config-3 step).
// Needs W % 32 == 0 (every map of the network: 512 / 256 / 128 / 64 wide at the training size) - else WM_EUNSUPPORTED and
// the caller stays on ATen.
#pragma once
#include <hip/hip_runtime.h>
#include "ss2d_core.hip.h"      // core_bf2 / core_bf8 / core_split2

namespace wm {

struct ConvWgradArgs {
    const float* gy;           // (B, Cout, H, W)
    const float* x;            // (B, Cin, H, W)
    float* part;               // [ci tile][position sub-range][OT * TAPS][16 (co)][16 (ci)]
    float* dW;                 // (Cout, Cin, KS, KS)
    int B, Cin, Cout, H, W;
    int upw;                   // units (32-column row segments) per position sub-range
    long long nunits;          // B * H * (W / 32)
    int nparts;                // position sub-ranges = partials per input tile
    int co0, nco;              // output channels co0 .. co0 + nco - 1 in this launch (96 = 64 + 32: the accumulators of six
                               // tiles x nine taps do not fit the registers)
    float* bpart;              // bias gradient: [position sub-range][kCwBiasRow] partial plane sums of gy (db != nullptr)
    float* db;                 // (Cout) or nullptr
};
constexpr int kCwBiasRow = 96;                                   // the most output channels a call takes

typedef float cw_f4 __attribute__((ext_vector_type(4)));
constexpr int kCwWaves = 4;

// tiles of the input a workgroup's waves share (wave -> tile wv % tpw of tile group blockIdx.y, position sub-range wv / tpw)
__host__ __device__ inline int cw_tiles_per_wg(int nit) { return nit >= 4 ? 4 : (nit == 3 ? 3 : nit); }

#ifndef WM_CW_WAVES_PER_SIMD
#define WM_CW_WAVES_PER_SIMD 2
#endif
template <int KS, int OT>
__global__ __launch_bounds__(64 * kCwWaves, WM_CW_WAVES_PER_SIMD) void conv_wgrad_kernel(const ConvWgradArgs a) {
    constexpr int TAPS = KS * KS, PAD = KS / 2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int nit = (a.Cin + 15) >> 4;
    const int tpw = cw_tiles_per_wg(nit), gpw = kCwWaves / tpw;   // input tiles / position sub-ranges per workgroup
    const int it = (int)blockIdx.y * tpw + wv % tpw;              // input-channel tile of this wave
    const int g = wv / tpw;
    if (it >= nit || g >= gpw) return;                            // (3 tiles: the fourth wave idles; ragged last tile group)
    const int prange = (int)blockIdx.x * gpw + g;                 // position sub-range = the partial this wave writes
    if (prange >= a.nparts) return;                               // (ragged last workgroup)
    const int ci = 16 * it + i16;
    const bool ci_ok = ci < a.Cin;
    const int segs = a.W >> 5;
    const long long HW = (long long)a.H * a.W;
    cw_f4 acc[OT][TAPS];
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[o][t] = (cw_f4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = a.db != nullptr && it == 0;              // uniform: one wave per position sub-range
    float gs[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) gs[o] = 0.0f;

    // Units in COLUMN order: u -> (image b, 32-column segment seg, row h), h fastest.  The waves of a workgroup that share a
    // position sub-range read the same gy (one first-level cache serves them), and walking DOWN a segment a wave keeps two
    // of a unit's three input rows in registers from the unit before: first the kernel ran row-major units with one input
    // tile per WORKGROUP - every gy line crossed the L2 once per input tile, every x row three times - and sat at the L2's
    // bandwidth (0.27 ms for 8 x 64 -> 64 x 256 x 256 whatever was done about latency or MFMA order).
    const int u_begin = prange * a.upw;                           // (nunits < 2^31: host check)
    const int u_end = (int)min(a.nunits, (long long)u_begin + a.upw);
    struct Row { float4 q[2]; float e[2]; };                      // columns w .. w + 7, and w - 1 / w + 8
    struct Raw { float4 g[OT][2]; Row x[KS]; };
    auto load_row = [&](const float* xp, int r, int w, Row& row) {
        const float* rp = xp + (long long)min(max(r, 0), a.H - 1) * a.W;
        row.q[0] = *reinterpret_cast<const float4*>(rp + w);
        row.q[1] = *reinterpret_cast<const float4*>(rp + w + 4);
        if (KS == 3) { row.e[0] = rp[max(w - 1, 0)]; row.e[1] = rp[min(w + 8, a.W - 1)]; }
    };
    // operands of unit u; `below`: u is the unit right under the one `prev` holds -> two of its rows are prev's
    auto fetch = [&](int u, Raw& r, const Raw& prev, bool below) {
        const int h = u % a.H;
        const int bs = u / a.H;
        const int seg = bs % segs, b = bs / segs;
        const int w = 32 * seg + 8 * kq;                          // the lane's first column
        const float* gp = a.gy + ((long long)b * a.Cout) * HW + (long long)h * a.W + w;
#pragma unroll
        for (int o = 0; o < OT; ++o) {
            const int co = min(a.co0 + 16 * o + i16, a.Cout - 1);
            r.g[o][0] = *reinterpret_cast<const float4*>(gp + (long long)co * HW);
            r.g[o][1] = *reinterpret_cast<const float4*>(gp + (long long)co * HW + 4);
        }
        const float* xp = a.x + ((long long)b * a.Cin + min(ci, a.Cin - 1)) * HW;
        if (KS == 3 && below) {                                   // uniform
            r.x[0] = prev.x[1]; r.x[1] = prev.x[KS - 1];
            load_row(xp, h + 1, w, r.x[KS - 1]);
        } else {
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) load_row(xp, h + ky - PAD, w, r.x[ky]);
        }
    };
#ifndef WM_CW_PREFETCH
#define WM_CW_PREFETCH 0
#endif
    Raw cur, nxt;
    if (u_begin < u_end) fetch(u_begin, cur, cur, false);
    for (int u = u_begin; u < u_end; ++u) {
        if (WM_CW_PREFETCH && u + 1 < u_end) fetch(u + 1, nxt, cur, (u + 1) % a.H != 0);   // uniform
        const int h = u % a.H;
        const int w = 32 * ((u / a.H) % segs) + 8 * kq;
        // gy tiles -> bf16 hi / lo (rows beyond the launch's output channels: zero)
        core_bf8 gh[OT], gl[OT];
#pragma unroll
        for (int o = 0; o < OT; ++o) {
            const bool ok = 16 * o + i16 < a.nco;
            const float v[8] = {cur.g[o][0].x, cur.g[o][0].y, cur.g[o][0].z, cur.g[o][0].w, cur.g[o][1].x, cur.g[o][1].y, cur.g[o][1].z, cur.g[o][1].w};
            if (do_bias) gs[o] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
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
            const Row& row = cur.x[ky];
            float e[10];
            e[0] = (KS == 3 && w > 0) ? row.e[0] : 0.0f;
            e[1] = row.q[0].x; e[2] = row.q[0].y; e[3] = row.q[0].z; e[4] = row.q[0].w;
            e[5] = row.q[1].x; e[6] = row.q[1].y; e[7] = row.q[1].z; e[8] = row.q[1].w;
            e[9] = (KS == 3 && w + 8 < a.W) ? row.e[1] : 0.0f;
            __bf16 eh[10], el[10];
#pragma unroll
            for (int j = 0; j < 10; j += 2) {
                core_bf2 h2, l2;
                core_split2(ci_ok ? e[j] : 0.f, ci_ok ? e[j + 1] : 0.f, h2, l2);
                eh[j] = h2[0]; eh[j + 1] = h2[1]; el[j] = l2[0]; el[j + 1] = l2[1];
            }
            // the KS windows first, then the products TERM by term over every (tap, output tile): consecutive MFMAs land
            // on different accumulators
            core_bf8 xh[KS], xl[KS];
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int s0 = KS == 3 ? kx : 1;                  // window = columns w + kx - PAD .. + 7 = e[s0 .. s0 + 7]
#pragma unroll
                for (int j = 0; j < 8; ++j) { xh[kx][j] = eh[s0 + j]; xl[kx][j] = el[s0 + j]; }
            }
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                    for (int o = 0; o < OT; ++o)
                        acc[o][ky * KS + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            term == 0 ? gl[o] : gh[o], term == 1 ? xl[kx] : xh[kx], acc[o][ky * KS + kx], 0, 0, 0);
        }
        if (WM_CW_PREFETCH) cur = nxt;
        else if (u + 1 < u_end) { nxt = cur; fetch(u + 1, cur, nxt, (u + 1) % a.H != 0); }
    }
    // D layout: lane holds rows 4 kq .. 4 kq + 3 (co within the tile) of column i16 (ci within the tile).  Every wave
    // writes its own partial (no sums across waves: bit-reproducible).
    float* out = a.part + ((long long)it * a.nparts + prange) * (OT * TAPS * 256);
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(o * TAPS + t) * 256 + (4 * kq + r) * 16 + i16] = acc[o][t][r];
    if (do_bias) {
#pragma unroll
        for (int o = 0; o < OT; ++o) {                            // the four column groups of a channel: lanes i16 + 16 kq
            float sb = gs[o];
            sb += __shfl_xor(sb, 16);
            sb += __shfl_xor(sb, 32);
            if (kq == 0 && 16 * o + i16 < a.nco) a.bpart[(long long)prange * kCwBiasRow + a.co0 + 16 * o + i16] = sb;
        }
    }
}

// dW[co][ci][tap] = sum over the partials of an input tile.  A block = 16 consecutive elements of a partial (one co row
// of one tile: a 64-byte run) x 16 groups of partials: thread (e16, pg) adds every 16th partial, the 16 group sums meet in
// LDS in a fixed order (bit-reproducible).  One thread per element looping over ALL partials was a chain of ~100 dependent
// loads on a grid of 8-16 blocks: 29 us per 1x1 convolution, more than its main kernel.
// grid (OT * TAPS * 16, input tiles), block (256).
template <int KS, int OT>
__global__ __launch_bounds__(256) void conv_wgrad_finish_kernel(const ConvWgradArgs a) {
    constexpr int TAPS = KS * KS, PER = OT * TAPS * 256;
    __shared__ float s_sum[16][17];
    const int e16 = threadIdx.x & 15, pg = threadIdx.x >> 4;
    const int f = blockIdx.x * 16 + e16;                          // element of one partial
    const int it = blockIdx.y;
    const float* p = a.part + ((long long)it * a.nparts) * PER + f;
    float s0 = 0.f, s1 = 0.f;
    int i = pg;
    for (; i + 16 < a.nparts; i += 32) { s0 += p[(long long)i * PER]; s1 += p[(long long)(i + 16) * PER]; }
    if (i < a.nparts) s0 += p[(long long)i * PER];
    s_sum[pg][e16] = s0 + s1;
    __syncthreads();
    if (pg == 0) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += s_sum[q][e16];
        const int c16 = f & 15, r16 = (f >> 4) & 15, tile = f >> 8, tap = tile % TAPS, o = tile / TAPS;
        const int co = 16 * o + r16, ci = 16 * it + c16;          // co: within this launch
        if (co < a.nco && ci < a.Cin) a.dW[((long long)(a.co0 + co) * a.Cin + ci) * TAPS + tap] = t;
    }
    if (a.db != nullptr && blockIdx.y == 0 && (int)blockIdx.x < OT) {       // uniform: block o = the bias gradient of output tile o
        __syncthreads();
        const int co = 16 * (int)blockIdx.x + e16;
        float sb = 0.0f;
        if (co < a.nco)
            for (int q = pg; q < a.nparts; q += 16) sb += a.bpart[(long long)q * kCwBiasRow + a.co0 + co];
        s_sum[pg][e16] = sb;
        __syncthreads();
        if (pg == 0 && co < a.nco) {
            float t = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += s_sum[q][e16];
            a.db[a.co0 + co] = t;
        }
    }
}

}  // namespace wm
