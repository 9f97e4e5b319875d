// ss2d.hip.h - fused SS2D four-direction scan core for gfx950 (MI355X).
//
// Replaces SS2D.forward_core (/root/reference/basicsr/archs/wavemamba_arch.py:446-478):
//   xs = [x row-major | x column-major | their flips]                         (:451-452)
//   x_dbl[k] = x_proj_weight[k] . xs[k]  -> split (dt_r | B | C)              (:453-454)
//   dts[k]   = dt_projs_weight[k] . dt_r                                      (:455)
//   y[k]     = selective_scan(xs[k], dts[k], -exp(A_logs), B, C, Ds, bias, softplus)   (:465-471)
//   flips / transposes back to row-major                                      (:474-478)
// without materialising xs / x_dbl / dts / flips / transposes (9.5 GB of intermediates per
// LFSSBlock at UHD level 1 in the reference).  Three kernel families:
//
//  1. ss2d_proj_kernel - the only GEMM-shaped piece: per position a (4 x (R+2N)) x D_in mat-vec,
//     i.e. an (136 x 64) x (64 x L) GEMM at the shipped config.  fp32-input MFMA
//     (v_mfma_f32_16x16x4_f32, exact fp32 = an fmaf chain): A = the stacked weights, kept in VGPRs
//     for the whole kernel (9 row-tiles x 16 K-steps), B = x read straight from NCHW in fragment
//     layout (16 lanes x 8 B = one 128-B line per channel row), D = per-position records
//         rec[b][k][p] = [dt_r(4) | B(16) | C(16)]   (p = row-major position, 144 B, 16-B aligned)
//     written as 16-byte pieces.  Every direction's record lives at the ROW-MAJOR position, so no
//     direction ever needs a transposed copy.
//  2. ss2d_row_kernel (k = 0, 2) - lane = channel, time = row-major l (reversed for k = 2): the
//     chunked scan of selscan.hip.h with u read from x, dt rebuilt in-register from dt_r
//     (R FMAs + softplus per step) and B/C taken from the record tile (plain linear LDS copy).
//  3. ss2d_col_kernel (k = 1, 3) - lanes = 64 adjacent COLUMNS, time = row h (reversed for k = 3):
//     u loads and y stores are coalesced straight in NCHW, every lane scans its own column segment;
//     a workgroup (4 waves x 2 channels) shares the record rows through LDS; A / dt weights of a
//     wave's channels are wave-uniform (SGPR operands of the packed ops).
//  All three write y in row-major (B, D, L), optionally accumulating (y1+y2+y3+y4 of :490).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "selscan.hip.h"

namespace wm {

constexpr int kRecPad = 4;                 // dt_r slots at the head of a record (R <= 4)
constexpr int kRS = kRecPad + 32;          // record stride in floats for N <= 16: 36 floats = 144 B

struct Ss2dArgs {
    const float* x;          // (B, D, H, W)
    float* rec;              // (B, 4, L, kRS)
    const float* Wx;         // (4, R + 2N, D)    x_proj_weight
    const float* Wdt;        // (4, D, R)         dt_projs_weight
    const float* dtb;        // (4, D)            dt_projs_bias
    const float* A_logs;     // (4 D, N)
    const float* Ds;         // (4 D)
    float* y;                // (B, D, L) output of direction k (row-major positions)
    float* wsP; float* wsH;  // chunk summaries [chunk][b*D + d][16]
    int B, D, H, W, L, N, R, k;
    int chunk_len, nchunks;  // row kernel: steps per chunk along l.  col kernel: rows per segment, W * nseg
    int nseg;                // col kernel: segments per column
    int accumulate;          // y += instead of y =
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// 1. projection: records for all four directions (MFMA fp32 16x16x4)
//    row tiles: tile 0 = dt_r rows (direction k in row group k: rows 4k .. 4k+3, r < R used),
//               tile 1 + 2k = B rows of direction k, tile 2 + 2k = C rows of direction k.
// ------------------------------------------------------------------------------------------------
constexpr int kProjTiles = 9;
constexpr int kProjKS = 16;                // K-steps of 4 -> D_in <= 64

// The 144 A fragments (9 tiles x 16 K-steps) live in LDS, padded to 12 per K-step so that a K-step is three
// conflict-free ds_read_b128 (operand 12 s + t of lane l at ((3 s + (t >> 2)) * 64 + l) * 4 + (t & 3)).  Held in
// registers they cost 144 VGPRs: one wave per SIMD, nothing to overlap its store phase with (MFMA busy 53 %).  The
// records of one direction at a time are assembled in a 4.6-KB slab per wave: 4 waves x 4.6 KB + 49 KB of fragments
// = two workgroups per compute unit.
constexpr int kProjWaves = 4;
constexpr int kProjWfrag = kProjKS * 12 * 64;           // floats
__global__ __launch_bounds__(64 * kProjWaves, 2) void ss2d_proj_kernel(Ss2dArgs p, int groups_per_batch) {
    __shared__ __attribute__((aligned(16))) float s_w[kProjWfrag];                      // 49,152 B
    __shared__ __attribute__((aligned(16))) float s_out[kProjWaves * 32 * kRS];         // 4 waves x 4,608 B
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int wave = blockIdx.x * kProjWaves + wv;
    const int nwaves = gridDim.x * kProjWaves;
    const int g4 = lane >> 4, j16 = lane & 15;
    const int C = p.R + 2 * p.N;

    // A fragments: lane l holds A[row = l & 15][kk = l >> 4] of every (tile, K-step): W[c(row)][d = 4 s + kk]
    for (int e = threadIdx.x; e < kProjKS * 12 * 64; e += 64 * kProjWaves) {
        const int l = e & 63, o = e >> 6, sidx = o / 12, t = o - 12 * sidx;
        const int r16 = l & 15, kk = l >> 4;
        float v = 0.0f;
        if (t < kProjTiles) {
            int kdir, c;
            bool ok;
            if (t == 0) { kdir = r16 >> 2; c = r16 & 3; ok = c < p.R; }
            else { kdir = (t - 1) >> 1; ok = r16 < p.N; c = p.R + ((t - 1) & 1) * p.N + r16; }
            const int d = 4 * sidx + kk;
            if (ok && d < p.D) v = p.Wx[((long long)kdir * C + c) * p.D + d];
        }
        s_w[((3 * sidx + (t >> 2)) * 64 + l) * 4 + (t & 3)] = v;
    }
    __syncthreads();

    const long long L = p.L;
    const long long total = (long long)p.B * groups_per_batch;          // groups of 32 positions
    // common case (D_in == 64, even L): unconditional 8-byte loads with clamped addresses - per-load
    // predication costs more issue slots than the MFMAs it feeds
    const bool fast = (p.D == 4 * kProjKS) && ((L & 1) == 0) && L >= 2;
    auto load_x = [&](long long grp, float2 (&v)[kProjKS]) {
        if (fast) {
            const long long gc = grp < total ? grp : total - 1;
            const int bb = (int)(gc / groups_per_batch);
            long long pc = (gc - (long long)bb * groups_per_batch) * 32 + 2 * j16;
            pc = pc < L - 2 ? pc : L - 2;
            const float* q = p.x + ((long long)bb * p.D + g4) * L + pc;
#pragma unroll
            for (int s = 0; s < kProjKS; ++s) v[s] = *reinterpret_cast<const float2*>(q + (long long)(4 * s) * L);
            return;
        }
        const int b = (int)(grp / groups_per_batch);
        const long long pj = (grp - (long long)b * groups_per_batch) * 32 + 2 * j16;
        const float* xb = p.x + (long long)b * p.D * L;
#pragma unroll
        for (int s = 0; s < kProjKS; ++s) {
            const int d = 4 * s + g4;
            v[s] = make_float2(0.f, 0.f);
            if (grp < total && d < p.D) {
                const float* q = xb + (long long)d * L + pj;
                if ((L & 1) == 0) { if (pj < L) v[s] = *reinterpret_cast<const float2*>(q); }
                else { if (pj < L) v[s].x = q[0]; if (pj + 1 < L) v[s].y = q[1]; }
            }
        }
    };
    float2 xnext[kProjKS];
    load_x(wave, xnext);
    float* slab = s_out + wv * (32 * kRS);
    for (long long grp = wave; grp < total; grp += nwaves) {
        const int b = (int)(grp / groups_per_batch);
        const long long p0 = (grp - (long long)b * groups_per_batch) * 32;
        f32x4 acc[kProjTiles][2];
#pragma unroll
        for (int t = 0; t < kProjTiles; ++t) { acc[t][0] = (f32x4){0, 0, 0, 0}; acc[t][1] = (f32x4){0, 0, 0, 0}; }
        float2 xv[kProjKS];
#pragma unroll
        for (int s = 0; s < kProjKS; ++s) xv[s] = xnext[s];
        load_x(grp + nwaves, xnext);                                    // in flight under the MFMAs below
#pragma unroll
        for (int s = 0; s < kProjKS; ++s) {
            float wf[12];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(&s_w[((3 * s + q) * 64 + lane) * 4]);
                wf[4 * q] = w4[0]; wf[4 * q + 1] = w4[1]; wf[4 * q + 2] = w4[2]; wf[4 * q + 3] = w4[3];
            }
#pragma unroll
            for (int t = 0; t < kProjTiles; ++t) {
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t], xv[s].x, acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t], xv[s].y, acc[t][1], 0, 0, 0);
            }
        }
        // D layout: lane holds rows 4*g4 .. 4*g4+3 of column j16.  Assemble the 32 records of one direction in the
        // wave's private LDS slab, then write them as ONE contiguous 4608-B run of 16-byte stores (direct stores
        // would be 16-B pieces at a 144-B stride).
        const int npos = (int)min((long long)32, L - p0);
#pragma unroll
        for (int kd = 0; kd < 4; ++kd) {
            __builtin_amdgcn_wave_barrier();              // the previous direction's reads are done (in-order LDS)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pl = 2 * j16 + i;
                if (g4 == kd) *reinterpret_cast<f32x4*>(&slab[pl * kRS]) = acc[0][i];        // dt_r of direction g4
                float* rk = &slab[pl * kRS + kRecPad + 4 * g4];
                *reinterpret_cast<f32x4*>(rk) = acc[1 + 2 * kd][i];
                *reinterpret_cast<f32x4*>(rk + 16) = acc[2 + 2 * kd][i];
            }
            __builtin_amdgcn_wave_barrier();
            float* dst = p.rec + (((long long)b * 4 + kd) * L + p0) * kRS;
#pragma unroll
            for (int it = 0; it < (32 * kRS / 4 + 63) / 64; ++it) {
                const int f = lane + 64 * it;
                if (4 * f < npos * kRS)
                    *reinterpret_cast<f32x4*>(dst + 4 * f) = *reinterpret_cast<const f32x4*>(&slab[4 * f]);
            }
        }
    }
}

// The same projection for d_state in (16, 32] (BASELINE config 5's block; training only - the inference core projects inside
// its scan kernels): records [dt_r (4) | B (32) | C (32)] = 68 floats of the TWO directions (kA, kB) one layout of the
// backward uses (row layout: 0 and 2, transposed layout: 1 and 3), rec (B, 2, L, 68).  Row tiles: 0 = dt_r rows of
// all four directions (as above), 1 + 4 i + {0, 1, 2, 3} = B[0:16], B[16:32], C[0:16], C[16:32] of direction i of the pair:
// nine tiles again, the same K loop.
constexpr int kRS32 = kRecPad + 64;
__global__ __launch_bounds__(64 * kProjWaves, 2) void ss2d_proj32_kernel(Ss2dArgs p, int groups_per_batch, int kA, int kB) {
    __shared__ __attribute__((aligned(16))) float s_w[kProjWfrag];                      // 49,152 B
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int wave = blockIdx.x * kProjWaves + wv;
    const int nwaves = gridDim.x * kProjWaves;
    const int g4 = lane >> 4, j16 = lane & 15;
    const int C = p.R + 2 * p.N;
    for (int e = threadIdx.x; e < kProjKS * 12 * 64; e += 64 * kProjWaves) {
        const int l = e & 63, o = e >> 6, sidx = o / 12, t = o - 12 * sidx;
        const int r16 = l & 15, kk = l >> 4;
        float v = 0.0f;
        if (t < kProjTiles) {
            int kdir, c;
            bool ok;
            if (t == 0) { kdir = r16 >> 2; c = r16 & 3; ok = c < p.R; }
            else {
                const int q = (t - 1) & 3, n = 16 * (q & 1) + r16;        // q: B lo, B hi, C lo, C hi
                kdir = (t - 1) >> 2 ? kB : kA; ok = n < p.N; c = p.R + (q >> 1) * p.N + n;
            }
            const int d = 4 * sidx + kk;
            if (ok && d < p.D) v = p.Wx[((long long)kdir * C + c) * p.D + d];
        }
        s_w[((3 * sidx + (t >> 2)) * 64 + l) * 4 + (t & 3)] = v;
    }
    __syncthreads();
    const long long L = p.L;
    const long long total = (long long)p.B * groups_per_batch;          // groups of 32 positions
    for (long long grp = wave; grp < total; grp += nwaves) {
        const int b = (int)(grp / groups_per_batch);
        const long long p0 = (grp - (long long)b * groups_per_batch) * 32;
        const long long pj = p0 + 2 * j16;
        const float* xb = p.x + (long long)b * p.D * L;
        f32x4 acc[kProjTiles][2];
#pragma unroll
        for (int t = 0; t < kProjTiles; ++t) { acc[t][0] = (f32x4){0, 0, 0, 0}; acc[t][1] = (f32x4){0, 0, 0, 0}; }
#pragma unroll
        for (int s = 0; s < kProjKS; ++s) {
            const int d = 4 * s + g4;
            float x0 = 0.0f, x1 = 0.0f;
            if (d < p.D) {
                const float* q = xb + (long long)d * L + pj;
                if (pj < L) x0 = q[0];
                if (pj + 1 < L) x1 = q[1];
            }
            float wf[12];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(&s_w[((3 * s + q) * 64 + lane) * 4]);
                wf[4 * q] = w4[0]; wf[4 * q + 1] = w4[1]; wf[4 * q + 2] = w4[2]; wf[4 * q + 3] = w4[3];
            }
#pragma unroll
            for (int t = 0; t < kProjTiles; ++t) {
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t], x0, acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t], x1, acc[t][1], 0, 0, 0);
            }
        }
        // D layout: lane holds rows 4 g4 .. 4 g4 + 3 of column j16: 16-byte pieces straight into the records (this path
        // serves training at d_state 32 only; the N <= 16 kernel assembles whole records in LDS first)
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const int kd = pi ? kB : kA;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const long long pos = pj + i;
                if (pos < L) {
                    float* rk = p.rec + (((long long)b * 2 + pi) * L + pos) * kRS32;
                    if (g4 == kd) *reinterpret_cast<f32x4*>(rk) = acc[0][i];                  // dt_r of direction g4
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(rk + kRecPad + 4 * g4 + 16 * q) = acc[1 + 4 * pi + q][i];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 2. row-major directions (k = 0 forward, k = 2 reversed): lane = channel
// ------------------------------------------------------------------------------------------------
template <int PHASE, bool REV, bool VEC>
__global__ __launch_bounds__(64) void ss2d_row_kernel(Ss2dArgs p) {
    constexpr int NP = 16;
    constexpr int T = 16;                            // steps per tile
    constexpr int ROW = 20;                          // padded LDS row of the u tile
    constexpr int NREC4 = (T * kRS / 4 + 63) / 64;   // float4 per lane of a record tile (144 -> 3)
    __shared__ __attribute__((aligned(16))) float s_u[64 * ROW];
    __shared__ __attribute__((aligned(16))) float s_rec[T * kRS];

    const int lane = threadIdx.x;
    const int chunk = blockIdx.x, b = blockIdx.y, k = p.k;
    const bool live = lane < p.D;
    const int d = live ? lane : 0;
    const int kd = k * p.D + d;
    const long long L = p.L;
    const int t_begin = chunk * p.chunk_len;
    const int t_end = min(p.L, t_begin + p.chunk_len);

    // parameter loads first, unconditional with clamped indices (see the column kernel)
    v2f A2[NP / 2];
    float araw[NP], wdt[kRecPad];
#pragma unroll
    for (int n = 0; n < NP; ++n) araw[n] = p.A_logs[(long long)kd * p.N + min(n, p.N - 1)];
#pragma unroll
    for (int r = 0; r < kRecPad; ++r) wdt[r] = p.Wdt[(long long)kd * p.R + min(r, p.R - 1)];
    const float bias = p.dtb[kd];
    const float Dd = p.Ds[kd];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        const float a = (n < p.N) ? -expf(araw[n]) * 1.4426950408889634f : 0.0f;
        if (n & 1) A2[n / 2].y = a; else A2[n / 2].x = a;
    }
#pragma unroll
    for (int r = 0; r < kRecPad; ++r) wdt[r] = (r < p.R) ? wdt[r] : 0.0f;

    v2f h[NP / 2];
    const long long wsrow = ((long long)chunk * p.B * p.D + (long long)b * p.D + d) * NP;
    if (PHASE == 3 && chunk > 0) {
#pragma unroll
        for (int q = 0; q < NP / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(p.wsH + wsrow + 4 * q);
            h[2 * q] = (v2f){v.x, v.y}; h[2 * q + 1] = (v2f){v.z, v.w};
        }
    } else {
#pragma unroll
        for (int n = 0; n < NP / 2; ++n) h[n] = splat(0.0f);
    }
    float sum_dt = 0.0f;

    const float* xb = p.x + (long long)b * p.D * L;
    const float* recb = p.rec + ((long long)b * 4 + k) * L * kRS;
    float* yb = (PHASE == 3) ? p.y + (long long)b * p.D * L : nullptr;

    float4 ru[4], rr[NREC4], ry[4];                 // ry: previous directions' y (accumulate mode)
    const int trow = lane >> 2, tq = lane & 3;

    // tile at step t0 covers positions plo .. plo+15 (column c <-> position plo + c);
    // step tt uses column tt (forward) or 15 - tt (reversed)
    auto tile_lo = [&](int t0) -> long long { return REV ? (L - 16 - t0) : (long long)t0; };

    auto fetch = [&](int t0) {
        const long long plo = tile_lo(t0);
        const int tl = min(T, t_end - t0);
        const int c_lo = REV ? T - tl : 0, c_hi = REV ? T : tl;          // valid columns [c_lo, c_hi)
        const int c = 4 * tq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 16 * i + trow;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < p.D) {
                const float* q = xb + (long long)r * L + plo + c;
                if constexpr (VEC) {
                    if (c >= c_lo && c < c_hi) v = *reinterpret_cast<const float4*>(q);
                } else {
                    if (c + 0 >= c_lo && c + 0 < c_hi) v.x = q[0];
                    if (c + 1 >= c_lo && c + 1 < c_hi) v.y = q[1];
                    if (c + 2 >= c_lo && c + 2 < c_hi) v.z = q[2];
                    if (c + 3 >= c_lo && c + 3 < c_hi) v.w = q[3];
                }
            }
            ru[i] = v;
            if (PHASE == 3) {
                float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.accumulate && r < p.D) {
                    const float* q = yb + (long long)r * L + plo + c;
                    if constexpr (VEC) {
                        if (c >= c_lo && c < c_hi) e = *reinterpret_cast<const float4*>(q);
                    } else {
                        if (c + 0 >= c_lo && c + 0 < c_hi) e.x = q[0];
                        if (c + 1 >= c_lo && c + 1 < c_hi) e.y = q[1];
                        if (c + 2 >= c_lo && c + 2 < c_hi) e.z = q[2];
                        if (c + 3 >= c_lo && c + 3 < c_hi) e.w = q[3];
                    }
                }
                ry[i] = e;
            }
        }
#pragma unroll
        for (int j = 0; j < NREC4; ++j) {
            const int f = lane + 64 * j;                       // float4 index inside the record tile
            const int col = (4 * f) / kRS;
            rr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool used = PHASE == 3 || (4 * f - col * kRS) < kRecPad + NP;    // reduce phase: no C
            if (f < T * kRS / 4 && col >= c_lo && col < c_hi && used)
                rr[j] = *reinterpret_cast<const float4*>(recb + plo * kRS + 4 * f);
        }
    };
    float4 yold[4];
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&s_u[(16 * i + trow) * ROW + 4 * tq]) = ru[i];
            if (PHASE == 3) yold[i] = ry[i];
        }
#pragma unroll
        for (int j = 0; j < NREC4; ++j) {
            const int f = lane + 64 * j;
            if (f < T * kRS / 4) *reinterpret_cast<float4*>(&s_rec[4 * f]) = rr[j];
        }
    };

    fetch(t_begin);
    for (int t0 = t_begin; t0 < t_end; t0 += T) {
        stage();
        __syncthreads();
        if (t0 + T < t_end) fetch(t0 + T);
        const int tl = min(T, t_end - t0);

#pragma unroll
        for (int q = 0; q < T / 4; ++q) {
            if (4 * q < tl) {
                const int cq = REV ? 3 - q : q;                          // column quad of this step quad
                const float4 u4 = *reinterpret_cast<const float4*>(&s_u[lane * ROW + 4 * cq]);
                const float uu[4] = {REV ? u4.w : u4.x, REV ? u4.z : u4.y, REV ? u4.y : u4.z, REV ? u4.x : u4.w};
                float dts[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = REV ? 15 - (4 * q + j) : 4 * q + j;
                    const float4 dr = *reinterpret_cast<const float4*>(&s_rec[col * kRS]);
                    dts[j] = fmaf(wdt[1], dr.y, fmaf(wdt[0], dr.x, bias));
                    if (p.R > 2) dts[j] = fmaf(wdt[3], dr.w, fmaf(wdt[2], dr.z, dts[j]));               // uniform
                }
                const v2f sa = softplus2((v2f){dts[0], dts[1]}), sb = softplus2((v2f){dts[2], dts[3]});
                dts[0] = sa.x; dts[1] = sa.y; dts[2] = sb.x; dts[3] = sb.y;
                float yy[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int tt = 4 * q + j;
                    if (tt < tl) {
                        const int col = REV ? 15 - tt : tt;
                        const float dt = dts[j], ut = uu[j];
                        const v2f dt2 = splat(dt), du2 = splat(dt * ut);
                        if (PHASE == 1) sum_dt += dt;
                        v2f y2 = splat(0.0f);
                        const float* rc = &s_rec[col * kRS + kRecPad];
#pragma unroll
                        for (int r = 0; r < NP / 4; ++r) {
                            const float4 bv = *reinterpret_cast<const float4*>(rc + 4 * r);
                            const v2f a0 = exp2_2(dt2 * A2[2 * r]);
                            const v2f a1 = exp2_2(dt2 * A2[2 * r + 1]);
                            h[2 * r] = a0 * h[2 * r] + du2 * (v2f){bv.x, bv.y};
                            h[2 * r + 1] = a1 * h[2 * r + 1] + du2 * (v2f){bv.z, bv.w};
                            if (PHASE == 3) {
                                const float4 cv = *reinterpret_cast<const float4*>(rc + NP + 4 * r);
                                y2 = (v2f){cv.x, cv.y} * h[2 * r] + y2;
                                y2 = (v2f){cv.z, cv.w} * h[2 * r + 1] + y2;
                            }
                        }
                        if (PHASE == 3) yy[j] = fmaf(Dd, ut, y2.x + y2.y);
                    }
                }
                if (PHASE == 3)
                    *reinterpret_cast<float4*>(&s_u[lane * ROW + 4 * cq]) =
                        REV ? make_float4(yy[3], yy[2], yy[1], yy[0]) : make_float4(yy[0], yy[1], yy[2], yy[3]);
            }
        }
        __syncthreads();
        if (PHASE == 3) {
            const long long plo = tile_lo(t0);
            const int c_lo = REV ? T - tl : 0, c_hi = REV ? T : tl;
            const int c = 4 * tq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * i + trow;
                if (r < p.D) {
                    const float4 v = *reinterpret_cast<const float4*>(&s_u[r * ROW + c]);
                    float* o = yb + (long long)r * L + plo + c;
                    const float4 e = yold[i];
                    if constexpr (VEC) {
                        if (c >= c_lo && c < c_hi)
                            *reinterpret_cast<float4*>(o) = make_float4(v.x + e.x, v.y + e.y, v.z + e.z, v.w + e.w);
                    } else {
                        const float vv[4] = {v.x + e.x, v.y + e.y, v.z + e.z, v.w + e.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (c + j >= c_lo && c + j < c_hi) o[j] = vv[j];
                    }
                }
            }
            __syncthreads();
        }
    }

    if (PHASE == 1 && live) {
#pragma unroll
        for (int q = 0; q < NP / 4; ++q) {
            *reinterpret_cast<float4*>(p.wsH + wsrow + 4 * q) =
                make_float4(h[2 * q].x, h[2 * q].y, h[2 * q + 1].x, h[2 * q + 1].y);
            const v2f p0 = exp2_2(splat(sum_dt) * A2[2 * q]), p1 = exp2_2(splat(sum_dt) * A2[2 * q + 1]);
            *reinterpret_cast<float4*>(p.wsP + wsrow + 4 * q) = make_float4(p0.x, p0.y, p1.x, p1.y);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 3. column-major directions (k = 1 forward, k = 3 reversed): lanes = 64 adjacent columns
//    scan order l = w*H + h: time tau = h (k=1) or H-1-h (k=3); column order omega = w or W-1-w;
//    chunk = (omega, segment of tau), chunk index = omega * nseg + seg.
// ------------------------------------------------------------------------------------------------
#ifndef WM_COLT
#define WM_COLT 4
#endif
#ifndef WM_COLWAVES
#define WM_COLWAVES 4
#endif
#ifndef WM_COL_LB
#define WM_COL_LB 1
#endif
#ifndef WM_COL_SLOTS
#define WM_COL_SLOTS 512     // resident column-scan workgroups: 256 compute units x 2
#endif
constexpr int kColT = WM_COLT;  // record rows per LDS batch
constexpr int kColCH = 2;       // channels per wave
constexpr int kColWaves = WM_COLWAVES;   // waves per workgroup -> 2 * kColWaves channels per workgroup

// LDS image of a record row batch: kColT rows x 64 columns x RSL floats.  The chunk-scan phase keeps whole records
// (36 floats); the chunk-reduce phase never reads C and keeps only [dt_r | B] (20 floats).  Both strides are
// conflict-free for the 16-byte B/C reads (every 16-lane group of ds_read_b128 lands on 64 distinct banks).
template <int PHASE> struct ColRec { static constexpr int RSL = PHASE == 3 ? kRS : kRecPad + 16; };
template <int PHASE> constexpr int col_lds_bytes() { return 2 * kColT * 64 * ColRec<PHASE>::RSL * 4; }

template <int PHASE, bool REV>
__global__ __launch_bounds__(64 * kColWaves, WM_COL_LB) void ss2d_col_kernel(Ss2dArgs p) {
    constexpr int NP = 16;
    constexpr int RSL = ColRec<PHASE>::RSL;          // floats per record in LDS
    constexpr int CH4 = RSL / 4;                     // 16-byte pieces per record (9 or 5)
    constexpr int BUF = kColT * 64 * RSL;            // floats per buffer
    extern __shared__ __attribute__((aligned(16))) float s_col[];           // 2 buffers

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = p.k;
    // XCD-aware block order.  The `cgroups` workgroups that scan different channels of the SAME
    // (column tile, segment, batch) read the same record rows; workgroup q is dispatched to XCD q % 8,
    // so give those workgroups consecutive slots of one XCD: its private L2 then serves the re-reads.
    const int cgroups = (p.D + kColCH * kColWaves - 1) / (kColCH * kColWaves);
    const int coltiles = (p.W + 63) / 64;
    const int ntiles = coltiles * p.nseg * p.B;
    const int q = blockIdx.x, xcd = q & 7, m = q >> 3;
    const int tile = (m / cgroups) * 8 + xcd;
    if (tile >= ntiles) return;
    const int w0 = (tile % coltiles) * 64;
    const int seg = (tile / coltiles) % p.nseg;
    const int b = tile / (coltiles * p.nseg);
    const int d0 = (m % cgroups) * (kColCH * kColWaves) + wv * kColCH;            // wave-uniform
    const int w = w0 + lane;
    const bool colok = w < p.W;
    const int H = p.H, W = p.W;
    const long long L = p.L;
    const int tau_begin = seg * p.chunk_len;
    const int tau_end = min(H, tau_begin + p.chunk_len);

    // wave-uniform per-channel constants (scalar registers)
    v2f A2[kColCH][NP / 2];
    float wdt[kColCH][kRecPad], bias[kColCH], Dd[kColCH];
    bool chok[kColCH];
    // All parameter loads first, unconditional with clamped indices (a `n < N ? load : 0` select compiles to a branch
    // around each load followed by its own wait: 32 serialised round trips before the first scan step).
    float araw[kColCH][NP];
#pragma unroll
    for (int c = 0; c < kColCH; ++c) {
        const int d = d0 + c;
        chok[c] = d < p.D;
        const int kd = k * p.D + (chok[c] ? d : 0);
#pragma unroll
        for (int n = 0; n < NP; ++n) araw[c][n] = p.A_logs[(long long)kd * p.N + min(n, p.N - 1)];
#pragma unroll
        for (int r = 0; r < kRecPad; ++r) wdt[c][r] = p.Wdt[(long long)kd * p.R + min(r, p.R - 1)];
        bias[c] = p.dtb[kd];
        Dd[c] = p.Ds[kd];
    }
#pragma unroll
    for (int c = 0; c < kColCH; ++c) {
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            float a = (n < p.N) ? -expf(araw[c][n]) * 1.4426950408889634f : 0.0f;
            // wave-uniform, but v_pk_mul_f32 cannot take an SGPR pair: pin the value in a VGPR once
            // (otherwise the compiler re-copies SGPR -> VGPR at every use, 16 v_mov per step)
            asm volatile("" : "+v"(a));
            if (n & 1) A2[c][n / 2].y = a; else A2[c][n / 2].x = a;
        }
#pragma unroll
        for (int r = 0; r < kRecPad; ++r) wdt[c][r] = (r < p.R) ? wdt[c][r] : 0.0f;
    }

    const int omega = REV ? W - 1 - w : w;
    const long long chunk = (long long)omega * p.nseg + seg;
    v2f h[kColCH][NP / 2];
    float sum_dt[kColCH];
#pragma unroll
    for (int c = 0; c < kColCH; ++c) {
        sum_dt[c] = 0.0f;
        const long long wsrow = (chunk * p.B * p.D + (long long)b * p.D + d0 + c) * NP;
        if (PHASE == 3 && colok && chok[c] && chunk > 0) {
#pragma unroll
            for (int q = 0; q < NP / 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(p.wsH + wsrow + 4 * q);
                h[c][2 * q] = (v2f){v.x, v.y}; h[c][2 * q + 1] = (v2f){v.z, v.w};
            }
        } else {
#pragma unroll
            for (int n = 0; n < NP / 2; ++n) h[c][n] = splat(0.0f);
        }
    }

    const float* recb = p.rec + ((long long)b * 4 + k) * L * kRS;
    float ur[kColCH][kColT];
    float yr[kColCH][kColT];        // previous directions' y (accumulate mode), fetched with u

    auto row_of = [&](int tau) { return REV ? H - 1 - tau : tau; };
    // u / y: unconditional loads with clamped addresses (a lane outside the map re-reads column W-1, a row past the
    // segment re-reads its last row; neither is ever used or stored): all 64-bit address arithmetic is done once
    // here, a fetch adds one wave-uniform row offset to per-lane bases.
    const int wcl = min(w, W - 1);
    const float* xbase[kColCH];
    const float* ybase[kColCH];
#pragma unroll
    for (int c = 0; c < kColCH; ++c) {
        const long long plane = ((long long)b * p.D + min(d0 + c, p.D - 1)) * H * (long long)W + wcl;
        xbase[c] = p.x + plane;
        ybase[c] = p.y + plane;
    }
    // records: LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write; the destination is wave-uniform
    // base + lane * 16, i.e. the batch image is filled in 1-KiB runs).  A batch is kColT * CH4 runs, run r = row
    // r / CH4, pieces [64 (r % CH4), +64) of that row's 64 * CH4; the wave takes runs wv, wv + kColWaves, ...
    // Piece e of a row is bytes [16 (e % CH4), +16) of the record of column e / CH4 (clamped into the map).
    constexpr int NRUN = kColT * CH4;
    constexpr int RPW = (NRUN + kColWaves - 1) / kColWaves;          // runs per wave
    unsigned pofs[RPW];                                              // the lane's source offset inside a record row (floats)
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int r = min(wv + kColWaves * j, NRUN - 1);
        const int e = (r % CH4) * 64 + lane;
        pofs[j] = (unsigned)(min(e / CH4, W - 1 - w0) * kRS + (e % CH4) * 4);
    }
    auto fetch = [&](int tau0, int buf) {
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int r = wv + kColWaves * j;                        // wave-uniform
            if (NRUN % kColWaves == 0 || r < NRUN) {
                const int row = row_of(min(tau0 + r / CH4, tau_end - 1));
                const float* g = recb + ((long long)row * W + w0) * kRS + pofs[j];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(s_col + buf * BUF + r * 256),
                                                 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < kColT; ++i) {
            const long long roff = (long long)row_of(min(tau0 + i, tau_end - 1)) * W;      // wave-uniform
#pragma unroll
            for (int c = 0; c < kColCH; ++c) {
                ur[c][i] = xbase[c][roff];
                yr[c][i] = (PHASE == 3 && p.accumulate) ? ybase[c][roff] : 0.0f;
            }
        }
    };

    fetch(tau_begin, 0);
    int buf = 0;
    for (int tau0 = tau_begin; tau0 < tau_end; tau0 += kColT, buf ^= 1) {
        float uc[kColCH][kColT], yc[kColCH][kColT];
#pragma unroll
        for (int c = 0; c < kColCH; ++c)
#pragma unroll
            for (int i = 0; i < kColT; ++i) { uc[c][i] = ur[c][i]; yc[c][i] = yr[c][i]; }
        // one barrier per batch: it publishes this batch's image (every wave waited for its own DMA runs) and
        // retires the other buffer (all waves finished the previous batch), which the next fetch overwrites
        __syncthreads();
        if (tau0 + kColT < tau_end) fetch(tau0 + kColT, buf ^ 1);
        const float* s_rec = s_col + buf * BUF;

#pragma unroll
        for (int i = 0; i < kColT; ++i) {
            if (tau0 + i < tau_end) {                                  // uniform
                const float* rc = &s_rec[(i * 64 + lane) * RSL];
                const float4 dr = *reinterpret_cast<const float4*>(rc);
                float4 bq[NP / 4], cq[NP / 4];
#pragma unroll
                for (int r = 0; r < NP / 4; ++r) {
                    bq[r] = *reinterpret_cast<const float4*>(rc + kRecPad + 4 * r);
                    if (PHASE == 3) cq[r] = *reinterpret_cast<const float4*>(rc + kRecPad + NP + 4 * r);
                }
                float dtv[kColCH];
#pragma unroll
                for (int c = 0; c < kColCH; ++c)
                    dtv[c] = fmaf(wdt[c][3], dr.w, fmaf(wdt[c][2], dr.z, fmaf(wdt[c][1], dr.y, fmaf(wdt[c][0], dr.x, bias[c]))));
                static_assert(kColCH == 2, "softplus pairing assumes two channels per wave");
                const v2f sp = softplus2((v2f){dtv[0], dtv[1]});
                dtv[0] = sp.x; dtv[1] = sp.y;
                const int hrow = row_of(tau0 + i);
#pragma unroll
                for (int c = 0; c < kColCH; ++c) {
                    const float dt = dtv[c], ut = uc[c][i];
                    const v2f dt2 = splat(dt), du2 = splat(dt * ut);
                    if (PHASE == 1) sum_dt[c] += dt;
                    v2f y2 = splat(0.0f);
#pragma unroll
                    for (int r = 0; r < NP / 4; ++r) {
                        const v2f a0 = exp2_2(dt2 * A2[c][2 * r]);
                        const v2f a1 = exp2_2(dt2 * A2[c][2 * r + 1]);
                        h[c][2 * r] = a0 * h[c][2 * r] + du2 * (v2f){bq[r].x, bq[r].y};
                        h[c][2 * r + 1] = a1 * h[c][2 * r + 1] + du2 * (v2f){bq[r].z, bq[r].w};
                        if (PHASE == 3) {
                            y2 = (v2f){cq[r].x, cq[r].y} * h[c][2 * r] + y2;
                            y2 = (v2f){cq[r].z, cq[r].w} * h[c][2 * r + 1] + y2;
                        }
                    }
                    if (PHASE == 3 && colok && chok[c]) {
                        float* o = p.y + (((long long)b * p.D + d0 + c) * H + hrow) * W + w;
                        *o = fmaf(Dd[c], ut, y2.x + y2.y) + yc[c][i];
                    }
                }
            }
        }
    }

    if (PHASE == 1 && colok) {
#pragma unroll
        for (int c = 0; c < kColCH; ++c) {
            if (!chok[c]) continue;
            const long long wsrow = (chunk * p.B * p.D + (long long)b * p.D + d0 + c) * NP;
#pragma unroll
            for (int q = 0; q < NP / 4; ++q) {
                *reinterpret_cast<float4*>(p.wsH + wsrow + 4 * q) =
                    make_float4(h[c][2 * q].x, h[c][2 * q].y, h[c][2 * q + 1].x, h[c][2 * q + 1].y);
                const v2f p0 = exp2_2(splat(sum_dt[c]) * A2[c][2 * q]);
                const v2f p1 = exp2_2(splat(sum_dt[c]) * A2[c][2 * q + 1]);
                *reinterpret_cast<float4*>(p.wsP + wsrow + 4 * q) = make_float4(p0.x, p0.y, p1.x, p1.y);
            }
        }
    }
}

}  // namespace wm
