// ss2d.hip.h - projection records of the SS2D core for gfx950 (MI355X): the forward half of the core's BACKWARD.
//
// SS2D.forward_core (/root/reference/basicsr/archs/wavemamba_arch.py:446-478) computes per direction k and position
//   x_dbl[k] = x_proj_weight[k] . xs[k]  -> split (dt_r | B | C)              (:453-454)
// The forward (ss2d_core.hip.h) keeps these records in LDS and never writes them; the backward re-runs the projection
// and stores them, because every backward kernel reads them several times:
//     rec[b][k][p] = [dt_r(4) | B(N) | C(N)]   (p = ROW-MAJOR position for every direction: no transposed copies)
//  - ss2d_proj_kernel   (N <= 16): all four directions in one pass, 36-float records, fp32-input MFMA
//    (v_mfma_f32_16x16x4_f32, exact fp32 = an fmaf chain): A = the stacked weights, B = x read straight from NCHW in
//    fragment layout (16 lanes x 8 B = one 128-B line per channel row), D = records written as 16-byte pieces;
//  - ss2d_proj32_kernel (N <= 32): one direction pair per pass, 68-float records.
// (The first-generation forward - separate row / column scan kernels reading these records from HBM - lived here until
// round 3; the second-generation core now takes every shape, odd widths included.)
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

}  // namespace wm
