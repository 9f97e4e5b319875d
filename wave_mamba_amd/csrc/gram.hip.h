// gram.hip.h - channel Gram matrices over the pixel axis for the HFE branch ("next" row).
//
//   G[b][i][j] = sum_l X[b][i][l] * Y[b][j][l]      nx[b][i] = sum_l X[b][i][l]^2      ny likewise
//
// serves both channel matching (torch.cdist(x, perception) over H*W-long rows, reference
// wavemamba_arch.py:659-666: d^2 = |x|^2 + |y|^2 - 2 x.y) and the transposed attention
// (normalize(q) @ normalize(k)^T, :787-790: G / (|q||k|)).  K = H*W is ~2 M at UHD with M = N = 32:
// a library GEMM runs one tiny tile with a huge K; here every wave owns a slice of l, feeds
// fp32 MFMA 16x16x4 straight from 16-byte row loads (both operands use the SAME lane->(row, l)
// mapping, so no layout shuffling).  The blocks' 32x32 partials go to a workspace and a second small kernel adds
// them in block order: bit-reproducible run to run (atomics were not, once more than two blocks met on an element).
// HBM-bound: 2*C*L*4 bytes read once.
#pragma once
#include <hip/hip_runtime.h>

namespace wm {

typedef float gram_f4 __attribute__((ext_vector_type(4)));

constexpr int kGramWaves = 8;          // waves per block: their 32x32 partials meet in LDS
constexpr int kGramPart = 32 * 32 + 64; // floats per block partial: G (32 x 32), nx (32), ny (32)
// part: [B][gridDim.x][kGramPart]
template <bool VEC /* L % 4 == 0 and 16-byte aligned operands */>
__global__ __launch_bounds__(64 * kGramWaves) void gram32_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                     float* __restrict__ part, int C, long long L, long long slice) {
    const int lane = threadIdx.x & 63;
    __shared__ float s_part[kGramWaves][32 * 32 + 64];
    const int wv = threadIdx.x >> 6;
    const long long wave = (long long)blockIdx.x * kGramWaves + wv;
    const int b = blockIdx.y;
    const int i16 = lane & 15, kq = lane >> 4;
    const long long l_begin = wave * slice, l_end = min(L, l_begin + slice);
    const float* xb = X + (long long)b * C * L;
    const float* yb = Y + (long long)b * C * L;

    gram_f4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = (gram_f4){0.f, 0.f, 0.f, 0.f};
    float sx[2] = {0.f, 0.f}, sy[2] = {0.f, 0.f};

    // 16-byte form: UNCONDITIONAL loads from clamped addresses, zeroed afterwards (an `ok ? load : 0` is a branch around
    // the load with a full wait behind it: the eight loads of an iteration went out one at a time, tools/isa_load_waits.py)
    auto load4 = [&](const float* base, int row, long long l) -> float4 {
        if constexpr (VEC) {
            const bool ok = row < C && l < l_end;
            const float4 v = *reinterpret_cast<const float4*>(base + (ok ? (long long)row * L + l : 0LL));
            return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < C) {
            const float* q = base + (long long)row * L + l;
            if (l + 0 < l_end) v.x = q[0];
            if (l + 1 < l_end) v.y = q[1];
            if (l + 2 < l_end) v.z = q[2];
            if (l + 3 < l_end) v.w = q[3];
        }
        return v;
    };

    // 32 positions per iteration: a lane reads two adjacent 16-byte pieces of a row, the four lanes of a row a whole
    // 128-byte line (the contraction order is free, so positions l0 + 8 kq .. + 7 go to this lane)
    for (long long l0 = l_begin; l0 < l_end; l0 += 32) {
        const long long l = l0 + 8 * kq;
        float4 xa[2][2], ya[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int u = 0; u < 2; ++u) { xa[h][u] = load4(xb, i16 + 16 * h, l + 4 * u); ya[h][u] = load4(yb, i16 + 16 * h, l + 4 * u); }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                sx[h] = fmaf(xa[h][u].x, xa[h][u].x, fmaf(xa[h][u].y, xa[h][u].y, fmaf(xa[h][u].z, xa[h][u].z, fmaf(xa[h][u].w, xa[h][u].w, sx[h]))));
                sy[h] = fmaf(ya[h][u].x, ya[h][u].x, fmaf(ya[h][u].y, ya[h][u].y, fmaf(ya[h][u].z, ya[h][u].z, fmaf(ya[h][u].w, ya[h][u].w, sy[h]))));
            }
            const float xc[2][4] = {{xa[0][u].x, xa[0][u].y, xa[0][u].z, xa[0][u].w}, {xa[1][u].x, xa[1][u].y, xa[1][u].z, xa[1][u].w}};
            const float yc[2][4] = {{ya[0][u].x, ya[0][u].y, ya[0][u].z, ya[0][u].w}, {ya[1][u].x, ya[1][u].y, ya[1][u].z, ya[1][u].w}};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb)
                        acc[a][bb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[a][c], yc[bb][c], acc[a][bb], 0, 0, 0);
        }
    }
    // D layout: lane holds rows 4*kq .. 4*kq+3 (i) of column i16 (j).  Block-level sum in LDS first:
    // one atomic per output element per BLOCK instead of per wave.
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                s_part[wv][(16 * a + 4 * kq + r) * 32 + 16 * bb + i16] = acc[a][bb][r];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float vx = sx[h], vy = sy[h];
        vx += __shfl_xor(vx, 16); vx += __shfl_xor(vx, 32);
        vy += __shfl_xor(vy, 16); vy += __shfl_xor(vy, 32);
        if (kq == 0) { s_part[wv][1024 + i16 + 16 * h] = vx; s_part[wv][1056 + i16 + 16 * h] = vy; }
    }
    __syncthreads();
    float* dst = part + ((long long)b * gridDim.x + blockIdx.x) * kGramPart;
    for (int e = threadIdx.x; e < kGramPart; e += 64 * kGramWaves) {
        float t = 0.0f;
#pragma unroll
        for (int w8 = 0; w8 < kGramWaves; ++w8) t += s_part[w8][e];
        dst[e] = t;
    }
}

// G, nx, ny = sum over the nblk block partials, in block order.  One thread per (element, quarter of the blocks);
// the four quarters meet in LDS in fixed order.
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* __restrict__ part, float* __restrict__ G,
                                                          float* __restrict__ nx, float* __restrict__ ny, int C, int nblk) {
    __shared__ float s_q[4][64];
    const int b = blockIdx.y, el = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    const int per = (nblk + 3) / 4, k0 = min(nblk, q * per), k1 = min(nblk, k0 + per);
    const float* src = part + (long long)b * nblk * kGramPart + e;
    float t = 0.0f;
    for (int k = k0; k < k1; k += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(long long)min(k + j, k1 - 1) * kGramPart];
#pragma unroll
        for (int j = 0; j < 8; ++j) t += (k + j < k1) ? v[j] : 0.0f;
    }
    s_q[q][el] = t;
    __syncthreads();
    if (q != 0) return;
    t = ((s_q[0][el] + s_q[1][el]) + s_q[2][el]) + s_q[3][el];
    if (e < 1024) {
        const int i = e >> 5, j = e & 31;
        if (i < C && j < C) G[((long long)b * C + i) * C + j] = t;
    } else if (e < 1056) {
        if (e - 1024 < C) nx[(long long)b * C + (e - 1024)] = t;
    } else {
        if (e - 1056 < C) ny[(long long)b * C + (e - 1056)] = t;
    }
}

}  // namespace wm
