// ss2d_bwd.hip.h - backward of the fused SS2D core (wm_ss2d_core_bwd): the pieces around the chunked scan backward.
//
// Reference: autograd of SS2D.forward_core (/root/reference/basicsr/archs/wavemamba_arch.py:446-478) - the four
// directional flattenings, the x_proj / dt_proj einsums (:453-455) and the selective scan (:465-471).
//
// Structure (host side in wavemamba_hip.hip):
//   row directions (k = 0, 2) scan the map in its own layout; column directions (k = 1, 3) are the row directions of
//   the TRANSPOSED map, so x and dy are transposed once (transpose_planes_kernel) and everything else is shared:
//   1. ss2d_proj_kernel (forward kernel, re-run) -> records [dt_r | B | C] per position of the layout
//   2. selscan_bwd_{reduce,chunk}_kernel<16, VEC, MODE 1 / 2> (selscan_bwd.hip.h): the chunked adjoint scan with its
//      operand tiles taken from x, dy and the records; dx += du, gradient planes [d dt_r | dB | dC] in x_proj row order,
//      per-chunk partials of dA, dD, dbias, dWdt
//   3. projbwd_dx_kernel: dx[d][p] += sum_k sum_c Wx[k][c][d] g_k[c][p]         (the x_proj einsum, transposed)
//      projgrad_kernel:   dWx[k][c][d] = sum_{b,p} g_k[c][p] x[d][p]             (fp32 MFMA 16x16x4, K = positions)
//   4. selscan_bwd_finish_kernel: dA_logs = dA * A, dDs, d dt_projs_bias, d dt_projs_weight
//   then dx += transpose(dx^T).
#pragma once
#include <hip/hip_runtime.h>

namespace wm {

// out[plane][w][h] = in[plane][h][w]; grid (ceil(W/32), ceil(H/32), planes), block (32, 8)
__global__ __launch_bounds__(256) void transpose_planes_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               int H, int W, int accumulate) {
    __shared__ float tile[32][33];
    const long long plane = blockIdx.z;
    const float* ip = in + plane * (long long)H * W;
    float* op = out + plane * (long long)H * W;
    const int w0 = blockIdx.x * 32, h0 = blockIdx.y * 32;
#pragma unroll
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int h = h0 + j, w = w0 + threadIdx.x;
        tile[j][threadIdx.x] = (h < H && w < W) ? ip[(long long)h * W + w] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int w = w0 + j, h = h0 + threadIdx.x;                 // output row = w, column = h
        if (w < W && h < H) {
            float* o = op + (long long)w * H + h;
            *o = accumulate ? *o + tile[threadIdx.x][j] : tile[threadIdx.x][j];
        }
    }
}

struct ProjBwdArgs {
    const float* g;            // gradient planes [b][kk][CP][L], kk = 0, 1: the two directions of this layout
    const float* x;            // (B, D, L) planes of this layout
    const float* Wx0;          // x_proj_weight of direction kk = 0: (CP, D)
    const float* Wx1;          // ... kk = 1
    float* dx;                 // (B, D, L), accumulated into
    float* dWx0;               // (CP, D), written by projgrad_finish_kernel
    float* dWx1;
    float* part;               // projgrad block partials [kk][block (x, b)][16 RT x 64]
    int B, D, CP;
    long long L;
};

// dx[b][d][p] += sum_kk sum_c Wx_kk[c][d] g[b][kk][c][p].  One thread = one position with its 2 CP gradient values in
// registers; the two weight matrices sit in LDS transposed ([d][kk * CP + c], 16-byte broadcast reads).
// grid (ceil(L / 256), B), block (256).  D <= 64, CP <= CPP = 36 (d_state <= 16) or 68 (d_state <= 32).
template <int CPP>
__global__ __launch_bounds__(256) void projbwd_dx_kernel(const ProjBwdArgs a) {
    constexpr int KP = 2 * CPP;                                      // padded row: both directions
    __shared__ __attribute__((aligned(16))) float s_w[64 * KP];
    for (int e = threadIdx.x; e < 64 * KP; e += 256) {
        const int d = e / KP, q = e - d * KP, kk = q / CPP, c = q - kk * CPP;
        s_w[e] = (d < a.D && c < a.CP) ? (kk ? a.Wx1 : a.Wx0)[(long long)c * a.D + d] : 0.0f;
    }
    __syncthreads();
    const long long p = blockIdx.x * 256ll + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= a.L) return;
    float g[KP];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const float* gp = a.g + (((long long)b * 2 + kk) * a.CP) * a.L + p;
#pragma unroll
        for (int c = 0; c < CPP; ++c) g[kk * CPP + c] = c < a.CP ? gp[(long long)c * a.L] : 0.0f;
    }
    float* o = a.dx + (long long)b * a.D * a.L + p;
    for (int d = 0; d < a.D; ++d) {
        const float4* wr = reinterpret_cast<const float4*>(&s_w[d * KP]);
        float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
        for (int q = 0; q < KP / 4; ++q) {
            const float4 w4 = wr[q];
            acc0 = fmaf(w4.x, g[4 * q], acc0); acc1 = fmaf(w4.y, g[4 * q + 1], acc1);
            acc0 = fmaf(w4.z, g[4 * q + 2], acc0); acc1 = fmaf(w4.w, g[4 * q + 3], acc1);
        }
        o[(long long)d * a.L] += acc0 + acc1;
    }
}

typedef float pg_f4 __attribute__((ext_vector_type(4)));
constexpr int kPgWaves = 8;      // (512 threads: up to 256 registers per lane - the 16 RT x 64 accumulator plus 14 loads in flight)

// dWx_kk[c][d] += sum_p g[b][kk][c][p] x[b][d][p] for a slice of p per wave, on the bf16 matrix cores with both operands
// split into two bf16 terms (hi + lo; the product accumulated in fp32 as hi.hi + hi.lo + lo.hi, ~4e-6 relative per product,
// the same three-product form as the forward's x_proj): K = 32 positions per v_mfma_f32_16x16x32_bf16, lane (row i16, kq)
// feeds positions 8 kq .. 8 kq + 7 of its plane row - two 16-byte loads per operand tile.  Round 2 used the fp32-input
// v_mfma_f32_16x16x4_f32, which runs at the fp32 VECTOR rate: 48 of them per 16 positions made the kernel matrix-bound
// (3.9 ms of a 92-ms BASELINE config-3 training step).  RT x 4 output tiles (16 RT x 64), block-level sum by LDS atomics
// (16 waves per block), then the block's 16 RT x 64 partial goes to the workspace and projgrad_finish_kernel adds the
// blocks up (global atomics from 256 blocks onto the same 2,304 addresses serialise in the memory-side cache - the 8 XCDs'
// L2s do not share lines).  grid (blocks, B, 2), block (64 * kPgWaves).
#ifndef WM_PROJGRAD_F32
#define WM_PROJGRAD_F32 0
#endif
template <int RT /* 16-row tiles of gradient planes: 3 (CP <= 48) or 5 (CP <= 80) */, bool VEC /* L % 4 == 0, 16-byte aligned planes */>
__global__ __launch_bounds__(64 * kPgWaves) void projgrad_kernel(const ProjBwdArgs a, long long slice) {
    __shared__ float s_part[16 * RT * 64];                          // the block's waves meet here by LDS atomics
    for (int e = threadIdx.x; e < 16 * RT * 64; e += 64 * kPgWaves) s_part[e] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y, kk = blockIdx.z;
    const int i16 = lane & 15, kq = lane >> 4;
    const long long wave = (long long)blockIdx.x * kPgWaves + wv;
    const long long l_begin = wave * slice, l_end = min(a.L, l_begin + slice);
    const float* gb = a.g + (((long long)b * 2 + kk) * a.CP) * a.L;
    const float* xb = a.x + (long long)b * a.D * a.L;
    pg_f4 acc[RT][4];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (pg_f4){0.f, 0.f, 0.f, 0.f};
    // The 16-byte form loads UNCONDITIONALLY from a clamped address and zeroes afterwards: an `ok ? load : 0` is a branch
    // around the load with an s_waitcnt vmcnt(0) behind it, i.e. the 14 loads of an iteration one after the other, each
    // waiting out its own latency - the kernel took 122 us at EVERY map size (8 iterations per wave at 64 x 64).
    auto load4 = [&](const float* base, int row, int nrows, long long l) -> float4 {
        if constexpr (VEC) {     // (a compile-time switch: a run-time one is a branch per load again)
            const bool ok = row < nrows && l < l_end;
            const float4 v = *reinterpret_cast<const float4*>(base + (ok ? (long long)row * a.L + l : 0LL));
            return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < nrows) {
            const float* q = base + (long long)row * a.L + l;
            if (l + 0 < l_end) v.x = q[0];
            if (l + 1 < l_end) v.y = q[1];
            if (l + 2 < l_end) v.z = q[2];
            if (l + 3 < l_end) v.w = q[3];
        }
        return v;
    };
#if WM_PROJGRAD_F32
    for (long long l0 = l_begin; l0 < l_end; l0 += 16) {
        const long long l = l0 + 4 * kq;
        float4 ga[RT], xa[4];
#pragma unroll
        for (int i = 0; i < RT; ++i) ga[i] = load4(gb, i16 + 16 * i, a.CP, l);
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[j] = load4(xb, i16 + 16 * j, a.D, l);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gv = c == 0 ? ga[i].x : c == 1 ? ga[i].y : c == 2 ? ga[i].z : ga[i].w;
                    const float xv = c == 0 ? xa[j].x : c == 1 ? xa[j].y : c == 2 ? xa[j].z : xa[j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv, xv, acc[i][j], 0, 0, 0);
                }
    }
#else
    auto split8 = [&](const float4 lo4, const float4 hi4, core_bf8& h, core_bf8& l) {
        const float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            core_bf2 h2, l2;
            core_split2(v[j], v[j + 1], h2, l2);
            h[j] = h2[0]; h[j + 1] = h2[1]; l[j] = l2[0]; l[j + 1] = l2[1];
        }
    };
    for (long long l0 = l_begin; l0 < l_end; l0 += 32) {
        const long long l = l0 + 8 * kq;
        // all 2 (RT + 4) loads of the iteration first, then the splits: written as load-split-load-split the compiler
        // kept that order, with a full wait after every load
        float4 xr[4][2], gr[RT][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) { xr[j][0] = load4(xb, i16 + 16 * j, a.D, l); xr[j][1] = load4(xb, i16 + 16 * j, a.D, l + 4); }
#pragma unroll
        for (int i = 0; i < RT; ++i) { gr[i][0] = load4(gb, i16 + 16 * i, a.CP, l); gr[i][1] = load4(gb, i16 + 16 * i, a.CP, l + 4); }
        __builtin_amdgcn_sched_barrier(0);
        core_bf8 xh[4], xl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split8(xr[j][0], xr[j][1], xh[j], xl[j]);
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            core_bf8 gh, gl;
            split8(gr[i][0], gr[i][1], gh, gl);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl, xh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, xl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, xh[j], acc[i][j], 0, 0, 0);
            }
        }
    }
#endif
    // D layout: lane holds rows 4 kq .. 4 kq + 3 of column i16
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(&s_part[(16 * i + 4 * kq + r) * 64 + 16 * j + i16], acc[i][j][r]);
    __syncthreads();
    float* o = a.part + (((long long)kk * gridDim.y + b) * gridDim.x + blockIdx.x) * (16 * RT * 64);
    for (int e = threadIdx.x; e < 16 * RT * 64; e += 64 * kPgWaves) o[e] = s_part[e];
}
// dWx_kk[c][d] = sum over the nb block partials.  grid (RT * 4, 2), block (256): thread = one element, reading its column
// of the [block][16 RT x 64] array (consecutive threads, consecutive words).
template <int RT>
__global__ __launch_bounds__(256) void projgrad_finish_kernel(const ProjBwdArgs a, int nb) {
    const int e = blockIdx.x * 256 + threadIdx.x, kk = blockIdx.y;
    const float* p = a.part + (long long)kk * nb * (16 * RT * 64) + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = 0;
    for (; i + 3 < nb; i += 4) {
        s0 += p[(long long)i * (16 * RT * 64)]; s1 += p[(long long)(i + 1) * (16 * RT * 64)];
        s2 += p[(long long)(i + 2) * (16 * RT * 64)]; s3 += p[(long long)(i + 3) * (16 * RT * 64)];
    }
    for (; i < nb; ++i) s0 += p[(long long)i * (16 * RT * 64)];
    const int c = e >> 6, d = e & 63;
    if (c < a.CP && d < a.D) (kk ? a.dWx1 : a.dWx0)[(long long)c * a.D + d] = (s0 + s1) + (s2 + s3);
}

// dW[o][i] += sum_t gy[t][o] x[t][i] for token-major operands gy (T, O), x (T, I): the weight gradient of nn.Linear
// (SS2D.in_proj / out_proj, reference :345 / :386; hipBLASLt runs a 32 x 32 tile with K = T ~ 5e5 at 0.5 ms a call).
// fp32 MFMA 16x16x4: lane (r = lane & 15, kq = lane >> 4) feeds gy[t0 + kq][16 a + r] and x[t0 + kq][16 b + r] (64-byte
// runs per token and operand tile), OT x IT output tiles per wave, a slice of tokens per wave, LDS atomics per block, one
// global atomic per element per block.  grid (blocks), block (64 * kLwWaves); O = 16 OT, I = 16 IT.
constexpr int kLwWaves = 8;
// One load of V = min(4, tiles) consecutive channels per lane feeds V row tiles: tile a, lane row r <-> channel
// (a / V) * 16 V + V r + (a % V) - a fixed permutation of the channels inside each group of 16 V.
template <int NT> struct LwVec { static constexpr int V = NT >= 4 ? 4 : (NT >= 2 ? 2 : 1); };
template <int NT>
__device__ __forceinline__ void lw_load(const float* __restrict__ row, int r, bool ok, float (&v)[NT]) {
    constexpr int V = LwVec<NT>::V;
#pragma unroll
    for (int g = 0; g < NT / V; ++g) {
        const float* q = row + g * 16 * V + V * r;
        if constexpr (V == 4) {
            const float4 t = ok ? *reinterpret_cast<const float4*>(q) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[4 * g] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
        } else if constexpr (V == 2) {
            const float2 t = ok ? *reinterpret_cast<const float2*>(q) : make_float2(0.f, 0.f);
            v[2 * g] = t.x; v[2 * g + 1] = t.y;
        } else {
            v[g] = ok ? q[0] : 0.0f;
        }
    }
}
template <int NT> __device__ __forceinline__ int lw_channel(int a, int r) {
    constexpr int V = LwVec<NT>::V;
    return (a / V) * 16 * V + V * r + (a % V);
}
template <int OT, int IT>
__global__ __launch_bounds__(64 * kLwWaves) void linear_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                     float* __restrict__ dW, long long T, long long slice) {
    constexpr int O = 16 * OT, I = 16 * IT;
    __shared__ float s_part[O * I];
    for (int e = threadIdx.x; e < O * I; e += 64 * kLwWaves) s_part[e] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const long long wave = (long long)blockIdx.x * kLwWaves + wv;
    const long long t_begin = wave * slice, t_end = min(T, t_begin + slice);
    pg_f4 acc[OT][IT];
#pragma unroll
    for (int a = 0; a < OT; ++a)
#pragma unroll
        for (int b = 0; b < IT; ++b) acc[a][b] = (pg_f4){0.f, 0.f, 0.f, 0.f};
    for (long long t0 = t_begin; t0 < t_end; t0 += 4) {
        const long long t = t0 + kq;
        const bool ok = t < t_end;
        float ga[OT], xa[IT];
        lw_load<OT>(gy + (ok ? t : 0) * O, r, ok, ga);
        lw_load<IT>(x + (ok ? t : 0) * I, r, ok, xa);
#pragma unroll
        for (int a = 0; a < OT; ++a)
#pragma unroll
            for (int b = 0; b < IT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[a], xa[b], acc[a][b], 0, 0, 0);
    }
    // D layout: lane holds tile rows 4 kq .. 4 kq + 3 (o side) of tile column r (i side)
#pragma unroll
    for (int a = 0; a < OT; ++a)
#pragma unroll
        for (int b = 0; b < IT; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                atomicAdd(&s_part[lw_channel<OT>(a, 4 * kq + q) * I + lw_channel<IT>(b, r)], acc[a][b][q]);
    __syncthreads();
    for (int e = threadIdx.x; e < O * I; e += 64 * kLwWaves) atomicAdd(dW + e, s_part[e]);
}

}  // namespace wm
