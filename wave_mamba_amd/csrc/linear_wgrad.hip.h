// linear_wgrad.hip.h - two small training-side kernels:
//   transpose_planes_kernel  (B D, H, W) -> (B D, W, H): the fused-core backward (ss2d_core_bwd.hip.h) runs its column directions as
//                            the row directions of the transposed map
//   linear_wgrad_kernel      the weight gradient of a bias-free nn.Linear over tokens (SS2D.in_proj / out_proj, reference
//                            /root/reference/basicsr/archs/wavemamba_arch.py:345, :386)
// (Until round 5 these lived in ss2d_bwd.hip.h next to the first-generation core backward - records through HBM, projbwd_dx,
// projgrad - which round 4's second generation replaced and round 5 deleted.)
#pragma once
#include <hip/hip_runtime.h>

namespace wm {

typedef float pg_f4 __attribute__((ext_vector_type(4)));
constexpr int kRecPad = 4;                 // dt_r slots at the head of a projection record (dt_rank <= 4)

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
