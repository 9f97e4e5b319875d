// hfe.hip.h - the small-tensor steps of the HFE branch as single kernels.  The reference spells
// each of them as 10-15 PyTorch ops on (B, C, C) / (B, C) tensors (5 us kernels, ~400 launches per UHD forward)
// plus, for SKFF, five full-map passes (stack, sum, mean, multiply, sum).
//
//   match_argmin   channel matching, /root/reference/basicsr/archs/wavemamba_arch.py:659-666 with every channel kept
//                  (match_factor = 1, the only value the U-Net instantiates, :969/:995): idx[b, c] = argmin_j
//                  cdist(x, p)[b, c, j] from the Gram matrix (d^2 = |x_c|^2 + |p_j|^2 - 2 x_c . p_j).
//   attn_fold      transposed attention :787-797: attn = softmax(normalize(q) @ normalize(k)^T * temperature), then
//                  project_out(attn @ v).  attn is (C/heads)^2 per head and project_out is 1x1, so
//                  project_out(attn @ v) = (W_po @ blockdiag(attn)) @ v: the kernel emits that folded C x C weight and
//                  the 1x1 convolution kernel applies it to v (one pass over the map instead of bmm + conv).
//   SKFF           :937-959: channel means of hl + lh + hh -> 1x1 (C -> d) -> PReLU -> three 1x1 (d -> C) -> softmax
//                  over the three bands -> weighted sum.  chansum3 (one read of the bands), skff_weights (tiny),
//                  skff_apply (one read of the bands, one write).
#pragma once
#include <hip/hip_runtime.h>

namespace wm {

// grid (B), block (64): thread c owns row c of the distance matrix.  Ties keep the smallest j.
__global__ __launch_bounds__(64) void match_argmin_kernel(const float* __restrict__ G, const float* __restrict__ nx,
                                                          const float* __restrict__ ny, int* __restrict__ idx, int C) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 64) {
        const float* g = G + ((long long)b * C + c) * C;
        const float xc = nx[b * C + c];
        float best = 3.4e38f;
        int bj = 0;
        for (int j = 0; j < C; ++j) {
            // sqrt and the clamp of the reference (clamp_min(1e-30).sqrt()) are monotone: compare the clamped squares
            const float d2 = fmaxf(xc + ny[b * C + j] - 2.0f * g[j], 1e-30f);
            if (d2 < best) { best = d2; bj = j; }
        }
        idx[b * C + c] = bj;
    }
}

// grid (B), block (256).  G (B*heads, ch, ch), nq/nk (B*heads, ch) squared norms, temperature (heads), Wpo (C, C),
// Wout (B, C, C) with C = heads * ch <= 64.
__global__ __launch_bounds__(256) void attn_fold_kernel(const float* __restrict__ G, const float* __restrict__ nq,
                                                        const float* __restrict__ nk, const float* __restrict__ temperature,
                                                        const float* __restrict__ Wpo, float* __restrict__ Wout, int C,
                                                        int heads) {
    __shared__ float s_attn[64 * 64];               // [h][i][j] packed as [(h * ch + i) * ch + j]
    const int b = blockIdx.x, ch = C / heads;
    for (int row = threadIdx.x; row < C; row += 256) {              // row = h * ch + i
        const int h = row / ch;
        const float* g = G + ((long long)(b * heads + h) * ch + (row - h * ch)) * ch;
        const float qi = fmaxf(sqrtf(nq[(b * heads + h) * ch + (row - h * ch)]), 1e-12f);     // F.normalize eps
        const float t = temperature[h];
        float mx = -3.4e38f;
        for (int j = 0; j < ch; ++j) {
            const float kj = fmaxf(sqrtf(nk[(b * heads + h) * ch + j]), 1e-12f);
            const float v = g[j] / (qi * kj) * t;
            s_attn[row * ch + j] = v;
            mx = fmaxf(mx, v);
        }
        float sum = 0.0f;
        for (int j = 0; j < ch; ++j) { const float e = expf(s_attn[row * ch + j] - mx); s_attn[row * ch + j] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int j = 0; j < ch; ++j) s_attn[row * ch + j] *= inv;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * C; e += 256) {
        const int o = e / C, col = e - o * C, h = col / ch, j = col - h * ch;
        float acc = 0.0f;
        for (int i = 0; i < ch; ++i) acc = fmaf(Wpo[o * C + h * ch + i], s_attn[(h * ch + i) * ch + j], acc);
        Wout[((long long)b * C + o) * C + col] = acc;
    }
}

// part[(b * C + c) * gridDim.x + block] = this block's share of the plane sum of (x0 + x1 + x2)[b, c]; the shares are
// added in block order by skff_weights_kernel (bit-reproducible; atomics were not).  grid (blocks per plane, B * C)
__global__ __launch_bounds__(256) void chansum3_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                       const float* __restrict__ x2, float* __restrict__ part,
                                                       long long HW, bool vec) {
    const long long plane = blockIdx.y;
    const float* p0 = x0 + plane * HW; const float* p1 = x1 + plane * HW; const float* p2 = x2 + plane * HW;
    float acc = 0.0f;
    if (vec) {
        const long long n4 = HW >> 2;
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            const float4 a = reinterpret_cast<const float4*>(p0)[i], b = reinterpret_cast<const float4*>(p1)[i],
                         c = reinterpret_cast<const float4*>(p2)[i];
            acc += ((a.x + b.x) + c.x) + ((a.y + b.y) + c.y) + ((a.z + b.z) + c.z) + ((a.w + b.w) + c.w);
        }
    } else {
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < HW; i += (long long)gridDim.x * 256)
            acc += (p0[i] + p1[i]) + p2[i];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    __shared__ float s_w[4];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[plane * gridDim.x + blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// sums[c] += sum over the batch and the plane of x[b, c]: bias gradients of the convolutions in training (ATen's
// generic reduce kernel spent 41 us per call on it).  grid (blocks per plane, B * C), block (256)
__global__ __launch_bounds__(256) void plane_sums_kernel(const float* __restrict__ x, float* __restrict__ sums, int C,
                                                         long long HW, bool vec) {
    const long long plane = blockIdx.y;
    const float* p0 = x + plane * HW;
    float acc = 0.0f;
    if (vec) {
        const long long n4 = HW >> 2;
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            const float4 a = reinterpret_cast<const float4*>(p0)[i];
            acc += (a.x + a.y) + (a.z + a.w);
        }
    } else {
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) acc += p0[i];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    __shared__ float s_w[4];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sums + (plane % C), (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
}

// grid (B), block (64): block partials of the plane sums (B, C, bpp) -> band weights w (B, 3, C).  Wdu (d, C), prelu (1), Wfc (3, C, d); no biases
// (SKFF is built with bias=False, :943/:947).  C <= 64, d <= 16.
__global__ __launch_bounds__(64) void skff_weights_kernel(const float* __restrict__ part, int bpp,
                                                          const float* __restrict__ Wdu, const float* __restrict__ prelu,
                                                          const float* __restrict__ Wfc, float* __restrict__ w, int C, int d,
                                                          float inv_hw) {
    __shared__ float s_mean[64], s_z[16];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < C) {
        const float* pp = part + ((long long)b * C + t) * bpp;
        float s = 0.0f;
        for (int k = 0; k < bpp; k += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = pp[min(k + j, bpp - 1)];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (k + j < bpp) ? v[j] : 0.0f;
        }
        s_mean[t] = s * inv_hw;
    }
    __syncthreads();
    if (t < d) {
        float z = 0.0f;
        for (int c = 0; c < C; ++c) z = fmaf(Wdu[t * C + c], s_mean[c], z);
        s_z[t] = z >= 0.0f ? z : prelu[0] * z;
    }
    __syncthreads();
    if (t < C) {
        float v[3], mx = -3.4e38f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float a = 0.0f;
            for (int k = 0; k < d; ++k) a = fmaf(Wfc[(i * C + t) * d + k], s_z[k], a);
            v[i] = a; mx = fmaxf(mx, a);
        }
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) { v[i] = expf(v[i] - mx); sum += v[i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) w[(b * 3 + i) * C + t] = v[i] / sum;
    }
}

// out[b, c] = w[b, 0, c] x0[b, c] + w[b, 1, c] x1[b, c] + w[b, 2, c] x2[b, c]; grid (blocks per plane, B * C)
__global__ __launch_bounds__(256) void skff_apply_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                         const float* __restrict__ x2, const float* __restrict__ w,
                                                         float* __restrict__ out, int C, long long HW, bool vec) {
    const long long plane = blockIdx.y;
    const int b = (int)(plane / C), c = (int)(plane - (long long)b * C);
    const float w0 = w[(b * 3 + 0) * C + c], w1 = w[(b * 3 + 1) * C + c], w2 = w[(b * 3 + 2) * C + c];
    const float* p0 = x0 + plane * HW; const float* p1 = x1 + plane * HW; const float* p2 = x2 + plane * HW;
    float* o = out + plane * HW;
    // same association as the reference's (stack * weights).sum(dim=1): ((x0 w0 + x1 w1) + x2 w2), products rounded
    if (vec) {
        const long long n4 = HW >> 2;
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            const float4 a = reinterpret_cast<const float4*>(p0)[i], bb = reinterpret_cast<const float4*>(p1)[i],
                         cc = reinterpret_cast<const float4*>(p2)[i];
            float4 r;
            r.x = __fadd_rn(__fadd_rn(__fmul_rn(a.x, w0), __fmul_rn(bb.x, w1)), __fmul_rn(cc.x, w2));
            r.y = __fadd_rn(__fadd_rn(__fmul_rn(a.y, w0), __fmul_rn(bb.y, w1)), __fmul_rn(cc.y, w2));
            r.z = __fadd_rn(__fadd_rn(__fmul_rn(a.z, w0), __fmul_rn(bb.z, w1)), __fmul_rn(cc.z, w2));
            r.w = __fadd_rn(__fadd_rn(__fmul_rn(a.w, w0), __fmul_rn(bb.w, w1)), __fmul_rn(cc.w, w2));
            reinterpret_cast<float4*>(o)[i] = r;
        }
    } else {
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < HW; i += (long long)gridDim.x * 256)
            o[i] = __fadd_rn(__fadd_rn(__fmul_rn(p0[i], w0), __fmul_rn(p1[i], w1)), __fmul_rn(p2[i], w2));
    }
}

}  // namespace wm
