// lfss.hip.h - the per-position glue of LFSSBlock / SS2D around the scan core, fused into three
// streaming kernels for gfx950.
//
// Reference (/root/reference/basicsr/archs/wavemamba_arch.py), per LFSSBlock (:520-528) on tokens
// (B, L, C) with D = 2C inner channels:
//   lfss_in  : ln_1 (:524) -> in_proj C->2D (:483) -> chunk x | z (:484) -> NHWC->NCHW copy (:486)
//   [dwconv3x3 + SiLU (:487), SS2D core (:488), y1+y2+y3+y4 (:490): dwconv.hip.h / ss2d.hip.h]
//   lfss_mid : transpose (:491) -> out_norm (:492) -> * silu(z) (:493) -> out_proj D->C (:494)
//              -> input*skip_scale + . (:525) -> ln_2 -> permute (:526) -> ffn.conv1 1x1 C->2C (:226)
//   [ffn.conv2 depth-wise 3x3 (:226): dwconv.hip.h]
//   lfss_out : gelu(x1)*x2 (:227-228) -> ffn.conv3 1x1 C->C (:230) -> x*skip_scale2 + . (:526)
// In the reference these are ~20 ATen launches per block moving full tensors (LayerNorm, Linear,
// permute/contiguous copies, chunk, mul, add ...).  Here: one thread = one position, every K <= 64
// mat-vec runs out of registers with the weights as scalar (SGPR) operands, LayerNorms are
// in-thread (no cross-lane traffic), planes (B, D, L) are read / written coalesced across the wave,
// token rows (C floats) with 16-byte accesses.  Pure streaming: HBM-bound by construction.
#pragma once
#include <hip/hip_runtime.h>

namespace wm {

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float silu_exact(float v) { return v / (1.0f + expf(-v)); }

template <int C>
__device__ __forceinline__ void load_token(const float* __restrict__ tok, bool nchw, long long b, long long p,
                                           long long L, float (&v)[C]) {
    if (nchw) {
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = tok[(b * C + c) * L + p];
    } else {
        const float4* q = reinterpret_cast<const float4*>(tok + (b * L + p) * C);
#pragma unroll
        for (int c = 0; c < C / 4; ++c) { const float4 t = q[c]; v[4 * c] = t.x; v[4 * c + 1] = t.y; v[4 * c + 2] = t.z; v[4 * c + 3] = t.w; }
    }
}
template <int C>
__device__ __forceinline__ void store_token(float* __restrict__ tok, bool nchw, long long b, long long p,
                                            long long L, const float (&v)[C]) {
    if (nchw) {
#pragma unroll
        for (int c = 0; c < C; ++c) tok[(b * C + c) * L + p] = v[c];
    } else {
        float4* q = reinterpret_cast<float4*>(tok + (b * L + p) * C);
#pragma unroll
        for (int c = 0; c < C / 4; ++c) q[c] = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    }
}
template <int C>
__device__ __forceinline__ void layer_norm(float (&v)[C], const float* __restrict__ w, const float* __restrict__ b,
                                           float eps) {
    float mean = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) mean += v[c];
    mean *= (1.0f / C);
    float var = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) { const float d = v[c] - mean; var = fmaf(d, d, var); }
    const float rstd = rsqrtf(var * (1.0f / C) + eps);
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = fmaf((v[c] - mean) * rstd, w[c], b[c]);
}

// ---- lfss_in: tok -> x (B, D, L), z (B, D, L) ------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void lfss_in_kernel(const float* __restrict__ tok, int tok_nchw,
                                                      const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                      float eps, const float* __restrict__ W_in /*(2D, C)*/,
                                                      float* __restrict__ x, float* __restrict__ z, int B, long long L) {
    constexpr int D = 2 * C;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * L) return;
    const long long b = idx / L, p = idx - b * L;
    float a[C];
    load_token<C>(tok, tok_nchw != 0, b, p, L, a);
    layer_norm<C>(a, ln_w, ln_b, eps);
#pragma unroll
    for (int m0 = 0; m0 < 2 * D; m0 += 16) {
        float o[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < C; ++k) acc = fmaf(W_in[(m0 + m) * C + k], a[k], acc);
            o[m] = acc;
        }
        float* dst = (m0 < D) ? x + (b * D + m0) * L + p : z + (b * D + (m0 - D)) * L + p;
#pragma unroll
        for (int m = 0; m < 16; ++m) dst[(long long)m * L] = o[m];
    }
}

// ---- lfss_mid: ysum, z, tok -> tok1 (B, L, C), f (B, D, L) -------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void lfss_mid_kernel(
    const float* __restrict__ ysum, int ny, long long ystride, const float* __restrict__ z, const float* __restrict__ tok, int tok_nchw,
    const float* __restrict__ on_w, const float* __restrict__ on_b, float on_eps,
    const float* __restrict__ W_out /*(C, D)*/, const float* __restrict__ skip1,
    const float* __restrict__ ln2_w, const float* __restrict__ ln2_b, float ln2_eps,
    const float* __restrict__ W1 /*(D, C)*/, const float* __restrict__ b1,
    float* __restrict__ tok1, float* __restrict__ f, int B, long long L) {
    constexpr int D = 2 * C;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * L) return;
    const long long b = idx / L, p = idx - b * L;
    float y[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float* q = ysum + (b * D + d) * L + p;
        y[d] = ny == 1 ? q[0] : ny == 2 ? q[0] + q[ystride] : ((q[0] + q[ystride]) + q[2 * ystride]) + q[3 * ystride];     // y1 + y2 + y3 + y4 (:490)
    }
    layer_norm<D>(y, on_w, on_b, on_eps);
#pragma unroll
    for (int d = 0; d < D; ++d) y[d] *= silu_exact(z[(b * D + d) * L + p]);
    float t[C];
    load_token<C>(tok, tok_nchw != 0, b, p, L, t);
#pragma unroll
    for (int m = 0; m < C; ++m) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < D; ++k) acc = fmaf(W_out[m * D + k], y[k], acc);
        t[m] = fmaf(t[m], skip1[m], acc);
    }
    store_token<C>(tok1, false, b, p, L, t);
    layer_norm<C>(t, ln2_w, ln2_b, ln2_eps);
#pragma unroll
    for (int m0 = 0; m0 < D; m0 += 16) {
        float o[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            float acc = b1[m0 + m];
#pragma unroll
            for (int k = 0; k < C; ++k) acc = fmaf(W1[(m0 + m) * C + k], t[k], acc);
            o[m] = acc;
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) f[(b * D + m0 + m) * L + p] = o[m];
    }
}

// ---- lfss_out: fc (B, D, L), tok1 -> tok2 --------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void lfss_out_kernel(const float* __restrict__ fc, const float* __restrict__ tok1,
                                                       const float* __restrict__ W3 /*(C, C)*/,
                                                       const float* __restrict__ b3, const float* __restrict__ skip2,
                                                       float* __restrict__ out, int out_nchw, int B, long long L) {
    constexpr int D = 2 * C;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * L) return;
    const long long b = idx / L, p = idx - b * L;
    float g[C];
#pragma unroll
    for (int c = 0; c < C; ++c)
        g[c] = gelu_erf(fc[(b * D + c) * L + p]) * fc[(b * D + C + c) * L + p];
    float t[C];
    load_token<C>(tok1, false, b, p, L, t);
#pragma unroll
    for (int m = 0; m < C; ++m) {
        float acc = b3[m];
#pragma unroll
        for (int k = 0; k < C; ++k) acc = fmaf(W3[m * C + k], g[k], acc);
        t[m] = fmaf(t[m], skip2[m], acc);
    }
    store_token<C>(out, out_nchw != 0, b, p, L, t);
}

// ---- LayerNorm2d: per-pixel LayerNorm over the C channels of an NCHW map (reference :532-569) -----------
// y = w * (x - mean_c) / sqrt(var_c + eps) + b, biased variance - one thread per pixel, planes coalesced.
template <int C>
__global__ __launch_bounds__(256) void layernorm2d_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float eps,
                                                          float* __restrict__ y, int B, long long L) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * L) return;
    const long long bb = idx / L, p = idx - bb * L;
    float v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = x[(bb * C + c) * L + p];
    float mean = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) mean += v[c];
    mean *= (1.0f / C);
    float var = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) { const float d = v[c] - mean; var = fmaf(d, d, var); }
    const float rstd = 1.0f / sqrtf(var * (1.0f / C) + eps);
#pragma unroll
    for (int c = 0; c < C; ++c) y[(bb * C + c) * L + p] = fmaf(w[c], (v[c] - mean) * rstd, b[c]);
}


}  // namespace wm

namespace wm {

// ---- LayerNorm2d backward (reference LayerNormFunction.backward, :545-557) --------------------------------
//   g = gy * w ;  gx = rstd * (g - yhat * mean_c(g * yhat) - mean_c(g)) ;  dw[c] = sum gy*yhat ;  db[c] = sum gy
// one thread = one pixel (mean / rstd / yhat recomputed from x); dw / db partials live in registers over a
// grid-stride loop and leave through a wave shuffle reduction + one atomicAdd per wave and channel.
template <int C>
__global__ __launch_bounds__(256) void layernorm2d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ gy, float eps,
                                                              float* __restrict__ gx, float* __restrict__ dw,
                                                              float* __restrict__ db, int B, long long L) {
    float pw[C], pb[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { pw[c] = 0.0f; pb[c] = 0.0f; }
    const long long total = (long long)B * L;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long bb = idx / L, p = idx - bb * L;
        float v[C], g[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { v[c] = x[(bb * C + c) * L + p]; g[c] = gy[(bb * C + c) * L + p]; }
        float mean = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) mean += v[c];
        mean *= (1.0f / C);
        float var = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) { const float d = v[c] - mean; var = fmaf(d, d, var); }
        const float rstd = 1.0f / sqrtf(var * (1.0f / C) + eps);
        float mg = 0.0f, mgy = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            v[c] = (v[c] - mean) * rstd;                          // yhat
            pw[c] = fmaf(g[c], v[c], pw[c]);
            pb[c] += g[c];
            g[c] *= w[c];
            mg += g[c];
            mgy = fmaf(g[c], v[c], mgy);
        }
        mg *= (1.0f / C); mgy *= (1.0f / C);
#pragma unroll
        for (int c = 0; c < C; ++c) gx[(bb * C + c) * L + p] = rstd * (g[c] - v[c] * mgy - mg);
    }
    __shared__ float s_red[4][2 * C];                  // per-wave sums -> one atomic per block and channel
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float a = pw[c], bsum = pb[c];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); bsum += __shfl_xor(bsum, off); }
        if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6][c] = a; s_red[threadIdx.x >> 6][C + c] = bsum; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * C) {
        const float t = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
        atomicAdd((threadIdx.x < C ? dw : db) + (threadIdx.x % C), t);
    }
}

// ---- LayerNorm2d backward, two lanes per pixel (C = 64: SS2D.out_norm on (B, D, L) planes in the NCHW training path; C = 32:
// LayerNorm2d of the HFE branch) ------------------------------------------------------------------------------------------
// Adjacent lanes (2 p, 2 p + 1) own pixel p: lane half h holds channels [h C / 2, (h + 1) C / 2) of x and gy in registers - read
// ONCE - and the channel sums (mean, variance, the two means of the gradient) cross the pair by one shuffle each.  Round 4's form
// for C = 64 streamed the channels three times with only the parameter-gradient partials in registers (370 us for a 134-MB map:
// 800 MB of cache re-reads); the one-lane-per-pixel form above needs 4 C registers per thread.  Twice the threads per pixel also
// fills the chip on the 64 x 64 maps of BASELINE config 3 (32,768 pixels).
template <int C>
__global__ __launch_bounds__(256) void layernorm2d_bwd_pair_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ gy, float eps,
                                                                   float* __restrict__ gx, float* __restrict__ dw,
                                                                   float* __restrict__ db, int B, long long L) {
    constexpr int CH = C / 2;
    const int h = threadIdx.x & 1;
    float pw[CH], pb[CH], wv[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { pw[c] = 0.0f; pb[c] = 0.0f; wv[c] = w[h * CH + c]; }
    const long long total = (long long)B * L;
    const long long step = ((long long)gridDim.x * blockDim.x) >> 1;
    for (long long idx = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 1; idx < total; idx += step) {   // (the pair shares idx)
        const long long bb = idx / L, p = idx - bb * L;
        const long long base = (bb * C + h * CH) * L + p;
        float v[CH], g[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) { v[c] = x[base + (long long)c * L]; g[c] = gy[base + (long long)c * L]; }
        float mean = 0.0f;
#pragma unroll
        for (int c = 0; c < CH; ++c) mean += v[c];
        mean += __shfl_xor(mean, 1);
        mean *= (1.0f / C);
        float var = 0.0f;
#pragma unroll
        for (int c = 0; c < CH; ++c) { const float d = v[c] - mean; var = fmaf(d, d, var); }
        var += __shfl_xor(var, 1);
        const float rstd = 1.0f / sqrtf(var * (1.0f / C) + eps);
        float mg = 0.0f, mgy = 0.0f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            v[c] = (v[c] - mean) * rstd;                          // yhat
            pw[c] = fmaf(g[c], v[c], pw[c]);
            pb[c] += g[c];
            g[c] *= wv[c];
            mg += g[c];
            mgy = fmaf(g[c], v[c], mgy);
        }
        mg += __shfl_xor(mg, 1); mgy += __shfl_xor(mgy, 1);
        mg *= (1.0f / C); mgy *= (1.0f / C);
#pragma unroll
        for (int c = 0; c < CH; ++c) gx[base + (long long)c * L] = rstd * (g[c] - v[c] * mgy - mg);
    }
    __shared__ float s_red[4][2 * C];                  // per-wave sums -> one atomic per block and channel
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        float a = pw[c], bsum = pb[c];
#pragma unroll
        for (int off = 32; off >= 2; off >>= 1) { a += __shfl_xor(a, off); bsum += __shfl_xor(bsum, off); }   // lanes of one half
        if ((threadIdx.x & 63) < 2) { s_red[threadIdx.x >> 6][h * CH + c] = a; s_red[threadIdx.x >> 6][C + h * CH + c] = bsum; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * C) {
        const float t = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
        atomicAdd((threadIdx.x < C ? dw : db) + (threadIdx.x % C), t);
    }
}

// ---- LayerNorm over the last (contiguous) axis of token tensors (..., C): nn.LayerNorm(C) of LFSSBlock.ln_1 / ln_2
// and SS2D.out_norm (reference :345-386, :493, :522-526), forward and backward for training --------------------------
// C / 4 lanes per token, 4 channels per lane: a wave instruction moves 64 x 16 contiguous bytes; the channel sums run over
// the token's lanes by shuffles.  Backward: dgamma / dbeta partials stay in the lane's 4 registers over a grid-stride loop,
// then lanes with the same channel quad meet by shuffles, waves through LDS, blocks through one atomic per channel.
template <int C>
__device__ __forceinline__ float tok_sum(float v) {
#pragma unroll
    for (int off = 1; off < C / 4; off <<= 1) v += __shfl_xor(v, off);
    return v;
}
template <int C>
__global__ __launch_bounds__(256) void layernorm_tok_kernel(const float4* __restrict__ x, const float4* __restrict__ w,
                                                            const float4* __restrict__ b, float eps,
                                                            float4* __restrict__ y, long long T) {
    constexpr int LPT = C / 4;
    const int q = threadIdx.x % LPT;                                 // channel quad of this lane
    const float4 wv = w[q], bv = b[q];
    const long long stride = (long long)gridDim.x * (256 / LPT);
    for (long long t = (long long)blockIdx.x * (256 / LPT) + threadIdx.x / LPT; t < T; t += stride) {
        const float4 v = x[t * LPT + q];
        const float mean = tok_sum<C>((v.x + v.y) + (v.z + v.w)) * (1.0f / C);
        const float4 d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
        const float var = tok_sum<C>(fmaf(d.x, d.x, fmaf(d.y, d.y, fmaf(d.z, d.z, d.w * d.w)))) * (1.0f / C);
        const float rstd = 1.0f / sqrtf(var + eps);
        y[t * LPT + q] = make_float4(fmaf(d.x * rstd, wv.x, bv.x), fmaf(d.y * rstd, wv.y, bv.y),
                                     fmaf(d.z * rstd, wv.z, bv.z), fmaf(d.w * rstd, wv.w, bv.w));
    }
}
template <int C>
__global__ __launch_bounds__(256) void layernorm_tok_bwd_kernel(const float4* __restrict__ x, const float4* __restrict__ w,
                                                                const float4* __restrict__ gy, float eps,
                                                                float4* __restrict__ gx, float* __restrict__ dw,
                                                                float* __restrict__ db, long long T) {
    constexpr int LPT = C / 4;
    const int q = threadIdx.x % LPT;
    const float4 wv = w[q];
    float4 pw = make_float4(0.f, 0.f, 0.f, 0.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long stride = (long long)gridDim.x * (256 / LPT);
    for (long long t = (long long)blockIdx.x * (256 / LPT) + threadIdx.x / LPT; t < T; t += stride) {
        const float4 v = x[t * LPT + q], g0 = gy[t * LPT + q];
        const float mean = tok_sum<C>((v.x + v.y) + (v.z + v.w)) * (1.0f / C);
        float4 d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
        const float var = tok_sum<C>(fmaf(d.x, d.x, fmaf(d.y, d.y, fmaf(d.z, d.z, d.w * d.w)))) * (1.0f / C);
        const float rstd = 1.0f / sqrtf(var + eps);
        d = make_float4(d.x * rstd, d.y * rstd, d.z * rstd, d.w * rstd);                    // yhat
        pw = make_float4(fmaf(g0.x, d.x, pw.x), fmaf(g0.y, d.y, pw.y), fmaf(g0.z, d.z, pw.z), fmaf(g0.w, d.w, pw.w));
        pb = make_float4(pb.x + g0.x, pb.y + g0.y, pb.z + g0.z, pb.w + g0.w);
        const float4 g = make_float4(g0.x * wv.x, g0.y * wv.y, g0.z * wv.z, g0.w * wv.w);
        const float mg = tok_sum<C>((g.x + g.y) + (g.z + g.w)) * (1.0f / C);
        const float mgy = tok_sum<C>(fmaf(g.x, d.x, fmaf(g.y, d.y, fmaf(g.z, d.z, g.w * d.w)))) * (1.0f / C);
        gx[t * LPT + q] = make_float4(rstd * (g.x - d.x * mgy - mg), rstd * (g.y - d.y * mgy - mg),
                                      rstd * (g.z - d.z * mgy - mg), rstd * (g.w - d.w * mgy - mg));
    }
    float acc[8] = {pw.x, pw.y, pw.z, pw.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int off = LPT; off < 64; off <<= 1) acc[i] += __shfl_xor(acc[i], off);     // lanes with the same quad
    __shared__ float s_red[4][2 * C];
    const int lane = threadIdx.x & 63;
    if (lane < LPT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { s_red[threadIdx.x >> 6][4 * lane + i] = acc[i]; s_red[threadIdx.x >> 6][C + 4 * lane + i] = acc[4 + i]; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * C) {
        const float t = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
        atomicAdd((threadIdx.x < C ? dw : db) + (threadIdx.x % C), t);
    }
}

}  // namespace wm
