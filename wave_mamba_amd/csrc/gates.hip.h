// gates.hip.h - the element-wise gates and skips of the LFSSBlock TRAINING path with their backward passes, for gfx950.
//
// Reference (/root/reference/basicsr/archs/wavemamba_arch.py): SS2D's output gate `y * F.silu(z)` (:493), the gated ffn
// `F.gelu(x1) * x2` (:228-229), LFSSBlock's two scaled skips `input * skip_scale + ...`, `x * skip_scale2 + ...`
// (:525-526).  PyTorch spells each of them and its autograd as three to six bandwidth-bound launches over (B, 64, H, W)
// planes (plus a full-tensor reduction per skip scale); here each is one launch forward and one backward:
//   gate:      out = act(a) * b                          ga = g * b * act'(a),  gb = g * act(a)      act: SiLU | GELU | sigmoid
//   scale_add: out = x * s[c] + o                        gx = g * s[c],  go = g,  gs[c] = sum_{b,p} g * x
// Exact activations (expf / erff: these feed gradients judged at 1e-4 against a float64 truth).
#pragma once
#include <hip/hip_runtime.h>

namespace wm {

template <int ACT> __device__ __forceinline__ void act_and_grad(float v, float& f, float& df);
template <> __device__ __forceinline__ void act_and_grad<1>(float v, float& f, float& df) {       // SiLU
    const float s = 1.0f / (1.0f + expf(-v));
    f = v * s;
    df = s * (1.0f + v * (1.0f - s));
}
template <> __device__ __forceinline__ void act_and_grad<2>(float v, float& f, float& df) {       // GELU (erf form, nn.GELU default)
    const float c = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    f = v * c;
    df = c + v * 0.39894228040143267794f * expf(-0.5f * v * v);
}

template <> __device__ __forceinline__ void act_and_grad<3>(float v, float& f, float& df) {       // sigmoid (PAConv's pixel gate, :697-699)
    f = 1.0f / (1.0f + expf(-v));
    df = f * (1.0f - f);
}

struct GateArgs {
    const float* a; const float* b; const float* g;      // activation input, multiplied input, upstream gradient (backward)
    float* out; float* ga; float* gb;                    // forward result | the two input gradients
    long long per_b;                                     // elements per batch item (C H W)
    long long sa, sb, sg, so, sga, sgb;                  // batch strides in elements (channel-chunk views: 2 C H W)
};

// grid (ceil(per_b / 1024), B), block (256): four consecutive elements per thread (16-byte accesses when VEC)
template <int ACT, bool BWD, bool VEC>
__global__ __launch_bounds__(256) void gate_kernel(const GateArgs p) {
    const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= p.per_b) return;
    const long long bi = blockIdx.y;
    float av[4], bv[4], gv[4] = {0.f, 0.f, 0.f, 0.f};
    const int n = (int)min(4LL, p.per_b - e);
    if constexpr (VEC) {
        const float4 a4 = *reinterpret_cast<const float4*>(p.a + bi * p.sa + e), b4 = *reinterpret_cast<const float4*>(p.b + bi * p.sb + e);
        av[0] = a4.x; av[1] = a4.y; av[2] = a4.z; av[3] = a4.w; bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
        if constexpr (BWD) {
            const float4 g4 = *reinterpret_cast<const float4*>(p.g + bi * p.sg + e);
            gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            av[j] = j < n ? p.a[bi * p.sa + e + j] : 0.0f; bv[j] = j < n ? p.b[bi * p.sb + e + j] : 0.0f;
            if constexpr (BWD) gv[j] = j < n ? p.g[bi * p.sg + e + j] : 0.0f;
        }
    }
    float r0[4], r1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float f, df;
        act_and_grad<ACT>(av[j], f, df);
        if constexpr (BWD) { r0[j] = gv[j] * bv[j] * df; r1[j] = gv[j] * f; }
        else r0[j] = f * bv[j];
    }
    if constexpr (VEC) {
        if constexpr (BWD) {
            *reinterpret_cast<float4*>(p.ga + bi * p.sga + e) = make_float4(r0[0], r0[1], r0[2], r0[3]);
            *reinterpret_cast<float4*>(p.gb + bi * p.sgb + e) = make_float4(r1[0], r1[1], r1[2], r1[3]);
        } else {
            *reinterpret_cast<float4*>(p.out + bi * p.so + e) = make_float4(r0[0], r0[1], r0[2], r0[3]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < n) {
                if constexpr (BWD) { p.ga[bi * p.sga + e + j] = r0[j]; p.gb[bi * p.sgb + e + j] = r1[j]; }
                else p.out[bi * p.so + e + j] = r0[j];
            }
    }
}

// out = x * s[c] + o over (B C) planes of L elements.  grid (ceil(L / 1024), B C), block (256)
template <bool VEC>
__global__ __launch_bounds__(256) void scale_add_fwd_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                            const float* __restrict__ o, float* __restrict__ out, int C,
                                                            long long L) {
    const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= L) return;
    const long long base = (long long)blockIdx.y * L + e;
    const float sc = s[blockIdx.y % C];
    if constexpr (VEC) {
        const float4 xv = *reinterpret_cast<const float4*>(x + base), ov = *reinterpret_cast<const float4*>(o + base);
        *reinterpret_cast<float4*>(out + base) = make_float4(fmaf(xv.x, sc, ov.x), fmaf(xv.y, sc, ov.y), fmaf(xv.z, sc, ov.z),
                                                             fmaf(xv.w, sc, ov.w));
    } else {
        for (int j = 0; j < 4 && e + j < L; ++j) out[base + j] = fmaf(x[base + j], sc, o[base + j]);
    }
}

// gx = g * s[c];  gs[c] += sum over the block's elements of g * x (wave shuffle + LDS, then ONE atomic per block: the
// scale gradients are sums of ~10^6 products of both signs - their order of accumulation is not fixed by ATen either).
template <bool VEC>
__global__ __launch_bounds__(256) void scale_add_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            const float* __restrict__ s, float* __restrict__ gx,
                                                            float* __restrict__ gs, int C, long long L) {
    // A block walks its plane in steps of gridDim.x * 1024 elements (the host launches at most 8 blocks per plane): one atomic per
    // block - at one block per 1024 elements a 256 x 256 map was 16,384 atomics on the 32 addresses of ONE cache line, and they,
    // not the 200 MB, were the launch (120 us; round 5).
    __shared__ float s_red[4];
    const int c = blockIdx.y % C;
    const float sc = s[c];
    float acc = 0.0f;
    for (long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; e < L; e += (long long)gridDim.x * 1024) {
        const long long base = (long long)blockIdx.y * L + e;
        if constexpr (VEC) {
            const float4 gv = *reinterpret_cast<const float4*>(g + base), xv = *reinterpret_cast<const float4*>(x + base);
            *reinterpret_cast<float4*>(gx + base) = make_float4(gv.x * sc, gv.y * sc, gv.z * sc, gv.w * sc);
            acc += fmaf(gv.x, xv.x, fmaf(gv.y, xv.y, fmaf(gv.z, xv.z, gv.w * xv.w)));
        } else {
            for (int j = 0; j < 4 && e + j < L; ++j) { gx[base + j] = g[base + j] * sc; acc = fmaf(g[base + j], x[base + j], acc); }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(gs + c, (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
}

}  // namespace wm
