// haar.hip.h - 2x2 Haar analysis / synthesis kernels for gfx950 (MI355X).
//
// Reference: dwt_init / iwt_init, /root/reference/basicsr/archs/wavemamba_arch.py:97-130.
// Pure streaming kernels (0.5 FLOP/B): the bound is HBM bandwidth.  Each lane moves 16 B per
// access (global_load/store_dwordx4); a thread owns an 8-wide x 2-tall patch of the full-resolution
// map and the matching 4 coefficients of each of the four sub-bands, so every byte is touched once.
//
//   analysis :  full-res map (Tf)  ->  4 sub-bands (Ts)      DWT forward, IWT backward
//   synthesis:  4 sub-bands (Ts)   ->  full-res map (Tf)     IWT forward, DWT backward
//
// The full-res side is dense (planes, 2h, 2w).  A sub-band k lives at  base_k + b*bs_k + c*h*w
// (its own batch stride), which covers both the reference's concatenated (B,4C,h,w) tensor and the
// un-concatenated (x_l, x_h) pair of upFRG.forward (:1006).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wm {

typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_bits_to_float(bf16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}
__device__ __forceinline__ bf16_t float_to_bf16_bits(float f) {   // round-to-nearest-even
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)0x7fc0;   // NaN
    return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
// ROUND = true reproduces eager bf16 tensor arithmetic: every elementwise op rounds to bf16
template <bool ROUND> __device__ __forceinline__ float rnd(float v) {
    if constexpr (ROUND) return bf16_bits_to_float(float_to_bf16_bits(v));
    else return v;
}

// ---- 16-byte (fp32 x4 / bf16 x8) and 8-byte (bf16 x4) vector access ---------------------------
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        w[i] = (uint32_t)float_to_bf16_bits(v[2 * i]) | ((uint32_t)float_to_bf16_bits(v[2 * i + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
    const uint2 a = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
    uint2 a;
    a.x = (uint32_t)float_to_bf16_bits(v[0]) | ((uint32_t)float_to_bf16_bits(v[1]) << 16);
    a.y = (uint32_t)float_to_bf16_bits(v[2]) | ((uint32_t)float_to_bf16_bits(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = a;
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return bf16_bits_to_float(*p); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = float_to_bf16_bits(v); }

// ---- the 2x2 butterflies, in the reference's operation order ------------------------------------
// analysis (:99-108): x1..x4 = a,b,c,d halves;  LL = x1+x2+x3+x4, HL = -x1-x2+x3+x4, ...
template <bool ROUND>
__device__ __forceinline__ void haar_fwd(float a, float b, float c, float d,
                                         float& ll, float& hl, float& lh, float& hh) {
    const float x1 = rnd<ROUND>(a * 0.5f), x2 = rnd<ROUND>(b * 0.5f);
    const float x3 = rnd<ROUND>(c * 0.5f), x4 = rnd<ROUND>(d * 0.5f);
    ll = rnd<ROUND>(rnd<ROUND>(rnd<ROUND>(x1 + x2) + x3) + x4);
    hl = rnd<ROUND>(rnd<ROUND>(rnd<ROUND>(-x1 - x2) + x3) + x4);
    lh = rnd<ROUND>(rnd<ROUND>(rnd<ROUND>(-x1 + x2) - x3) + x4);
    hh = rnd<ROUND>(rnd<ROUND>(rnd<ROUND>(x1 - x2) - x3) + x4);
}
// synthesis (:117-128): a = out[2i,2j], b = out[2i+1,2j], c = out[2i,2j+1], d = out[2i+1,2j+1]
template <bool ROUND>
__device__ __forceinline__ void haar_inv(float s1, float s2, float s3, float s4,
                                         float& a, float& b, float& c, float& d) {
    const float x1 = rnd<ROUND>(s1 * 0.5f), x2 = rnd<ROUND>(s2 * 0.5f);
    const float x3 = rnd<ROUND>(s3 * 0.5f), x4 = rnd<ROUND>(s4 * 0.5f);
    a = rnd<ROUND>(rnd<ROUND>(rnd<ROUND>(x1 - x2) - x3) + x4);
    b = rnd<ROUND>(rnd<ROUND>(rnd<ROUND>(x1 - x2) + x3) - x4);
    c = rnd<ROUND>(rnd<ROUND>(rnd<ROUND>(x1 + x2) - x3) - x4);
    d = rnd<ROUND>(rnd<ROUND>(rnd<ROUND>(x1 + x2) + x3) + x4);
}

struct HaarGeom {
    int C, h, w;          // channels per batch item, sub-band height / width
    long long rows;       // B * C * h  (sub-band rows in total)
    long long bs[4];      // batch stride of each sub-band, in elements
};

// One thread = one sub-band row index (blockIdx.x * blockDim.y + threadIdx.y) x 4 sub-band columns.
// Block (64, 4): a wave spans 64 column groups = 1 KiB contiguous per full-res row per access.
template <typename Tf, typename Ts, bool ROUND, bool VEC>
__global__ __launch_bounds__(256) void haar_analysis_kernel(const Tf* __restrict__ full,
                                                            Ts* __restrict__ s0, Ts* __restrict__ s1,
                                                            Ts* __restrict__ s2, Ts* __restrict__ s3,
                                                            HaarGeom g) {
    const long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y;
    const int cg = blockIdx.y * 64 + threadIdx.x;
    constexpr int CW = VEC ? 4 : 1;                   // sub-band columns per thread
    if (row >= g.rows || cg * CW >= g.w) return;
    const long long plane = row / g.h;
    const int i = (int)(row - plane * g.h);
    const int b = (int)(plane / g.C), c = (int)(plane - (long long)b * g.C);
    const long long hw = (long long)g.h * g.w;
    const int W = 2 * g.w;
    const Tf* r0 = full + (plane * 2 * g.h + 2 * i) * (long long)W + 2 * CW * cg;
    const Tf* r1 = r0 + W;
    const long long so = (long long)c * hw + (long long)i * g.w + CW * cg;
    if constexpr (VEC) {
        float t[8], u[8], o0[4], o1[4], o2[4], o3[4];
        load8(r0, t);
        load8(r1, u);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            haar_fwd<ROUND>(t[2 * j], u[2 * j], t[2 * j + 1], u[2 * j + 1], o0[j], o1[j], o2[j], o3[j]);
        store4(s0 + b * g.bs[0] + so, o0);
        store4(s1 + b * g.bs[1] + so, o1);
        store4(s2 + b * g.bs[2] + so, o2);
        store4(s3 + b * g.bs[3] + so, o3);
    } else {
        float ll, hl, lh, hh;
        haar_fwd<ROUND>(ld1(r0), ld1(r1), ld1(r0 + 1), ld1(r1 + 1), ll, hl, lh, hh);
        st1(s0 + b * g.bs[0] + so, ll);
        st1(s1 + b * g.bs[1] + so, hl);
        st1(s2 + b * g.bs[2] + so, lh);
        st1(s3 + b * g.bs[3] + so, hh);
    }
}

template <typename Ts, typename Tf, bool ROUND, bool VEC>
__global__ __launch_bounds__(256) void haar_synthesis_kernel(const Ts* __restrict__ s0,
                                                             const Ts* __restrict__ s1,
                                                             const Ts* __restrict__ s2,
                                                             const Ts* __restrict__ s3,
                                                             Tf* __restrict__ full, HaarGeom g) {
    const long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y;
    const int cg = blockIdx.y * 64 + threadIdx.x;
    constexpr int CW = VEC ? 4 : 1;
    if (row >= g.rows || cg * CW >= g.w) return;
    const long long plane = row / g.h;
    const int i = (int)(row - plane * g.h);
    const int b = (int)(plane / g.C), c = (int)(plane - (long long)b * g.C);
    const long long hw = (long long)g.h * g.w;
    const int W = 2 * g.w;
    Tf* r0 = full + (plane * 2 * g.h + 2 * i) * (long long)W + 2 * CW * cg;
    Tf* r1 = r0 + W;
    const long long so = (long long)c * hw + (long long)i * g.w + CW * cg;
    if constexpr (VEC) {
        float i0[4], i1[4], i2[4], i3[4], t[8], u[8];
        load4(s0 + b * g.bs[0] + so, i0);
        load4(s1 + b * g.bs[1] + so, i1);
        load4(s2 + b * g.bs[2] + so, i2);
        load4(s3 + b * g.bs[3] + so, i3);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            haar_inv<ROUND>(i0[j], i1[j], i2[j], i3[j], t[2 * j], u[2 * j], t[2 * j + 1], u[2 * j + 1]);
        store8(r0, t);
        store8(r1, u);
    } else {
        float a, bb, cc, d;
        haar_inv<ROUND>(ld1(s0 + b * g.bs[0] + so), ld1(s1 + b * g.bs[1] + so),
                        ld1(s2 + b * g.bs[2] + so), ld1(s3 + b * g.bs[3] + so), a, bb, cc, d);
        st1(r0, a); st1(r1, bb); st1(r0 + 1, cc); st1(r1 + 1, d);
    }
}

}  // namespace wm
