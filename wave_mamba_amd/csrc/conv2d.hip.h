// conv2d.hip.h - dense 3x3 (stride 1, zero padding 1) and 1x1 convolutions over NCHW fp32 on the bf16 matrix
// cores of gfx950, with fp32-class accuracy from a two-term bf16 split of both operands, and the element-wise
// neighbours of those convolutions in the HFE branch fused into the operand fetch and the epilogue.
//
// Reference call sites (the HFE branch and the U-Net plumbing;
// /root/reference/basicsr/archs/wavemamba_arch.py): PAConv k2 (1x1) / k3 / k4 (:690-697) on
// cat([x, gather(candidates, idx)]) (:666, :713), CMTAttention.qkv / .project_out (:768-771, :797),
// FeedForward.project_in[0] / .project_out[2] (:733-742), DownFRG.l_conv on cat([x_LL, x_d]) (:966, :975),
// upFRG.h_out_conv (:993, :1006), UNet.conv_01 / .last / ps_down* (:1015-1021, :1037).  MIOpen served the 3x3 ones
// with fp32 Winograd kernels (20 ms of the 70 ms UHD step), hipBLASLt the 1x1 ones, each followed by separate
// bias / gate / residual / concatenation kernels (profiles/r01/bench_per_step_kernel_breakdown.txt).
//
// Arithmetic.  x = x_hi + x_lo + r with x_hi = bf16(x), x_lo = bf16(x - x_hi), |r| <= 2^-18 |x|.  The product
// w*x is accumulated in fp32 as w_hi*x_hi + w_hi*x_lo + w_lo*x_hi (three `v_mfma_f32_32x32x16_bf16`); the
// dropped terms are <= 3 * 2^-18 |w*x| per product (measured: 3-4e-6 relative on the output against an fp64
// convolution, tests/test_gpu_parity.py).  The fp32-input MFMA is exact but runs at 1/16 of the bf16 rate:
// 0.98 ms for the UHD level-1 64->64 3x3 convolution at 100 % of peak against 0.19 ms for three bf16 MFMAs.
//
// Implicit GEMM, M = output channels (32 per MFMA row tile), N = 32 pixels of one image row, K = (tap,
// 16 input channels).  D[cout][pixel] puts 32 consecutive pixels of one channel in lanes 0-31 of every
// accumulator register, so NCHW stores are 128-byte runs.  A workgroup (4 waves) owns a (4*RW) x 32 pixel tile
// and 32*MT output channels: wave w owns RW rows.  Per 16-channel chunk the halo tile is fetched from NCHW with
// lanes along W, split into bf16 hi/lo in registers and laid out in LDS as four planes [split][k-half][pixel] of
// 16-byte fragments (a wave's B-operand read is one conflict-free contiguous KiB); the chunk's weights arrive
// pre-split and fragment-ordered (conv2d_prep_kernel) by LDS-DMA.  Every staged row feeds the KS kernel rows
// that touch it (B fragments are read once per kx, not once per tap).  Two workgroups share a compute unit
// (<= 256 registers per wave, <= 80 KB of LDS each): one's fetch / convert / store phases run under the other's
// MFMA phase.  Measured bound (s_memtime phase stamps): the L2 -> L1 fill path, 13-19 B/clk per compute unit for
// this access pattern - a 34-pixel halo row touches 3 cache lines (110 KB of lines per 16-channel chunk for 39 KB of
// pixels) and each workgroup re-fetches the weights.  Tried: a persistent wave-specialised variant (4 loader + 4 MFMA
// waves per compute unit, double-buffered LDS, 16-byte loads): correct, 0.65 ms against 0.62 ms - the loaders stall
// on the same fill path (8.5 k of 11.5 k cycles per chunk in load issue; 0.8 k with the loads stubbed out), so the
// lever is not overlap.  Also tried: output tiles shifted one pixel left so a staged row is one aligned line plus
// two pixels of the previous one (a third fewer input lines, stores straddling two lines): 0.60 ms against 0.56 ms -
// what the misaligned stores cost outweighs the saved fills.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wm {

using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;
using f16x8_t = __attribute__((ext_vector_type(8))) _Float16;
using f32x16_t = __attribute__((ext_vector_type(16))) float;

constexpr int kCvTW = 32;             // tile width in pixels = one MFMA column tile

struct Conv2dArgs {
    const float* xa;                  // input channels [0, Ca)      (B, Ca, H, W)
    const float* xb;                  // input channels [Ca, Ca+Cb): (B, Cbsrc, H, W) or null - cat([xa, xb'], 1) fused
    const int* xb_idx;                // (B, Cb) or null: channel Ca + c is xb[b, xb_idx[b, c]] (torch.gather fused)
    const uint4* wfrag;               // conv2d_prep_kernel output
    const float* bias;                // (Cout) or null
    const float* gate;                // (B, Cout, H, W) or null: out *= sigmoid(gate)
    const float* res;                 // (B, Cout, H, W) or null: out += res
    const uint4* wfrag1;              // G1X1 kernels: prepared 1x1 weights over the same input (PAConv.k2), else unused
    const float* bias1;               // G1X1: bias of that 1x1 (or null)
    float* y;                         // (B, Cout, H, W)
    int Ca, Cb, Cbsrc, Cout, H, W;
    int nch;                          // ceil((Ca + Cb) / 16) input-channel chunks
    int mtot;                         // ceil(Cout / 32) row tiles in wfrag
    int mbase;                        // first row tile of this launch
    const float* amax;                // F16 kernels: device floats {max |input|, max |weight|} (the power-of-two operand scales), else null
    const float* ln_w;                // LNIN kernels (1x1, Ca == 32, no second input): LayerNorm2d over the 32 input channels of a pixel
    const float* ln_b;                //   applied while the pixel is staged: y = conv1x1(weight * (x - mean) / sqrt(var + eps) + bias)
    float ln_eps;
};

union Frag16 {
    bf16x8_t v;
    f16x8_t h;
    uint4 u;
};

// ---- the fp16 form (training): x = x_hi + x_lo with x_hi = fp16(s x), x_lo = fp16(s x - x_hi) carries 22 significant bits per
// operand (bf16: 16) on the same three matrix instructions per product (v_mfma_f32_32x32x16_f16 runs at the bf16 rate).  fp16's
// narrow exponent is met by a power-of-two scale per TENSOR taken from its largest magnitude (a device float: no host
// synchronisation): s max|x| lies in [2^14, 2^15), elements down to 2^-28 of the largest keep a normal hi part, smaller ones
// lose bits gradually (absolute error <= 2^-39 of the largest element).  Products are exact in the fp32 accumulator; the
// epilogue multiplies by 1 / (s_x s_w) - exact.
__device__ __forceinline__ float cv_pow2_scale(float amax) {
    if (!(amax > 0.0f) || !(amax < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(amax, &e);                       // amax = m 2^e, m in [0.5, 1)
    e = 15 - e;
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    return ldexpf(1.0f, e);
}
// |v| of a FINITE v, else 0: the magnitude pass takes its maximum over the finite elements (with an Inf in the tensor
// the scale fell back to 1 and every finite element above 65504 overflowed fp16 too).  A non-finite element itself still
// becomes NaN in every output it touches (hi = Inf, lo = Inf - Inf), where an fp32 convolution would give Inf / NaN.
__device__ __forceinline__ float cv_fabs_fin(float v) { const float a = fabsf(v); return a <= 3.4028234e38f ? a : 0.0f; }
template <bool F16>
__device__ __forceinline__ void cv_split(float v, Frag16& hi, Frag16& lo, int j) {
    if constexpr (F16) {
        const _Float16 hv = (_Float16)v;
        hi.h[j] = hv;
        lo.h[j] = (_Float16)(v - (float)hv);
    } else {
        const __bf16 hv = (__bf16)v;
        hi.v[j] = hv;
        lo.v[j] = (__bf16)(v - (float)hv);
    }
}
template <bool F16>
__device__ __forceinline__ f32x16_t cv_mfma(const Frag16& A, const Frag16& B, f32x16_t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(A.h, B.h, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, B.v, c, 0, 0, 0);
}

// fragment item `idx` of the prepared weights (layout: conv2d_prep_kernel)
template <bool F16>
__device__ __forceinline__ void cv_prep_item(const float* __restrict__ w, uint4* __restrict__ wfrag, long long idx, int Cout, int Cin,
                                             int taps, int mtot, float sw, int dgrad) {
    const int lane = (int)(idx & 63), split = (int)((idx >> 6) & 1);
    long long rest = idx >> 7;
    const int m = (int)(rest % mtot); rest /= mtot;
    const int tap = (int)(rest % taps);
    const int cc = (int)(rest / taps);
    const int co = m * 32 + (lane & 31), ci0 = cc * 16 + (lane >> 5) * 8;
    Frag16 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = ci0 + j;
        // dgrad: `w` is the FORWARD convolution's (Cin, Cout, KS, KS) weight and the fragments are those of its transposed,
        // 180-degree-rotated form w'[co][ci][tap] = w[ci][co][taps - 1 - tap] - the input-gradient convolution's weight, which
        // autograd's formula materialises with a flip and a copy kernel per convolution and step
        const float v = (co < Cout && ci < Cin)
                            ? (dgrad ? w[((long long)ci * Cout + co) * taps + (taps - 1 - tap)] : w[((long long)co * Cin + ci) * taps + tap])
                            : 0.0f;
        Frag16 hi, lo;
        cv_split<F16>(F16 ? v * sw : v, hi, lo, j);
        if constexpr (F16) f.h[j] = split ? lo.h[j] : hi.h[j];
        else f.v[j] = split ? lo.v[j] : hi.v[j];
    }
    wfrag[idx] = f.u;
}

// w (Cout, Cin, KS, KS) fp32 -> wfrag[chunk][tap][m][split][lane] x 8 bf16: lane l of fragment (chunk, tap, m)
// holds w[32 m + (l & 31)][16 chunk + 8 (l >> 5) + j][tap], j = 0..7 (the A-operand layout of
// v_mfma_f32_32x32x16_bf16); channels beyond Cout / Cin are zero.
template <bool F16 = false>
__global__ __launch_bounds__(256) void conv2d_prep_kernel(const float* __restrict__ w, uint4* __restrict__ wfrag,
                                                          int Cout, int Cin, int taps, int nch, int mtot,
                                                          const float* __restrict__ amax = nullptr, int dgrad = 0) {
    const float sw = F16 ? cv_pow2_scale(amax[1]) : 1.0f;
    const long long total = (long long)nch * taps * mtot * 2 * 64;
    for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256)
        cv_prep_item<F16>(w, wfrag, idx, Cout, Cin, taps, mtot, sw, dgrad);
}

// Operand magnitudes + weight preparation of the training step's fp16-split convolutions (88 per BASELINE config-3 step) in ONE launch
// (round 4: a memset node, a magnitude kernel - one atomic per workgroup and at most 512 workgroups: 8,192 wave-level atomics on one
// address had cost 70 us of a 106-us launch over 134 MB - and a preparation kernel):
// the first workgroups take max |x| (one atomicMax each on amax[0], which the caller zeroed); the last
// `nprep` workgroups each take max |w| of the whole weight for themselves (<= 96 x 96 x 9 elements, L2-resident), the first stores
// it to amax[1], and each writes its share of the fragments with the scale it has just found - the preparation needs nothing from
// the pass over x and runs beside it.  (ONE preparing workgroup was 40-90 us of scattered 4-byte loads for the 64 x 64 x 9 weights:
// longer than the pass over x.)
__global__ __launch_bounds__(256) void cv_amax_prep_kernel(const float* __restrict__ x, long long nx, const float* __restrict__ w,
                                                           long long nw, unsigned* __restrict__ amax, uint4* __restrict__ wfrag,
                                                           int Cout, int Cin, int taps, int nch, int mtot, int dgrad, int nprep) {
    __shared__ float s_m[4];
    const int nxb = (int)gridDim.x - nprep;                         // workgroups of the pass over x; the last `nprep` prepare the weights
    const bool wblock = (int)blockIdx.x >= nxb;
    float m = 0.0f;
    if (!wblock) {
        const long long stride = (long long)nxb * 256;
        const bool vec = (reinterpret_cast<size_t>(x) & 15) == 0;
        const long long nq = vec ? nx >> 2 : 0;
        const float4* q = reinterpret_cast<const float4*>(x);
        long long i = (long long)blockIdx.x * 256 + threadIdx.x;
        for (; i + 3 * stride < nq; i += 4 * stride) {               // four loads in flight per thread
            const float4 a = q[i], b = q[i + stride], c = q[i + 2 * stride], d = q[i + 3 * stride];
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(cv_fabs_fin(a.x), cv_fabs_fin(a.y)), fmaxf(cv_fabs_fin(a.z), cv_fabs_fin(a.w))),
                               fmaxf(fmaxf(cv_fabs_fin(b.x), cv_fabs_fin(b.y)), fmaxf(cv_fabs_fin(b.z), cv_fabs_fin(b.w)))));
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(cv_fabs_fin(c.x), cv_fabs_fin(c.y)), fmaxf(cv_fabs_fin(c.z), cv_fabs_fin(c.w))),
                               fmaxf(fmaxf(cv_fabs_fin(d.x), cv_fabs_fin(d.y)), fmaxf(cv_fabs_fin(d.z), cv_fabs_fin(d.w)))));
        }
        for (; i < nq; i += stride) {
            const float4 v = q[i];
            m = fmaxf(fmaxf(m, fmaxf(cv_fabs_fin(v.x), cv_fabs_fin(v.y))), fmaxf(cv_fabs_fin(v.z), cv_fabs_fin(v.w)));
        }
        for (long long e = 4 * nq + (long long)blockIdx.x * 256 + threadIdx.x; e < nx; e += stride) m = fmaxf(m, cv_fabs_fin(x[e]));
    } else {                                                         // every preparing workgroup: max |w| of the WHOLE weight (L2-resident)
        const long long nq = (reinterpret_cast<size_t>(w) & 15) == 0 ? nw >> 2 : 0;
        const float4* q = reinterpret_cast<const float4*>(w);
        for (long long i = threadIdx.x; i < nq; i += 256) {
            const float4 v = q[i];
            m = fmaxf(fmaxf(m, fmaxf(cv_fabs_fin(v.x), cv_fabs_fin(v.y))), fmaxf(cv_fabs_fin(v.z), cv_fabs_fin(v.w)));
        }
        for (long long e = 4 * nq + threadIdx.x; e < nw; e += 256) m = fmaxf(m, cv_fabs_fin(w[e]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    if (!wblock) {
        if (threadIdx.x == 0 && m > 0.0f) atomicMax(amax, __float_as_uint(m));
        return;
    }
    const int pid = (int)blockIdx.x - nxb;
    if (pid == 0 && threadIdx.x == 0) amax[1] = __float_as_uint(m);
    const float sw = cv_pow2_scale(m);
    const long long total = (long long)nch * taps * mtot * 2 * 64;
    for (long long idx = pid * 256ll + threadIdx.x; idx < total; idx += nprep * 256ll)
        cv_prep_item<true>(w, wfrag, idx, Cout, Cin, taps, mtot, sw, dgrad);
}

// G1X1 (3x3 only): a second, 1x1 convolution of the same input (its own prepared weights a.wfrag1 / a.bias1) rides on
// the centre tap's B fragments into a second accumulator set and gates the output: y = conv3x3(X) * sigmoid(conv1x1(X) +
// b1) - PAConv's k3(x) * sigmoid(k2(x)) (reference :694-697) without the gate tensor ever existing.
// LNIN (1x1 only): the convolution's input is LayerNorm2d(x) (reference LayerNorm2d, :532-569: per pixel over the channels, biased
// variance) of a 32-channel map - HFEBlock's norm1 -> attn.qkv and norm2 -> ffn.project_in[0] (:843-851).  A thread stages whole
// pixels (all channels of its PIT pixels pass through its registers chunk by chunk), so the statistics need no other thread:
// both 16-channel chunks are fetched up front, normalised in registers, and the chunk loop stages from them - the LayerNorm
// launch and its 256 B per position (written, read back) are gone.
template <int KS /*1 or 3*/, int RW /*rows per wave*/, int MT /*32-channel row tiles per launch*/, bool G1X1 = false, bool F16 = false,
          bool LNIN = false>
__global__ __launch_bounds__(256, 2) void conv2d_mfma_kernel(const Conv2dArgs a) {
    static_assert(!(F16 && G1X1), "the fp16 form serves the plain convolutions of the training step");
    static_assert(!LNIN || (KS == 1 && !G1X1 && !F16), "LayerNorm2d rides on the inference 1x1");
    const float sx = F16 ? cv_pow2_scale(a.amax[0]) : 1.0f;
    const float osc = F16 ? 1.0f / (sx * cv_pow2_scale(a.amax[1])) : 1.0f;
    extern __shared__ __attribute__((aligned(16))) unsigned char cv_smem[];
    constexpr int PAD = KS / 2, TAPS = KS * KS;
    constexpr int PW = kCvTW + 2 * PAD;              // staged row pitch in pixels
    constexpr int TH = 4 * RW;                       // tile rows per workgroup
    constexpr int NPIX = (TH + 2 * PAD) * PW;        // staged pixels per chunk
    constexpr int PIT = (NPIX + 255) / 256;          // staged pixels per thread (x 2 k-halves x 8 channels)
    constexpr int W_ITEMS = TAPS * MT * 2 * 64;      // 16-byte weight fragments per chunk (a multiple of 64)
    constexpr int W_IT = (W_ITEMS + 255) / 256;
    constexpr int W1_ITEMS = G1X1 ? MT * 2 * 64 : 0; // the 1x1's fragments per chunk
    static_assert(!G1X1 || KS == 3, "the gating 1x1 rides on a 3x3");
    uint4* s_in = reinterpret_cast<uint4*>(cv_smem);                 // [split * 2 + khalf][NPIX]
    uint4* s_w = reinterpret_cast<uint4*>(cv_smem) + 4 * NPIX;       // [tap][m][split][lane]
    uint4* s_w1 = s_w + W_ITEMS;                                     // [m][split][lane]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroup q runs on XCD q mod 8 (each with a private L2): every XCD gets a contiguous band of row-major tiles
    const int tiles_x = (a.W + kCvTW - 1) / kCvTW, tiles_y = (a.H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y, nper = (ntiles + 7) / 8;
    const int tile = (blockIdx.x & 7) * nper + (blockIdx.x >> 3);
    if (tile >= ntiles) return;                       // uniform
    const int w0 = (tile % tiles_x) * kCvTW, h0 = (tile / tiles_x) * TH, b = blockIdx.y;
    const int H = a.H, W = a.W;
    const long long HW = (long long)H * W;
    const float* xa = a.xa + (long long)b * a.Ca * HW;
    const float* xb = a.xb ? a.xb + (long long)b * a.Cbsrc * HW : a.xa;
    const int* xb_idx = a.xb_idx ? a.xb_idx + (long long)b * a.Cb : nullptr;

    // the thread's staged pixels: offset inside a channel plane (0 when outside the image: zero padding) - the
    // same for every chunk, so all per-load address arithmetic is one uniform base plus this 32-bit offset
    unsigned poff[PIT];
    bool pok[PIT];
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
        const int p = tid + it * 256;
        const int pr = p / PW, pc = p - pr * PW;
        const int h = h0 - PAD + pr, w = w0 - PAD + pc;
        pok[it] = p < NPIX && h >= 0 && h < H && w >= 0 && w < W;
        poff[it] = pok[it] ? (unsigned)(h * W + w) : 0u;
    }
    float pin[2][PIT][8];

    auto fetch_in = [&](int cc) {
        // source plane of each of the chunk's 16 channels first (scalar loads of the gather indices, batched), then
        // the vector loads.  Padded channels read plane 0 of xa and are zeroed in stage() (masking here would make
        // the compiler branch around the loads and wait for each one)
        const float* bj[2][8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int c0 = cc * 16 + half * 8;                       // first of the 8 channels (wave-uniform)
            const bool from_a = c0 < a.Ca;
            const int cl = from_a ? c0 : c0 - a.Ca;                  // channel inside its source
            const int cn = (from_a ? a.Ca : a.Cb) - cl;              // channels left in that source (may be <= 0)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool cok = j < cn;
                const int ch = (cok && !from_a && xb_idx) ? xb_idx[cl + j] : cl + j;
                bj[half][j] = cok ? (from_a ? xa : xb) + (long long)ch * HW : xa;
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int it = 0; it < PIT; ++it) pin[half][it][j] = bj[half][j][poff[it]];
    };
    auto fetch_w = [&](int cc) {
        // weights: straight to LDS (LDS-DMA, 1 KiB per wave instruction, no staging registers); destination =
        // wave-uniform base + lane * 16, which is exactly the [tap][m][split][lane] fragment order
        const uint4* wsrc = a.wfrag + ((long long)cc * TAPS * a.mtot) * 128;
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int item0 = it * 256 + wave * 64;                  // wave-uniform
            if (W_ITEMS % 256 == 0 || item0 < W_ITEMS) {
                const int tm = item0 >> 7, tap = tm / MT, m = tm - tap * MT;
                const uint4* g = wsrc + (tap * a.mtot + a.mbase + m) * 128 + (item0 & 64) + lane;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(s_w + item0), 16, 0, 0);
            }
        }
        if constexpr (G1X1) {
            const uint4* w1src = a.wfrag1 + ((long long)cc * a.mtot) * 128;
            const int item0 = wave * 64;                             // MT * 128 fragments: waves 0 .. 2 MT - 1
            if (item0 < W1_ITEMS) {
                const uint4* g = w1src + (a.mbase + (item0 >> 7)) * 128 + (item0 & 64) + lane;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(s_w1 + item0), 16, 0, 0);
            }
        }
    };

    auto stage = [&](int cc) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int c0 = cc * 16 + half * 8;
            const int cn = c0 < a.Ca ? a.Ca - c0 : a.Ca + a.Cb - c0;   // valid channels of this 8-group (uniform)
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int p = tid + it * 256;
                if (NPIX % 256 == 0 || p < NPIX) {
                    Frag16 hi, lo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = (j < cn && pok[it]) ? pin[half][it][j] : 0.0f;
                        cv_split<F16>(F16 ? v * sx : v, hi, lo, j);
                    }
                    s_in[half * NPIX + p] = hi.u;
                    s_in[(2 + half) * NPIX + p] = lo.u;
                }
            }
        }
    };

    f32x16_t acc[MT][RW];
    f32x16_t acc1[G1X1 ? MT : 1][G1X1 ? RW : 1];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[m][r][i] = 0.0f; if (G1X1) acc1[m][r][i] = 0.0f; }

    const int khalf = lane >> 5, px = lane & 31;
    auto mma = [&]() {
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            Frag16 Ah[KS][MT], Al[KS][MT];
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    Ah[ky][m].u = s_w[(((ky * KS + kx) * MT + m) * 2 + 0) * 64 + lane];
                    Al[ky][m].u = s_w[(((ky * KS + kx) * MT + m) * 2 + 1) * 64 + lane];
                }
#pragma unroll
            for (int j = 0; j < RW + 2 * PAD; ++j) {
                const int pidx = (wave * RW + j) * PW + px + kx;
                Frag16 Bh, Bl;
                Bh.u = s_in[khalf * NPIX + pidx];
                Bl.u = s_in[(2 + khalf) * NPIX + pidx];
                // the staged row j feeds output row r = j - ky of kernel row ky; one pass per product term so
                // consecutive MFMAs land on different accumulators
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky) {
                        const int r = j - ky;
                        if (r < 0 || r >= RW) continue;
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            acc[m][r] = cv_mfma<F16>(term == 2 ? Al[ky][m] : Ah[ky][m], term == 1 ? Bl : Bh, acc[m][r]);
                    }
                if constexpr (G1X1) {
                    if (kx == PAD && j >= PAD && j < RW + PAD) {       // centre tap: staged row j is output row j - PAD
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            Frag16 Wh, Wl;
                            Wh.u = s_w1[(m * 2 + 0) * 64 + lane];
                            Wl.u = s_w1[(m * 2 + 1) * 64 + lane];
                            f32x16_t c1 = acc1[m][j - PAD];
                            c1 = cv_mfma<false>(Wh, Bh, c1);
                            c1 = cv_mfma<false>(Wh, Bl, c1);
                            c1 = cv_mfma<false>(Wl, Bh, c1);
                            acc1[m][j - PAD] = c1;
                        }
                    }
                }
            }
        }
    };
    // One output-channel tile per launch leaves the registers to hold the NEXT chunk's input pixels while this chunk's
    // MFMAs run (16 PIT VGPRs): without it every chunk exposes a full global-load round trip between two barriers.
    // The two-tile and gated forms fit it too (244 / 208 VGPRs) but lose more to the registers than they gain
    // (-0.4 % on the whole step); the 64-channel x 16-row form spills with it.
    constexpr bool PIPE = (MT == 1) && !G1X1;
    if constexpr (LNIN) {
        // host contract: Ca == 32, Cb == 0 (nch == 2)
        float xin[2][2][PIT][8];
        fetch_in(0);
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int it = 0; it < PIT; ++it)
#pragma unroll
                for (int j = 0; j < 8; ++j) xin[0][half][it][j] = pin[half][it][j];
        fetch_in(1);
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int it = 0; it < PIT; ++it)
#pragma unroll
                for (int j = 0; j < 8; ++j) xin[1][half][it][j] = pin[half][it][j];
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            float mean = 0.0f;
#pragma unroll
            for (int c = 0; c < 32; ++c) mean += xin[c >> 4][(c >> 3) & 1][it][c & 7];
            mean *= (1.0f / 32.0f);
            float var = 0.0f;
#pragma unroll
            for (int c = 0; c < 32; ++c) { const float q = xin[c >> 4][(c >> 3) & 1][it][c & 7] - mean; var = fmaf(q, q, var); }
            const float rstd = 1.0f / sqrtf(var * (1.0f / 32.0f) + a.ln_eps);
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                float& v = xin[c >> 4][(c >> 3) & 1][it][c & 7];
                v = fmaf((v - mean) * rstd, a.ln_w[c], a.ln_b[c]);
            }
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int it = 0; it < PIT; ++it)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pin[half][it][j] = xin[cc][half][it][j];
            fetch_w(cc);
            stage(cc);
            __syncthreads();
            mma();
            __syncthreads();
        }
    } else if constexpr (PIPE) {
        fetch_in(0);
        fetch_w(0);
        stage(0);
        __syncthreads();
        for (int cc = 0; cc < a.nch; ++cc) {
            const bool more = cc + 1 < a.nch;                        // uniform
            if (more) fetch_in(cc + 1);
            mma();
            __syncthreads();                                         // every wave is done with s_in / s_w
            if (more) {
                fetch_w(cc + 1);
                stage(cc + 1);
                __syncthreads();
            }
        }
    } else {
        for (int cc = 0; cc < a.nch; ++cc) {
            fetch_in(cc);
            fetch_w(cc);
            stage(cc);
            __syncthreads();
            mma();
            __syncthreads();
        }
    }

    // D layout of v_mfma_f32_32x32x*: column = lane & 31 (pixel), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int w = w0 + px;
    const bool full = (a.mbase + MT) * 32 <= a.Cout;                 // uniform: no channel guard on the stores
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int chb = (a.mbase + m) * 32 + 4 * khalf;
        // Epilogue loads are UNCONDITIONAL, from channel indices clamped to Cout - 1 (the stores are guarded): a
        // `ch < Cout ? load : 0` per element is a branch and a full wait per load - 16 bias loads, then 16 gate / residual
        // loads per row, one after the other (tools/isa_load_waits.py: 119 full waits in the 1x1 kernel).
        float bv[16], b1v[G1X1 ? 16 : 1];
        if (a.bias) {
#pragma unroll
            for (int i = 0; i < 16; ++i) bv[i] = a.bias[min(chb + (i & 3) + 8 * (i >> 2), a.Cout - 1)];
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) bv[i] = 0.0f;
        }
        if constexpr (G1X1) {
            if (a.bias1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) b1v[i] = a.bias1[min(chb + (i & 3) + 8 * (i >> 2), a.Cout - 1)];
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) b1v[i] = 0.0f;
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int h = h0 + wave * RW + r;
            if (h >= H || w >= W) continue;
            const long long o = (((long long)b * a.Cout + chb) * H + h) * W + w;
            const long long oplane = ((long long)b * a.Cout * H + h) * W + w;        // channel 0 of this pixel
            // all gate / residual loads of the 16 channels first, then the stores: y may alias neither, but the
            // compiler cannot know and would wait for every load before the store that follows it
            float gv[16], rv[16];
            if (a.gate) {
#pragma unroll
                for (int i = 0; i < 16; ++i) gv[i] = a.gate[oplane + (long long)min(chb + (i & 3) + 8 * (i >> 2), a.Cout - 1) * HW];
            }
            if (a.res) {
#pragma unroll
                for (int i = 0; i < 16; ++i) rv[i] = a.res[oplane + (long long)min(chb + (i & 3) + 8 * (i >> 2), a.Cout - 1) * HW];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int dc = (i & 3) + 8 * (i >> 2);
                float v = (F16 ? acc[m][r][i] * osc : acc[m][r][i]) + bv[i];
                if constexpr (G1X1) v = v / (1.0f + __expf(-(acc1[m][r][i] + b1v[i])));
                if (a.gate) v = v / (1.0f + __expf(-gv[i]));
                if (a.res) v += rv[i];
                if (full || chb + dc < a.Cout) a.y[o + dc * HW] = v;
            }
        }
    }
}

}  // namespace wm
