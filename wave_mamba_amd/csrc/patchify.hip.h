// patchify.hip.h - the UNet's pixel-unshuffled image inputs, ps_down{1,2,3} = nn.Sequential(nn.PixelUnshuffle(r), nn.Conv2d(r r Cin, wf, 1))
// (/root/reference/basicsr/archs/wavemamba_arch.py:1014-1025, applied at :1043-1045) as ONE kernel per level for gfx950:
//
//   y[b, o, yo, xo] = bias[o] + sum_{c, i, j} w[o, (c r + i) r + j] * img[b, c, r yo + i, r xo + j]            (r = 2, 4, 8)
//
// i.e. an r x r convolution with stride r straight from the image.  The two-op form materialised the unshuffled tensor with a
// strided copy (the full-resolution image read and written once per level: 3 x 0.17 ms per UHD image at 1.2 TB/s) and read it
// back in a 1x1 convolution; here the image is read once per level in 4 r-byte runs per lane (8 / 16 / 2 x 16 bytes: a wave covers
// 64 r consecutive pixels of an image row) and nothing else is written but y.
// HBM-bound by construction (Cin r r <= 192 fp32 FMAs per output value against 4 (Cin r r / CO + 1) bytes per output value moved):
// plain fp32 FMAs in the (c, i, j) order of PixelUnshuffle's channel index - no operand splitting, fp32-exact products.
// lane = output column, CO accumulators per lane; the weights sit in LDS as [k][CO] and are read as wave-uniform (broadcast)
// 16-byte rows; a workgroup walks several row segments so that its 4 CO k r r byte weight copy is amortised.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wm {

template <int R> struct PatchRow;
template <> struct PatchRow<2> { typedef float2 vec; };
template <> struct PatchRow<4> { typedef float4 vec; };
template <> struct PatchRow<8> { typedef float4 vec; };      // two of them

template <int R, int CO>
__global__ __launch_bounds__(256) void patchify_conv_kernel(const float* __restrict__ img, const float* __restrict__ w /*(CO, Cin R R)*/,
                                                            const float* __restrict__ bias /*(CO) or null*/, float* __restrict__ y,
                                                            int B, int Cin, int H, int W, int segs_per_row, long long nsegs, int spb) {
    extern __shared__ __attribute__((aligned(16))) float pw_s[];          // [K][CO]
    const int K = Cin * R * R;
    for (int e = threadIdx.x; e < K * CO; e += 256) {
        const int o = e / K, k = e - o * K;                               // coalesced read of w, transposed into LDS
        pw_s[k * CO + o] = w[e];
    }
    __syncthreads();
    const int Ho = H / R, Wo = W / R;
    const long long plane_o = (long long)Ho * Wo, plane_i = (long long)H * W;
    for (int si = 0; si < spb; ++si) {
        const long long sgi = (long long)blockIdx.x * spb + si;           // segment = 256 consecutive output columns of one output row
        if (sgi >= nsegs) break;
        const long long rowid = sgi / segs_per_row;                       // b * Ho + yo
        const int xo = (int)(sgi - rowid * segs_per_row) * 256 + threadIdx.x;
        const int b = (int)(rowid / Ho), yo = (int)(rowid - (long long)b * Ho);
        const bool ok = xo < Wo;
        const int xc = ok ? xo : Wo - 1;
        float acc[CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] = bias ? bias[o] : 0.0f;
        const float* ip = img + (long long)b * Cin * plane_i + (long long)(R * yo) * W + (long long)R * xc;
        for (int c = 0; c < Cin; ++c) {
            // the R rows of the patch: all loads of a channel in flight before the first FMA
            float v[R][R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const float* rp = ip + (long long)c * plane_i + (long long)i * W;
                if constexpr (R == 2) {
                    const float2 t = *reinterpret_cast<const float2*>(rp);
                    v[i][0] = t.x; v[i][1] = t.y;
                } else {
#pragma unroll
                    for (int q = 0; q < R / 4; ++q) {
                        const float4 t = *reinterpret_cast<const float4*>(rp + 4 * q);
                        v[i][4 * q] = t.x; v[i][4 * q + 1] = t.y; v[i][4 * q + 2] = t.z; v[i][4 * q + 3] = t.w;
                    }
                }
            }
            const float* wk = pw_s + (long long)c * R * R * CO;
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const float xv = v[i][j];
#pragma unroll
                    for (int o4 = 0; o4 < CO / 4; ++o4) {
                        const float4 w4 = *reinterpret_cast<const float4*>(wk + (i * R + j) * CO + 4 * o4);      // wave-uniform address
                        acc[4 * o4] = fmaf(w4.x, xv, acc[4 * o4]); acc[4 * o4 + 1] = fmaf(w4.y, xv, acc[4 * o4 + 1]);
                        acc[4 * o4 + 2] = fmaf(w4.z, xv, acc[4 * o4 + 2]); acc[4 * o4 + 3] = fmaf(w4.w, xv, acc[4 * o4 + 3]);
                    }
                }
        }
        if (ok) {
            float* yp = y + (long long)b * CO * plane_o + (long long)yo * Wo + xo;
#pragma unroll
            for (int o = 0; o < CO; ++o) yp[(long long)o * plane_o] = acc[o];
        }
    }
}

}  // namespace wm
