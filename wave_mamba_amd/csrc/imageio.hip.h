// imageio.hip.h - the caller's I/O step around the network as one kernel each way.
//
// Reference: /root/reference/inference_wavemamba.py:99-113 with basicsr/utils/img_util.py - an (h, w, 3) uint8 BGR image
// -> img2tensor (BGR -> RGB, HWC -> CHW, float32) -> / 255. -> reflect-pad bottom / right to multiples of 128 (:28-36);
// and back: crop [:h, :w] -> tensor2img (clamp [0, 1], * 255, round half to even, RGB -> BGR, CHW -> HWC, uint8).
// The PyTorch spelling is 6 passes over the fp32 image (99.5 MB at UHD); here one byte-read + one plane-write pass in,
// one plane-read + one byte-write pass out.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wm {

// out (3, Hp, Wp) fp32 = reflect_pad(chw(img) / 255); grid (ceil(Wp / 256), Hp), block (256)
__global__ __launch_bounds__(256) void image_pre_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int h,
                                                        int w, int Hp, int Wp, int swap_rb) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= Wp) return;
    const int sy = y < h ? y : 2 * (h - 1) - y, sx = x < w ? x : 2 * (w - 1) - x;       // F.pad(..., 'reflect')
    const uint8_t* p = img + ((long long)sy * w + sx) * 3;
    const float c0 = (float)p[0] / 255.0f, c1 = (float)p[1] / 255.0f, c2 = (float)p[2] / 255.0f;
    const long long plane = (long long)Hp * Wp, o = (long long)y * Wp + x;
    out[o] = swap_rb ? c2 : c0;
    out[plane + o] = c1;
    out[2 * plane + o] = swap_rb ? c0 : c2;
}

// img (h, w, 3) uint8 = hwc(round(clamp(in[:, :h, :w], 0, 1) * 255)); grid (ceil(w / 256), h), block (256)
__global__ __launch_bounds__(256) void image_post_kernel(const float* __restrict__ in, uint8_t* __restrict__ img, int h,
                                                         int w, int Hp, int Wp, int swap_rb) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const long long plane = (long long)Hp * Wp, o = (long long)y * Wp + x;
    auto q = [](float v) -> uint8_t {
        v = fminf(fmaxf(v, 0.0f), 1.0f);                   // NaN -> 0 like clamp_ then the uint8 cast of the reference? (inputs are finite)
        return (uint8_t)rintf(v * 255.0f);                 // numpy round: half to even
    };
    const uint8_t c0 = q(in[o]), c1 = q(in[plane + o]), c2 = q(in[2 * plane + o]);
    uint8_t* p = img + ((long long)y * w + x) * 3;
    p[0] = swap_rb ? c2 : c0; p[1] = c1; p[2] = swap_rb ? c0 : c2;
}

}  // namespace wm
