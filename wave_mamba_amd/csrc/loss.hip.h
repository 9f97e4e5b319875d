// loss.hip.h - mean absolute difference of two fp32 tensors, forward and backward, for gfx950.
//
// Reference: the two terms of FeMaSRModel.optimize_parameters (/root/reference/basicsr/models/femasr_model.py:171-179):
// `self.l1 = nn.L1Loss()` on the prediction, and FFTLoss (/root/reference/basicsr/losses/losses.py:306-313) = an L1 mean over the
// stacked real / imaginary parts of rfft2 - the same reduction over the interleaved (re, im) floats of the complex tensor.
//
// Why not ATen's l1_loss: its multi-block reduction zeroes its semaphores with hipMemsetAsync.  Captured into a HIP graph
// (trainer.GraphedTrainStep) that is a memset node, and this runtime fills with a recycled pattern when the graph is launched
// again after eager work (profiles/r06/graph_memset_node.md): the weights stayed right, the REPORTED loss of a replay did not
// (0.33 -> 1.10).  Here: per-thread partial sums, wave shuffle, one atomic per workgroup into a scalar zeroed by a kernel.
#pragma once
#include <hip/hip_runtime.h>

namespace wm {

__global__ __launch_bounds__(256) void l1_mean_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ out, long long n, float inv_n, int vec) {
    const long long stride = (long long)gridDim.x * 256;
    float s = 0.0f;
    if (vec) {
        const float4* a4 = reinterpret_cast<const float4*>(a);
        const float4* b4 = reinterpret_cast<const float4*>(b);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += stride) {
            const float4 x = a4[i], y = b4[i];
            s += (fabsf(x.x - y.x) + fabsf(x.y - y.y)) + (fabsf(x.z - y.z) + fabsf(x.w - y.w));
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) s += fabsf(a[i] - b[i]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, ((part[0] + part[1]) + (part[2] + part[3])) * inv_n);
}

// ga = gout * sign(a - b) / n  (torch: l1_loss_backward; sign(0) = 0)
__global__ __launch_bounds__(256) void l1_mean_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ gout, float* __restrict__ ga, long long n,
                                                          float inv_n, int vec) {
    const long long stride = (long long)gridDim.x * 256;
    const float g = gout[0] * inv_n;
    auto sg = [g](float d) { return d > 0.0f ? g : (d < 0.0f ? -g : 0.0f); };
    if (vec) {
        const float4* a4 = reinterpret_cast<const float4*>(a);
        const float4* b4 = reinterpret_cast<const float4*>(b);
        float4* g4 = reinterpret_cast<float4*>(ga);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += stride) {
            const float4 x = a4[i], y = b4[i];
            g4[i] = make_float4(sg(x.x - y.x), sg(x.y - y.y), sg(x.z - y.z), sg(x.w - y.w));
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) ga[i] = sg(a[i] - b[i]);
    }
}

}  // namespace wm
