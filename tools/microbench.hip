// microbench.hip - instruction-rate and HBM-copy probes for MI355X (numbers quoted in DESIGN.md).
// hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + 0.001f * (threadIdx.x + i);
    const float a = 0.999f, b = 0.001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) v[i] = fmaf(v[i], a, b);                               // v_fma_f32
            if (MODE == 1) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.5f - 0.6f;      // v_exp_f32 + fma-ish
            if (MODE == 2) v[i] = __builtin_amdgcn_exp2f(v[i]);                    // v_exp_f32 only
            if (MODE == 3) { v[i] = fmaf(v[i], a, b); v[i] = fmaf(v[i], a, b); v[i] = fmaf(v[i], a, b);
                             v[i] = fmaf(v[i], a, b); v[i] = __builtin_amdgcn_exp2f(v[i] - 1.0f); }  // scan mix 4:1
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

template <int MODE>
static double run_rate(float* d, int iters, const char* name, double ops_per_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * iters * 16 * ops_per_iter;
    printf("%-28s %8.3f ms  %8.2f Tops/s (lane-ops)\n", name, ms, ops / ms * 1e-9);
    return ops / ms * 1e-9;
}

int main() {
    float* d; CK(hipMalloc(&d, 256 * 8 * 256 * sizeof(float)));
    run_rate<0>(d, 4000, "v_fma_f32", 1);
    run_rate<1>(d, 4000, "v_exp_f32 + v_fma", 2);
    run_rate<2>(d, 4000, "v_exp_f32", 1);
    run_rate<3>(d, 2000, "4 fma + 1 exp (scan mix)", 5);
    const size_t bytes = (size_t)2 << 30;
    float4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {2048, 8192, 32768}) {
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, a, b, bytes / 16);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, a, b, bytes / 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("float4 copy, %6d blocks   %8.3f ms/iter  %8.1f GB/s (read+write)\n", blocks, ms / 5, 2.0 * bytes * 5 / ms * 1e-6);
    }
    return 0;
}
