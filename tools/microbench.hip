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

// the access pattern of the NCHW streaming kernels: one thread = one position, NPL planes of L floats each
template <int NPL>
__global__ __launch_bounds__(256) void plane_copy_kernel(const float* __restrict__ in, float* __restrict__ out, long long L) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= L) return;
    float v[NPL];
#pragma unroll
    for (int d = 0; d < NPL; ++d) v[d] = in[d * L + p];
#pragma unroll
    for (int d = 0; d < NPL; ++d) out[d * L + p] = v[d];
}
template <int NPL>
__global__ __launch_bounds__(256) void plane_fill_kernel(float* __restrict__ out, long long L) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= L) return;
#pragma unroll
    for (int d = 0; d < NPL; ++d) out[d * L + p] = (float)d;
}
__global__ __launch_bounds__(256) void fill_kernel(float4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void sum_kernel(const float4* __restrict__ in, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (; i < n; i += stride) { const float4 v = in[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) out[0] = s;
}

// the access pattern of the SS2D core's tile loader / storer: a wave moves [64 planes][16 floats] tiles, 16 bytes per
// lane, i.e. 64-byte runs one plane stride apart (PMC calibration of FETCH_SIZE / WRITE_SIZE for that pattern)
__global__ __launch_bounds__(256) void run64_copy_kernel(const float* __restrict__ in, float* __restrict__ out, long long L,
                                                         long long tiles) {
    const int lane = threadIdx.x & 63, trow = lane >> 2, tq = lane & 3;
    long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long stride = (long long)gridDim.x * 4;
    for (; t < tiles; t += stride) {
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(in + (16 * i + trow) * L + 16 * t + 4 * tq);
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(out + (16 * i + trow) * L + 16 * t + 4 * tq) = v[i];
    }
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
    {
        const int blocks = 8192;
        hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, 0, b, bytes / 16);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, 0, b, bytes / 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("float4 fill (write only)       %8.3f ms/iter  %8.1f GB/s\n", ms / 5, 1.0 * bytes * 5 / ms * 1e-6);
        hipLaunchKernelGGL(sum_kernel, dim3(blocks), dim3(256), 0, 0, a, d, bytes / 16);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(sum_kernel, dim3(blocks), dim3(256), 0, 0, a, d, bytes / 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("float4 sum (read only)         %8.3f ms/iter  %8.1f GB/s\n", ms / 5, 1.0 * bytes * 5 / ms * 1e-6);
    }
    {   // 64 planes of 1088 x 1920 floats (UHD level 1, D = 64): 535 MB each way
        const long long L = 1088LL * 1920;
        const int blocks = (int)((L + 255) / 256);
        hipLaunchKernelGGL(plane_copy_kernel<64>, dim3(blocks), dim3(256), 0, 0, (const float*)a, (float*)b, L);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(plane_copy_kernel<64>, dim3(blocks), dim3(256), 0, 0, (const float*)a, (float*)b, L);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("plane copy, 64 planes x 1 dword per lane   %8.3f ms/iter  %8.1f GB/s (read+write)\n", ms / 5, 2.0 * 64 * L * 4 * 5 / ms * 1e-6);
        {
            const long long tiles = L / 16;
            hipLaunchKernelGGL(run64_copy_kernel, dim3(4096), dim3(256), 0, 0, (const float*)a, (float*)b, L, tiles);
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(run64_copy_kernel, dim3(4096), dim3(256), 0, 0, (const float*)a, (float*)b, L, tiles);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("64-byte-run tile copy, 64 planes (core tile pattern) %8.3f ms/iter  %8.1f GB/s (read+write)\n", ms / 5, 2.0 * 64 * L * 4 * 5 / ms * 1e-6);
        }
        hipLaunchKernelGGL(plane_copy_kernel<128>, dim3(blocks), dim3(256), 0, 0, (const float*)a, (float*)b, L);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(plane_copy_kernel<128>, dim3(blocks), dim3(256), 0, 0, (const float*)a, (float*)b, L);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("plane copy, 128 planes x 1 dword per lane  %8.3f ms/iter  %8.1f GB/s (read+write)\n", ms / 5, 2.0 * 128 * L * 4 * 5 / ms * 1e-6);
        hipLaunchKernelGGL(plane_fill_kernel<128>, dim3(blocks), dim3(256), 0, 0, (float*)b, L);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(plane_fill_kernel<128>, dim3(blocks), dim3(256), 0, 0, (float*)b, L);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("plane fill, 128 planes x 1 dword per lane  %8.3f ms/iter  %8.1f GB/s (write only)\n", ms / 5, 128.0 * L * 4 * 5 / ms * 1e-6);
        hipLaunchKernelGGL(plane_copy_kernel<16>, dim3(blocks), dim3(256), 0, 0, (const float*)a, (float*)b, L);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(plane_copy_kernel<16>, dim3(blocks), dim3(256), 0, 0, (const float*)a, (float*)b, L);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("plane copy, 16 planes x 1 dword per lane   %8.3f ms/iter  %8.1f GB/s (read+write)\n", ms / 5, 2.0 * 16 * L * 4 * 5 / ms * 1e-6);
    }
    return 0;
}
