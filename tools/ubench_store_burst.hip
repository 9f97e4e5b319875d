// ubench_store_burst.hip - how fast can ONE compute unit get a convolution tile's epilogue out?
// One workgroup per compute unit, 4 waves, each wave stores a 32-pixel x 4-row x 64-channel block of a
// (64, 1088, 1920) fp32 tensor (131 KB per workgroup, the tile epilogue of conv3x3_ws_kernel<4, 2>) either as
// 128 global_store_dword (the MFMA accumulator layout: lanes 0-31 one 128-byte run, lanes 32-63 the run four planes on)
// or as 32 global_store_dwordx4 (after a 4x4 lane/register transpose: 8 runs of 128 bytes per instruction).
// Prints cycles from the first store to s_waitcnt vmcnt(0), per workgroup (median), with all or a quarter of the
// compute units storing at once.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_store_burst tools/ubench_store_burst.hip && tools/ubench_store_burst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int C = 64, H = 1088, W = 1920;
constexpr long long HW = (long long)H * W;

template <int VEC>
__global__ __launch_bounds__(256) void burst_kernel(float* __restrict__ y, long long* __restrict__ cyc, int active_mod, int reps) {
    if ((int)(blockIdx.x >> 3) % active_mod != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_x = W / 64;
    long long total = 0;
    for (int rep = 0; rep < reps; ++rep) {
        const int tile = (blockIdx.x + rep * gridDim.x) % (tiles_x * (H / 8));
        const int w0 = (tile % tiles_x) * 64 + (wave & 1) * 32, h0 = (tile / tiles_x) * 8 + (wave >> 1) * 4;
        const float v = (float)lane;
        __syncthreads();
        const long long t0 = __builtin_readcyclecounter();
        if (VEC == 1) {
            const int px = lane & 31, khalf = lane >> 5;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int ch = m * 32 + 4 * khalf + (i & 3) + 8 * (i >> 2);
                        y[ch * HW + (long long)(h0 + r) * W + w0 + px] = v;
                    }
        } else {
            const int t = lane & 3, q = (lane >> 2) & 7, khalf = lane >> 5;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ch = m * 32 + 4 * khalf + t + 8 * g;
                        *reinterpret_cast<float4*>(y + ch * HW + (long long)(h0 + r) * W + w0 + 4 * q) = make_float4(v, v, v, v);
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        total += __builtin_readcyclecounter() - t0;
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = total / reps;
}

template <int VEC>
static void run(const char* what, float* y, long long* dcyc, int active_mod) {
    const int grid = 256, reps = 16;
    hipMemset(dcyc, 0, sizeof(long long) * grid);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(burst_kernel<VEC>, dim3(grid), dim3(256), 0, 0, y, dcyc, active_mod, reps);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(burst_kernel<VEC>, dim3(grid), dim3(256), 0, 0, y, dcyc, active_mod, reps);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), dcyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    std::vector<long long> act;
    for (int b = 0; b < grid; ++b) if ((b >> 3) % active_mod == 0) act.push_back(h[b]);
    std::sort(act.begin(), act.end());
    const double bytes = 131072.0 * act.size() * reps;
    printf("%-44s %3zu CUs storing: median %6lld cycles per 131 KB burst (min %lld max %lld); kernel %.3f ms = %.0f GB/s\n",
           what, act.size(), act[act.size() / 2], act.front(), act.back(), ms, bytes / ms / 1e6);
}

int main() {
    float* y; long long* dcyc;
    hipMalloc(&y, sizeof(float) * C * HW);
    hipMalloc(&dcyc, sizeof(long long) * 256);
    run<1>("128 x global_store_dword per wave", y, dcyc, 1);
    run<4>(" 32 x global_store_dwordx4 per wave", y, dcyc, 1);
    run<1>("128 x global_store_dword per wave", y, dcyc, 4);
    run<4>(" 32 x global_store_dwordx4 per wave", y, dcyc, 4);
    run<1>("128 x global_store_dword per wave", y, dcyc, 32);
    run<4>(" 32 x global_store_dwordx4 per wave", y, dcyc, 32);
    return 0;
}
