// ubench_active_cus.hip - is the selective-scan inner loop bound per compute unit (VALU issue) or chip-wide (power / clock)?
// The chunk-scan inner loop of csrc/ss2d_core.hip.h (16 v_exp_f32 + 32 packed fp32 operations per step, B_t / C_t wave-uniform
// from LDS, no HBM traffic), one 1024-thread workgroup per compute unit (100 KB of LDS), on G = 8 .. 256 workgroups:
// time per launch, and the shader clock seen by the waves (s_memtime cycles / 100 MHz wall clock).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_active_cus.hip -o tools/ubench_active_cus
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f ex2(v2f x) { return (v2f){__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }

template <int MODE>   // 0: scan step (exp + packed), 1: packed operations only, 2: exponentials only
__global__ __launch_bounds__(1024) void loop(float* out, unsigned long long* clk, int steps) {
    extern __shared__ float s_rec[];                      // 64 steps of [dt_r(4) | B(16) | C(16)]
    for (int i = threadIdx.x; i < 64 * 36; i += blockDim.x) s_rec[i] = 0.001f * (i % 37) - 0.01f;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    v2f A2[8], h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { A2[i] = (v2f){-1.44f * (2 * i + 1), -1.44f * (2 * i + 2)}; h[i] = (v2f){0.f, 0.f}; }
    float acc = 0.f, u = 0.3f + 1e-3f * (threadIdx.x & 63);
    for (int t = 0; t < steps; ++t) {
        const float* rc = &s_rec[(t & 63) * 36];
        const float dt = 0.01f + 0.02f * rc[0] * u;
        const v2f dt2 = {dt, dt}, du2 = {dt * u, dt * u};
        v2f y2 = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 bv = *reinterpret_cast<const float4*>(rc + 4 + 4 * r);
            v2f a0, a1;
            if (MODE == 1) { a0 = dt2 * A2[2 * r] + 1.0f; a1 = dt2 * A2[2 * r + 1] + 1.0f; }
            else { a0 = ex2(dt2 * A2[2 * r]); a1 = ex2(dt2 * A2[2 * r + 1]); }
            if (MODE == 2) { h[2 * r] = h[2 * r] + a0; h[2 * r + 1] = h[2 * r + 1] + a1; }
            else {
                h[2 * r] = a0 * h[2 * r] + du2 * (v2f){bv.x, bv.y};
                h[2 * r + 1] = a1 * h[2 * r + 1] + du2 * (v2f){bv.z, bv.w};
                const float4 cv = *reinterpret_cast<const float4*>(rc + 20 + 4 * r);
                y2 = (v2f){cv.x, cv.y} * h[2 * r] + y2;
                y2 = (v2f){cv.z, cv.w} * h[2 * r + 1] + y2;
            }
        }
        acc += y2.x + y2.y;
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = acc;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += h[i].x + h[i].y;
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE> void run(const char* name, float* d, unsigned long long* dc, int steps) {
    hipFuncSetAttribute((const void*)loop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    printf("%s, %d steps per wave, 16 waves per workgroup, one workgroup per compute unit\n", name, steps);
    printf("  workgroups    ms/launch   state-steps/ns   shader clock GHz (cycles / 100 MHz wall), min .. max over workgroups\n");
    const int gs[] = {8, 32, 64, 128, 192, 224, 256, 512};
    for (int g : gs) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(loop<MODE>, dim3(g), dim3(1024), 100 * 1024, 0, d, dc, steps);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(loop<MODE>, dim3(g), dim3(1024), 100 * 1024, 0, d, dc, steps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        std::vector<unsigned long long> c(2 * g);
        hipMemcpy(c.data(), dc, sizeof(unsigned long long) * 2 * g, hipMemcpyDeviceToHost);
        double lo = 1e9, hi = 0;
        for (int i = 0; i < g; ++i) { const double ghz = (double)c[2 * i] / ((double)c[2 * i + 1] * 10.0); lo = ghz < lo ? ghz : lo; hi = ghz > hi ? ghz : hi; }
        printf("  %6d      %8.3f    %10.2f        %.2f .. %.2f\n", g, ms, (double)g * 1024 * 16.0 * steps / (ms * 1e6), lo, hi);
    }
}

// waves per SIMD: the same loop with 4, 8, 12, 16 waves per workgroup (1 - 4 per SIMD), 256 workgroups
void occupancy(float* d, unsigned long long* dc, int steps) {
    hipFuncSetAttribute((const void*)loop<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    printf("scan step, 256 workgroups, waves per workgroup (one workgroup per compute unit):\n  waves   ms/launch   state-steps/ns\n");
    for (int w : {4, 8, 12, 16}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(loop<0>, dim3(256), dim3(64 * w), 100 * 1024, 0, d, dc, steps);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(loop<0>, dim3(256), dim3(64 * w), 100 * 1024, 0, d, dc, steps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("  %5d    %8.3f    %10.2f\n", w, ms, 256.0 * 64 * w * 16.0 * steps / (ms * 1e6));
    }
}

int main() {
    float* d; unsigned long long* dc;
    hipMalloc(&d, 512 * 1024 * sizeof(float)); hipMalloc(&dc, 1024 * sizeof(unsigned long long));
    const int steps = 20000;
    occupancy(d, dc, steps);
    run<0>("scan step (16 v_exp_f32 + 32 packed fp32)", d, dc, steps);
    run<1>("packed fp32 only", d, dc, steps);
    run<2>("v_exp_f32 + 1 packed add", d, dc, steps);
    return 0;
}
