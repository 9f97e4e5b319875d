// ubench_exp_overlap.hip - does the order of v_exp_f32 and packed-fp32 instructions inside a scan step matter?
// The chunk-scan step of csrc/ss2d_core.hip.h per lane (= channel) and state pair: a = exp2(dt A); h = a h + (dt u) B;
// y += C h  - 16 v_exp_f32 and ~32 packed fp32 operations per step.  v_exp_f32 runs on the transcendental unit
// (3.1 x the issue time of an FMA when alone, tools/microbench); the question is how much of it hides under the packed
// operations, and whether the instruction order decides that:
//   V0  the order of the product kernel (per group of four states: exponentials, then the FMAs)
//   V1  all 16 exponentials of the step first, then all FMAs
//   V2  software-pipelined: the exponentials of step t + 1 are issued between the FMAs of step t (one v_exp, three
//       packed operations, ...), enforced with sched_group_barrier
//   E   exponentials only      F   packed operations only
// Reported: cycles per step per wave-slot at 4 waves per SIMD (s_memtime over the loop / steps).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_exp_overlap tools/ubench_exp_overlap.hip && tools/ubench_exp_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f ex2(v2f x) { return (v2f){__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }

template <int V>
__global__ __launch_bounds__(256) void step_kernel(float* out, long long* cyc, int steps) {
    __shared__ float s_rec[64 * 36];                      // 64 steps of [dt_r(4) | B(16) | C(16)]
    for (int i = threadIdx.x; i < 64 * 36; i += 256) s_rec[i] = 0.001f * (i % 37) - 0.01f;
    __syncthreads();
    v2f A2[8], h[8], an[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { A2[i] = (v2f){-1.44f * (2 * i + 1), -1.44f * (2 * i + 2)}; h[i] = (v2f){0.f, 0.f}; }
    float acc = 0.f, u = 0.3f + 1e-3f * threadIdx.x;
    {
        const float dt0 = 0.01f + 0.02f * s_rec[0] * u;
#pragma unroll
        for (int i = 0; i < 8; ++i) an[i] = ex2((v2f){dt0, dt0} * A2[i]);
    }
    const long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < steps; ++t) {
        const float* rc = &s_rec[(t & 63) * 36];
        const float* rn = &s_rec[((t + 1) & 63) * 36];
        const float dt = 0.01f + 0.02f * rc[0] * u;
        const float dtn = 0.01f + 0.02f * rn[0] * u;
        const v2f dt2 = {dt, dt}, du2 = {dt * u, dt * u}, dtn2 = {dtn, dtn};
        v2f y2 = {0.f, 0.f};
        if (V == 1) {
            v2f a[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = ex2(dt2 * A2[i]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 bv = *reinterpret_cast<const float4*>(rc + 4 + 4 * r);
                const float4 cv = *reinterpret_cast<const float4*>(rc + 20 + 4 * r);
                h[2 * r] = a[2 * r] * h[2 * r] + du2 * (v2f){bv.x, bv.y};
                h[2 * r + 1] = a[2 * r + 1] * h[2 * r + 1] + du2 * (v2f){bv.z, bv.w};
                y2 = (v2f){cv.x, cv.y} * h[2 * r] + y2;
                y2 = (v2f){cv.z, cv.w} * h[2 * r + 1] + y2;
            }
        } else if (V == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 bv = *reinterpret_cast<const float4*>(rc + 4 + 4 * r);
                const float4 cv = *reinterpret_cast<const float4*>(rc + 20 + 4 * r);
                const v2f a0 = an[2 * r], a1 = an[2 * r + 1];
                an[2 * r] = ex2(dtn2 * A2[2 * r]);                 // next step's exponentials, between this step's FMAs
                an[2 * r + 1] = ex2(dtn2 * A2[2 * r + 1]);
                h[2 * r] = a0 * h[2 * r] + du2 * (v2f){bv.x, bv.y};
                h[2 * r + 1] = a1 * h[2 * r + 1] + du2 * (v2f){bv.z, bv.w};
                y2 = (v2f){cv.x, cv.y} * h[2 * r] + y2;
                y2 = (v2f){cv.z, cv.w} * h[2 * r + 1] + y2;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);     // one transcendental
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     // two other VALU operations
                }
            }
        } else if (V == 3) {                               // exponentials only
#pragma unroll
            for (int i = 0; i < 8; ++i) { const v2f a = ex2(dt2 * A2[i]); y2 += a; }
        } else if (V == 4) {                               // packed operations only
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 bv = *reinterpret_cast<const float4*>(rc + 4 + 4 * r);
                const float4 cv = *reinterpret_cast<const float4*>(rc + 20 + 4 * r);
                const v2f a0 = dt2 * A2[2 * r], a1 = dt2 * A2[2 * r + 1];
                h[2 * r] = a0 * h[2 * r] + du2 * (v2f){bv.x, bv.y};
                h[2 * r + 1] = a1 * h[2 * r + 1] + du2 * (v2f){bv.z, bv.w};
                y2 = (v2f){cv.x, cv.y} * h[2 * r] + y2;
                y2 = (v2f){cv.z, cv.w} * h[2 * r + 1] + y2;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 bv = *reinterpret_cast<const float4*>(rc + 4 + 4 * r);
                const v2f a0 = ex2(dt2 * A2[2 * r]), a1 = ex2(dt2 * A2[2 * r + 1]);
                h[2 * r] = a0 * h[2 * r] + du2 * (v2f){bv.x, bv.y};
                h[2 * r + 1] = a1 * h[2 * r + 1] + du2 * (v2f){bv.z, bv.w};
                const float4 cv = *reinterpret_cast<const float4*>(rc + 20 + 4 * r);
                y2 = (v2f){cv.x, cv.y} * h[2 * r] + y2;
                y2 = (v2f){cv.z, cv.w} * h[2 * r + 1] + y2;
            }
        }
        acc += y2.x + y2.y;
    }
    const long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += h[i].x + h[i].y + an[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
static void run(const char* what, float* out, long long* cyc) {
    const int grid = 1024, steps = 4096;                  // 4 workgroups = 16 waves per compute unit = 4 per SIMD
    hipLaunchKernelGGL(step_kernel<V>, dim3(grid), dim3(256), 0, 0, out, cyc, steps);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(step_kernel<V>, dim3(grid), dim3(256), 0, 0, out, cyc, steps);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < grid; ++i) m += (double)h[i];
    m /= grid;
    printf("%-64s %7.1f cycles per step per wave (4 waves / SIMD: %6.1f per SIMD-step), kernel %.3f ms\n", what, m / steps, m / steps / 4, ms);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * 1024 * 256); hipMalloc(&cyc, sizeof(long long) * 1024);
    run<0>("V0 product order (per four states: exp, then FMAs)", out, cyc);
    run<1>("V1 all 16 exponentials, then all FMAs", out, cyc);
    run<2>("V2 next step's exponentials between this step's FMAs", out, cyc);
    run<3>("E  exponentials only", out, cyc);
    run<4>("F  packed operations only", out, cyc);
    return 0;
}
