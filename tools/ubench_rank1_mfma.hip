// ubench_rank1_mfma.hip - the scan step's rank-1 update h[n][ch] += B_t[n] * (dt u)[ch] on the matrix pipe?
// In the lane = channel layout the update of four states is exactly one v_mfma_f32_4x4x1_16b_f32: block = lane / 4, column
// j = lane % 4 (the lane's own channel), row i = register (state 4 g + i): D[i][j] = A[i] * B[j] + C[i][j] with
// A-operand lane m = B_t[4 g + m % 4] (wave-uniform record, read with a lane-dependent offset), B-operand = dt u (the
// lane's own), C = a * h (packed multiply on the VALU).  It takes the 8 packed FMAs per step off the VALU (of 32 packed +
// 16 v_exp_f32) and puts four 2-pass matrix instructions beside them.
//   MODE 0: the shipped step   MODE 1: rank-1 update on v_mfma_f32_4x4x1_16b_f32
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_rank1_mfma.hip -o tools/ubench_rank1_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v2f ex2(v2f x) { return (v2f){__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }

template <int MODE, bool SCAN>
__global__ __launch_bounds__(1024) void loop(float* out, int steps) {
    extern __shared__ float s_rec[];                      // 64 steps of [dt_r(4) | B(16) | C(16)]
    for (int i = threadIdx.x; i < 64 * 36; i += blockDim.x) s_rec[i] = 0.001f * (i % 37) - 0.01f;
    __syncthreads();
    v2f A2[8];
    v4f h[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) A2[i] = (v2f){-1.44f * (2 * i + 1), -1.44f * (2 * i + 2)};
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    float acc = 0.f, u = 0.3f + 1e-3f * (threadIdx.x & 63);
    const int l4 = threadIdx.x & 3;
    for (int t = 0; t < steps; ++t) {
        const float* rc = &s_rec[(t & 63) * 36];
        const float dt = 0.01f + 0.02f * rc[0] * u;
        const v2f dt2 = {dt, dt};
        const float du = dt * u;
        const v2f du2 = {du, du};
        v2f y2 = {0.f, 0.f};
        float4 bq;
        if (MODE == 1) bq = *reinterpret_cast<const float4*>(rc + 4 + 4 * l4);
        const float bA[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const v2f a0 = ex2(dt2 * A2[2 * r]), a1 = ex2(dt2 * A2[2 * r + 1]);
            v2f h0 = {h[r].x, h[r].y}, h1 = {h[r].z, h[r].w};
            if (MODE == 0) {
                const float4 bv = *reinterpret_cast<const float4*>(rc + 4 + 4 * r);
                h0 = a0 * h0 + du2 * (v2f){bv.x, bv.y};
                h1 = a1 * h1 + du2 * (v2f){bv.z, bv.w};
                h[r] = (v4f){h0.x, h0.y, h1.x, h1.y};
            } else {
                h0 = a0 * h0; h1 = a1 * h1;
                h[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(bA[r], du, (v4f){h0.x, h0.y, h1.x, h1.y}, 0, 0, 0);
            }
            if (SCAN) {
                const float4 cv = *reinterpret_cast<const float4*>(rc + 20 + 4 * r);
                y2 = (v2f){cv.x, cv.y} * (v2f){h[r].x, h[r].y} + y2;
                y2 = (v2f){cv.z, cv.w} * (v2f){h[r].z, h[r].w} + y2;
            }
        }
        acc += y2.x + y2.y;
    }
    float s = acc;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += h[i].x + h[i].y + h[i].z + h[i].w;
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int MODE, bool SCAN> void run(const char* name, float* d, int steps) {
    hipFuncSetAttribute((const void*)loop<MODE, SCAN>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int w : {8, 12, 16}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((loop<MODE, SCAN>), dim3(256), dim3(64 * w), 100 * 1024, 0, d, steps);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((loop<MODE, SCAN>), dim3(256), dim3(64 * w), 100 * 1024, 0, d, steps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-58s %2d waves / CU  %8.3f ms  %9.1f state-steps/ns\n", name, w, ms, 256.0 * 64 * w * 16.0 * steps / (ms * 1e6));
    }
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 1024 * sizeof(float));
    const int steps = 20000;
    run<0, true>("chunk-scan step, packed FMA update", d, steps);
    run<1, true>("chunk-scan step, v_mfma_f32_4x4x1 rank-1 update", d, steps);
    run<0, false>("chunk-reduce step, packed FMA update", d, steps);
    run<1, false>("chunk-reduce step, v_mfma_f32_4x4x1 rank-1 update", d, steps);
    // (the layout claim itself - bit-equal outputs of the shipped kernel with the update on either unit - is checked by
    // building the library with -DWM_CORE_RANK1_MFMA=1 and running tests/test_gpu_parity.py -k core: profiles/r04/core_forward_experiments.txt)
    return 0;
}
