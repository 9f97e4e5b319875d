// ubench_mfma_valu.hip - does the fp32-input MFMA (v_mfma_f32_16x16x4_f32) overlap with fp32 VALU work on gfx950?
// hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_valu.hip -o tools/ubench_mfma_valu
// One 512-thread workgroup per compute unit = two waves per SIMD.  Modes:
//   0  every wave: MFMA only               1  every wave: packed-FMA only
//   2  every wave: 1 MFMA + 8 packed FMA interleaved in one instruction stream
//   3  waves 0-3 MFMA only, waves 4-7 packed-FMA only (one of each per SIMD)
//   4/5/6 = 0/2/3 with the bf16 MFMA (v_mfma_f32_16x16x32_bf16)
//   VK = 1: the VALU work is v_exp_f32 (48 per 12 MFMA) instead of v_pk_fma_f32; VK = 2: plain v_fma_f32 (96)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int MODE, int VK>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wv = threadIdx.x >> 6;
    f4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    v2f v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (v2f){0.5f + threadIdx.x * 1e-3f, 0.25f + i};
    const v2f a = {0.999f, 0.998f}, b = {0.001f, 0.002f};
    float wa = threadIdx.x * 1e-4f, xb = threadIdx.x * 2e-4f;
    bf8 ha, hb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(wa + i); hb[i] = (__bf16)(xb - i); }
    constexpr bool BF = MODE >= 4;
    constexpr int M = BF ? (MODE == 4 ? 0 : MODE == 5 ? 2 : 3) : MODE;
    const bool do_mfma = M == 0 || M == 2 || (M == 3 && wv < 4);
    const bool do_valu = M == 1 || M == 2 || (M == 3 && wv >= 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            if (do_mfma) {
                if constexpr (BF) acc[u % 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[u % 3], 0, 0, 0);
                else acc[u % 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, xb, acc[u % 3], 0, 0, 0);
            }
            if (do_valu) {
                if constexpr (VK == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = v[i] * a + b;
                } else if constexpr (VK == 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i].x = __builtin_amdgcn_exp2f(v[i].x);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i].x = fmaf(v[i].x, a.x, b.x);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
#pragma unroll
    for (int t = 0; t < 3; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int VK = 0> static void run(float* d, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL((k<MODE, VK>), dim3(256), dim3(512), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, VK>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2 waves; per wave per iteration 12 MFMA and / or 96 packed FMA
    printf("%-58s %8.3f ms   %7.1f ns per iteration (12 MFMA and/or 96 pk_fma per wave)\n", name, ms, ms * 1e6 / iters);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * sizeof(float));
    run<0>(d, "f32 MFMA 16x16x4 only (2 waves/SIMD)");
    run<1>(d, "v_pk_fma_f32 only (2 waves/SIMD)");
    run<2>(d, "f32 MFMA + pk_fma interleaved in every wave");
    run<3>(d, "f32 MFMA waves beside pk_fma waves (1 + 1 per SIMD)");
    run<4>(d, "bf16 MFMA 16x16x32 only");
    run<5>(d, "bf16 MFMA + pk_fma interleaved in every wave");
    run<6>(d, "bf16 MFMA waves beside pk_fma waves");
    run<1, 1>(d, "v_exp_f32 only (48 per iteration per wave)");
    run<2, 1>(d, "f32 MFMA + v_exp_f32 interleaved in every wave");
    run<3, 1>(d, "f32 MFMA waves beside v_exp_f32 waves");
    run<5, 1>(d, "bf16 MFMA + v_exp_f32 interleaved in every wave");
    run<1, 2>(d, "v_fma_f32 only (96 per iteration per wave)");
    run<2, 2>(d, "f32 MFMA + v_fma_f32 interleaved in every wave");
    run<3, 2>(d, "f32 MFMA waves beside v_fma_f32 waves");
    run<5, 2>(d, "bf16 MFMA + v_fma_f32 interleaved in every wave");
    return 0;
}
