// ubench_scan_formulations.hip - the two mappings of the selective-scan recurrence on gfx950, inner loops only
// (operands synthesised in registers / LDS, no HBM): cycles per state-step (t, d, n).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_scan_formulations.hip -o tools/ubench_scan_formulations
//
//   A  lane = channel (what csrc/ss2d_core.hip.h does): the N states of a channel in the lane's registers, time
//      sequential, B_t / C_t wave-uniform from LDS.  L-split => TWO passes, each with its own exponentials:
//        reduce: a = exp2(dt A); h = a h + (dt u) B                     scan: same + y += C h
//   B  lane = time block: a lane owns T consecutive steps of ONE channel, the wave 64 T steps; every exponential is
//      evaluated ONCE (kept in registers), the lanes' (P, H) aggregates are combined by a wave-level scan of the
//      monoid (a1,b1) o (a2,b2) = (a1 a2, a2 b1 + b2) on DPP row shifts / broadcasts, then every lane replays its T
//      steps from its carry-in: h = a h + b, y += C h.  B_t / C_t are per-lane values here (LDS, not broadcast).
// Reported: ns per state-step per SIMD at 4 waves per SIMD, and the ratio B / A.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f ex2(v2f x) { return (v2f){__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }

// ---- A: lane = channel ------------------------------------------------------------------------------------------
template <bool SCAN>
__global__ __launch_bounds__(256) void lane_channel(float* out, int steps) {
    __shared__ float s_rec[64 * 36];                      // 64 steps of [dt_r(4) | B(16) | C(16)]
    for (int i = threadIdx.x; i < 64 * 36; i += 256) s_rec[i] = 0.001f * (i % 37) - 0.01f;
    __syncthreads();
    v2f A2[8], h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { A2[i] = (v2f){-1.44f * (2 * i + 1), -1.44f * (2 * i + 2)}; h[i] = (v2f){0.f, 0.f}; }
    float acc = 0.f, u = 0.3f + 1e-3f * threadIdx.x;
    for (int t = 0; t < steps; ++t) {
        const float* rc = &s_rec[(t & 63) * 36];
        const float dt = 0.01f + 0.02f * rc[0] * u;       // stands in for the dt projection + softplus (same for both mappings)
        const v2f dt2 = {dt, dt}, du2 = {dt * u, dt * u};
        v2f y2 = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 bv = *reinterpret_cast<const float4*>(rc + 4 + 4 * r);
            const v2f a0 = ex2(dt2 * A2[2 * r]), a1 = ex2(dt2 * A2[2 * r + 1]);
            h[2 * r] = a0 * h[2 * r] + du2 * (v2f){bv.x, bv.y};
            h[2 * r + 1] = a1 * h[2 * r + 1] + du2 * (v2f){bv.z, bv.w};
            if (SCAN) {
                const float4 cv = *reinterpret_cast<const float4*>(rc + 20 + 4 * r);
                y2 = (v2f){cv.x, cv.y} * h[2 * r] + y2;
                y2 = (v2f){cv.z, cv.w} * h[2 * r + 1] + y2;
            }
        }
        if (SCAN) acc += y2.x + y2.y;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += h[i].x + h[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// ---- B: lane = time block ---------------------------------------------------------------------------------------
// wave-level inclusive scan of (P, H) over the 64 lanes, then shifted by one lane -> exclusive carry-in per lane
__device__ __forceinline__ float dpp_shr(float v, int ctrl_is) {   // ctrl: 1,2,4,8 = row_shr; 15 / 31 = row_bcast
    int iv = __float_as_int(v), r;
    switch (ctrl_is) {
        case 1: r = __builtin_amdgcn_update_dpp(0, iv, 0x111, 0xf, 0xf, true); break;     // row_shr:1, bound_ctrl (0 in)
        case 2: r = __builtin_amdgcn_update_dpp(0, iv, 0x112, 0xf, 0xf, true); break;
        case 4: r = __builtin_amdgcn_update_dpp(0, iv, 0x114, 0xf, 0xf, true); break;
        case 8: r = __builtin_amdgcn_update_dpp(0, iv, 0x118, 0xf, 0xf, true); break;
        case 15: r = __builtin_amdgcn_update_dpp(0, iv, 0x142, 0xa, 0xf, true); break;    // row_bcast:15 into rows 1, 3
        default: r = __builtin_amdgcn_update_dpp(0, iv, 0x143, 0xc, 0xf, true); break;    // row_bcast:31 into rows 2, 3
    }
    return __int_as_float(r);
}
template <int T>
__global__ __launch_bounds__(256) void lane_time(float* out, int blocks_of_T) {
    __shared__ float s_bc[64 * T * 32 / 8];               // B, C of the wave's steps (shared by the block's 4 waves here)
    for (int i = threadIdx.x; i < 64 * T * 4; i += 256) s_bc[i] = 0.001f * (i % 37) - 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float acc = 0.f, u = 0.3f + 1e-3f * threadIdx.x;
    v2f carry[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) carry[i] = (v2f){0.f, 0.f};
    for (int blk = 0; blk < blocks_of_T; ++blk) {          // one iteration = 64 T steps of one channel, all 16 states
        float dt[T], du[T], y[T];
#pragma unroll
        for (int t = 0; t < T; ++t) { dt[t] = 0.01f + 0.02f * s_bc[(lane * T + t) & 255] * u; du[t] = dt[t] * u; y[t] = 0.f; }
#pragma unroll
        for (int r = 0; r < 8; ++r) {                       // state pair (2r, 2r+1)
            const v2f A2 = {-1.44f * (2 * r + 1), -1.44f * (2 * r + 2)};
            v2f a[T], b[T], P = {1.f, 1.f}, H = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < T; ++t) {                   // local pass: every exponential once
                const v2f bt = *reinterpret_cast<const v2f*>(&s_bc[((lane * T + t) * 4 + (r & 3)) * 2 & 1022]);
                a[t] = ex2((v2f){dt[t], dt[t]} * A2);
                b[t] = (v2f){du[t], du[t]} * bt;
                H = a[t] * H + b[t];
                P = P * a[t];
            }
            // wave scan of (P, H): Kogge-Stone inside the 16-lane rows, then row broadcasts
#pragma unroll
            for (int s = 1; s <= 8; s <<= 1) {
                const v2f Pl = {dpp_shr(P.x, s), dpp_shr(P.y, s)}, Hl = {dpp_shr(H.x, s), dpp_shr(H.y, s)};
                const bool has = (lane & 15) >= s;
                H = has ? P * Hl + H : H;
                P = has ? P * Pl : P;
            }
#pragma unroll
            for (int s = 15; s <= 31; s += 16) {
                const v2f Pl = {dpp_shr(P.x, s), dpp_shr(P.y, s)}, Hl = {dpp_shr(H.x, s), dpp_shr(H.y, s)};
                const bool has = s == 15 ? (lane & 16) != 0 : lane >= 32;
                H = has ? P * Hl + H : H;
                P = has ? P * Pl : P;
            }
            // exclusive: the previous lane's inclusive state (lane 0: the carry of the previous 64 T steps)
            v2f hin = {__shfl_up(H.x, 1), __shfl_up(H.y, 1)};
            if (lane == 0) hin = carry[r];
            carry[r] = (v2f){__shfl(H.x, 63), __shfl(H.y, 63)};
            v2f h = hin;
#pragma unroll
            for (int t = 0; t < T; ++t) {                   // replay with the carry-in, emit y
                const v2f ct = *reinterpret_cast<const v2f*>(&s_bc[((lane * T + t) * 4 + (r & 3)) * 2 + 512 & 1022]);
                h = a[t] * h + b[t];
                const v2f yy = ct * h;
                y[t] += yy.x + yy.y;
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) acc += y[t];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename F> static float timeit(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; hipMalloc(&d, 1024 * 256 * sizeof(float));
    const int blocks = 1024;                              // 4 waves per block, 4 blocks per compute unit = 4 waves per SIMD
    const int steps = 8192;
    const double st_a = (double)blocks * 256 * steps * 16;            // state-steps per launch
    const float r = timeit([&] { hipLaunchKernelGGL(lane_channel<false>, dim3(blocks), dim3(256), 0, 0, d, steps); });
    const float s = timeit([&] { hipLaunchKernelGGL(lane_channel<true>, dim3(blocks), dim3(256), 0, 0, d, steps); });
    printf("A lane = channel : reduce %.3f ms, scan %.3f ms -> %.4f + %.4f = %.4f ns per state-step per SIMD (two passes)\n", r, s,
           r * 1e6 / st_a * 1024, s * 1e6 / st_a * 1024, (r + s) * 1e6 / st_a * 1024);
    const float a_total = (r + s) * 1e6 / st_a * 1024;
    {
        const int nb = 64;                                // 64 x (64 x 8) steps
        const float t8 = timeit([&] { hipLaunchKernelGGL(lane_time<8>, dim3(blocks), dim3(256), 0, 0, d, nb); });
        const double st = (double)blocks * 256 * nb * 8 * 16;
        printf("B lane = time, T = 8 : %.3f ms -> %.4f ns per state-step per SIMD (single pass, one exp per state-step); B / A = %.2f\n",
               t8, t8 * 1e6 / st * 1024, t8 * 1e6 / st * 1024 / a_total);
    }
    {
        const int nb = 32;
        const float t16 = timeit([&] { hipLaunchKernelGGL(lane_time<16>, dim3(blocks), dim3(256), 0, 0, d, nb); });
        const double st = (double)blocks * 256 * nb * 16 * 16;
        printf("B lane = time, T = 16: %.3f ms -> %.4f ns per state-step per SIMD; B / A = %.2f\n", t16, t16 * 1e6 / st * 1024,
               t16 * 1e6 / st * 1024 / a_total);
    }
    return 0;
}
