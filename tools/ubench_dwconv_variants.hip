// ubench_dwconv_variants.hip - source variants of dwconv3x3_kernel<0, true, bf16> (csrc/dwconv.hip.h), the one kernel of the
// inference forward whose results change while a 3x3 matrix-core convolution runs on another stream (tools/repro_victim_sweep.py):
// which ingredient of the compiler's code makes it vulnerable?  Driven by tools/repro_dwconv_variants.py.
//   VAR 0 the kernel as shipped in round 4 (bf16 in, bf16 out)      1 tap weights kept in VGPRs (no SGPR operands in the packed FMAs)
//       2 no wave shuffles (halo columns read as zero)              3 every row's loads waited for before any arithmetic (nothing in flight)
//       5 bf16 in, fp32 out (no bf16 rounding / packing code)       6 fp32 in, bf16 out
//       8 as 0 with the halo element taken out of a dword load (no global_load_ushort)
//       7 as 0 with an opaque barrier per output element (defeats the SLP vectoriser: no packed fp32 instructions)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../wave_mamba_amd/csrc/haar.hip.h"
namespace wm {
constexpr int kRows = 16;
template <int VAR, typename TI, typename TO>
__global__ __launch_bounds__(256) void dwv_kernel(const TI* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias,
                                                  TO* __restrict__ y, int C, int H, int W, long long planes) {
    const int lane = threadIdx.x;
    const int cg = blockIdx.x * 64 + lane;
    const int h0 = (blockIdx.y * 4 + threadIdx.y) * kRows;
    for (long long plane = blockIdx.z; plane < planes; plane += gridDim.z) {
        const int c = (int)(plane % C);
        float k[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) { k[i] = wgt[c * 9 + i]; if (VAR == 1) asm volatile("" : "+v"(k[i])); }
        float bv = bias ? bias[c] : 0.0f;
        if (VAR == 1) asm volatile("" : "+v"(bv));
        const TI* xp = x + plane * (long long)H * W;
        TO* yp = y + plane * (long long)H * W;
        const int w0 = cg * 4;
        const bool colok = w0 < W;
        auto fetch = [&](int r, float (&q)[4], float& e) {
            const int rc = min(max(r, 0), H - 1);
            const TI* rowp = xp + (long long)rc * W;
            load4(rowp + (colok ? w0 : 0), q);
            int we = lane == 0 ? w0 - 1 : w0 + 4;
            we = min(max(we, 0), W - 1);
            if (VAR == 8) {                                   // the halo element out of an aligned dword load (no sub-dword load)
                const uint32_t d = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(rowp) + ((size_t)we * sizeof(TI) & ~(size_t)3));
                e = __uint_as_float((we & 1) ? (d & 0xffff0000u) : (d << 16));
            } else e = ld1(rowp + we);
            if (VAR == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(e)); }
        };
        auto finish = [&](int r, const float (&q)[4], float e, float (&v)[6]) {
            const bool rowok = r >= 0 && r < H, ok = rowok && colok;
            const float q0 = ok ? q[0] : 0.f, q1 = ok ? q[1] : 0.f, q2 = ok ? q[2] : 0.f, q3 = ok ? q[3] : 0.f;
            float left = VAR == 2 ? 0.0f : __shfl_up(q3, 1), right = VAR == 2 ? 0.0f : __shfl_down(q0, 1);
            if (lane == 0) left = (rowok && w0 - 1 >= 0 && w0 - 1 < W) ? e : 0.0f;
            if (lane == 63) right = (rowok && w0 + 4 < W) ? e : 0.0f;
            v[0] = left; v[1] = q0; v[2] = q1; v[3] = q2; v[4] = q3; v[5] = right;
        };
        if (h0 < H) {
            float r0[6], r1[6], r2[6];
            const int hend = min(H, h0 + kRows);
            auto body = [&](int h) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float acc = bv;
                    acc = fmaf(k[0], r0[j], acc); acc = fmaf(k[1], r0[j + 1], acc); acc = fmaf(k[2], r0[j + 2], acc);
                    acc = fmaf(k[3], r1[j], acc); acc = fmaf(k[4], r1[j + 1], acc); acc = fmaf(k[5], r1[j + 2], acc);
                    acc = fmaf(k[6], r2[j], acc); acc = fmaf(k[7], r2[j + 1], acc); acc = fmaf(k[8], r2[j + 2], acc);
                    if (VAR == 7) asm volatile("" : "+v"(acc));
                    o[j] = acc;
                }
                if (colok) store4(yp + (long long)h * W + w0, o);
#pragma unroll
                for (int j = 0; j < 6; ++j) { r0[j] = r1[j]; r1[j] = r2[j]; }
            };
            float qa[4], qb[4], qc[4], qn[4], ea, eb, ec, en;
            fetch(h0 - 1, qa, ea); fetch(h0, qb, eb); fetch(h0 + 1, qc, ec);
            finish(h0 - 1, qa, ea, r0);
            finish(h0, qb, eb, r1);
            for (int h = h0; h < hend; ++h) {
                fetch(h + 2, qn, en);
                finish(h + 1, qc, ec, r2);
                body(h);
#pragma unroll
                for (int j = 0; j < 4; ++j) qc[j] = qn[j];
                ec = en;
            }
        }
    }
}
}  // namespace wm
extern "C" int dwv_launch(int var, const void* x, const float* w, const float* b, void* y, int C, int H, int W, long long planes, void* stream) {
    using namespace wm;
    const dim3 block(64, 4), grid((unsigned)((W + 255) / 256), (unsigned)((H + 4 * kRows - 1) / (4 * kRows)), (unsigned)(planes < 65535 ? planes : 65535));
    hipStream_t st = (hipStream_t)stream;
#define GO(V, TI, TO) hipLaunchKernelGGL((dwv_kernel<V, TI, TO>), grid, block, 0, st, (const TI*)x, w, b, (TO*)y, C, H, W, planes)
    switch (var) {
        case 0: GO(0, bf16_t, bf16_t); break;
        case 1: GO(1, bf16_t, bf16_t); break;
        case 2: GO(2, bf16_t, bf16_t); break;
        case 3: GO(3, bf16_t, bf16_t); break;
        case 5: GO(5, bf16_t, float); break;
        case 6: GO(6, float, bf16_t); break;
        case 7: GO(7, bf16_t, bf16_t); break;
        case 8: GO(8, bf16_t, bf16_t); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
