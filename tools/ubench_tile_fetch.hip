// ubench_tile_fetch.hip - what does the operand fetch of the 3x3 convolution cost on its own?
// Every workgroup walks the 16-channel chunks of its halo tile of a (64, 1088, 1920) fp32 tensor (UHD level 1) the way
// conv2d_mfma_kernel does - load the chunk, wait, consume - with nothing else in the loop, for different tile shapes,
// load widths, chunks in flight and workgroups per compute unit.  Prints the time per pass over the tensor (the
// convolution spends 0.24 ms of its 0.59 ms exposed on this fetch; a streaming copy of the tensor reads it in 0.1 ms).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_tile_fetch tools/ubench_tile_fetch.hip && tools/ubench_tile_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int C = 64, H = 1088, W = 1920;

// TW x TH interior pixels, HALO columns / rows of halo, VEC floats per load (1 or 4; halo columns are whole quads when
// VEC == 4), DEPTH 16-channel chunks issued before the wait, MARCH tiles stacked vertically per workgroup (rows are
// fetched once: a sliding window)
template <int TW, int TH, int HALO, int VEC, int DEPTH, int MARCH>
__global__ __launch_bounds__(256) void fetch_kernel(const float* __restrict__ x, float* __restrict__ out) {
    extern __shared__ float lds[];
    constexpr int UPR = VEC == 1 ? TW + 2 * HALO : TW / 4 + 2 * HALO;     // load units per staged row
    const int tiles_x = W / TW, tiles_y = (H + TH * MARCH - 1) / (TH * MARCH);
    const int ntiles = tiles_x * tiles_y, nper = (ntiles + 7) / 8;
    const int tile = (blockIdx.x & 7) * nper + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    const int w0 = (tile % tiles_x) * TW, hbase = (tile / tiles_x) * TH * MARCH;
    const int tid = threadIdx.x;
    float acc = 0.0f;
    for (int mt = 0; mt < MARCH; ++mt) {
        // first tile of a strip: TH + 2 HALO rows; the following ones: TH new rows
        const int r0 = (mt == 0) ? hbase - HALO : hbase + mt * TH + HALO;
        const int nrows = (mt == 0) ? TH + 2 * HALO : TH;
        const int NP = nrows * UPR;
        constexpr int PIT = ((TH + 2 * HALO) * UPR + 255) / 256;
        unsigned off[PIT];
        bool ok[PIT];
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int p = tid + it * 256;
            const int pr = p / UPR, pu = p - pr * UPR;
            const int h = r0 + pr;
            const int w = VEC == 1 ? w0 - HALO + pu : w0 - 4 * HALO + 4 * pu;
            ok[it] = p < NP && h >= 0 && h < H && w >= 0 && w + VEC <= W;
            off[it] = ok[it] ? (unsigned)(h * W + w) : 0u;
        }
        for (int cc = 0; cc < C / 16; cc += DEPTH) {
            float v[DEPTH][16][PIT][VEC];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float* pl = x + (long long)((cc + d) * 16 + j) * H * W;
#pragma unroll
                    for (int it = 0; it < PIT; ++it) {
                        if (VEC == 1) v[d][j][it][0] = pl[off[it]];
                        else {
                            const float4 q = *reinterpret_cast<const float4*>(pl + off[it]);
                            v[d][j][it][0] = q.x; v[d][j][it][1] = q.y; v[d][j][it][2] = q.z; v[d][j][it][3] = q.w;
                        }
                    }
                }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
#pragma unroll
                for (int j = 0; j < 16; ++j)
#pragma unroll
                    for (int it = 0; it < PIT; ++it)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc += ok[it] ? v[d][j][it][e] : 0.0f;
            lds[tid] = acc;
            __syncthreads();
        }
    }
    out[blockIdx.x * 256ll + tid] = acc + lds[tid ^ 1];
}

template <int TW, int TH, int HALO, int VEC, int DEPTH, int MARCH>
static void run(const char* what, const float* x, float* out, int lds_bytes) {
    auto kern = fetch_kernel<TW, TH, HALO, VEC, DEPTH, MARCH>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const int tiles = (W / TW) * ((H + TH * MARCH - 1) / (TH * MARCH));
    const int grid = ((tiles + 7) / 8) * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, 0, x, out);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, 0, x, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double payload = 4.0 * C * H * W;
    printf("%-58s LDS %6d  %4d WGs: %.3f ms  (%.0f GB/s of payload)\n", what, lds_bytes, grid, ms, payload / ms / 1e6);
}

int main() {
    float *x, *out;
    hipMalloc(&x, sizeof(float) * C * H * W);
    hipMalloc(&out, sizeof(float) * 256 * 70000);
    hipMemset(x, 0, sizeof(float) * C * H * W);
    const int L2 = 80 * 1024, L4 = 40 * 1024, L8 = 20 * 1024;     // 2, 4, 8 workgroups per compute unit
    run<32, 8, 1, 1, 1, 1>("32x8 +halo, dword, 1 chunk in flight (the convolution)", x, out, L2);
    run<32, 8, 1, 1, 1, 1>("  same, 4 workgroups / CU", x, out, L4);
    run<32, 8, 1, 1, 1, 1>("  same, 8 workgroups / CU", x, out, L8);
    run<32, 8, 1, 1, 2, 1>("  2 chunks in flight", x, out, L2);
    run<32, 8, 1, 1, 4, 1>("  4 chunks in flight", x, out, L2);
    run<32, 8, 0, 1, 1, 1>("32x8 no halo, dword", x, out, L2);
    run<32, 8, 0, 1, 4, 1>("32x8 no halo, dword, 4 chunks in flight", x, out, L2);
    run<32, 8, 0, 4, 1, 1>("32x8 no halo, float4", x, out, L2);
    run<32, 8, 1, 4, 1, 1>("32x8 +halo (whole quads), float4", x, out, L2);
    run<32, 8, 1, 4, 4, 1>("32x8 +halo (whole quads), float4, 4 chunks in flight", x, out, L2);
    run<64, 4, 1, 1, 1, 1>("64x4 +halo, dword", x, out, L2);
    run<64, 4, 1, 4, 1, 1>("64x4 +halo, float4", x, out, L2);
    run<64, 4, 1, 4, 4, 1>("64x4 +halo, float4, 4 chunks in flight", x, out, L2);
    run<64, 8, 1, 4, 1, 1>("64x8 +halo, float4", x, out, L2);
    run<128, 4, 1, 4, 1, 1>("128x4 +halo, float4", x, out, L2);
    run<128, 4, 1, 4, 2, 1>("128x4 +halo, float4, 2 chunks in flight", x, out, L2);
    run<32, 8, 1, 1, 1, 8>("32x8 +halo, dword, marching 8 tiles down", x, out, L2);
    run<32, 8, 1, 4, 1, 8>("32x8 +halo, float4, marching 8 tiles down", x, out, L2);
    run<64, 4, 1, 4, 1, 16>("64x4 +halo, float4, marching 16 tiles down", x, out, L2);
    run<64, 4, 1, 4, 4, 16>("64x4 +halo, float4, marching 16, 4 chunks in flight", x, out, L2);
    return 0;
}
