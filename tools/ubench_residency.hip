// ubench_residency.hip - how many workgroups of a given shape are resident at once on MI355X?
// Each workgroup spins ~200 us and records its start time (wall clock of the device) and XCC / CU ids.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned long long* out, int spin) {
    extern __shared__ float lds[];
    const unsigned long long t0 = wall_clock64();
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    lds[threadIdx.x] = (float)t0;
    __syncthreads();
    while (wall_clock64() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = t0; out[blockIdx.x * 4 + 1] = wall_clock64(); out[blockIdx.x * 4 + 2] = xcc & 15; out[blockIdx.x * 4 + 3] = hwid; }
}
static void run(int threads, int lds_bytes, int blocks) {
    unsigned long long* d; hipMalloc(&d, blocks * 4 * sizeof(unsigned long long));
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds_bytes, 0, d, 20000);      // 100 MHz wall clock: 200 us
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < blocks; ++b) t0 = std::min(t0, h[b * 4]);
    int first = 0;
    for (int b = 0; b < blocks; ++b) if (h[b * 4] - t0 < 10000) ++first;           // started within the first 100 us
    int per_xcc[16] = {0};
    for (int b = 0; b < blocks; ++b) if (h[b * 4] - t0 < 10000) per_xcc[h[b * 4 + 2]]++;
    printf("threads %4d  LDS %6d B  blocks %4d: %4d resident in the first round; per XCC:", threads, lds_bytes, blocks, first);
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
    hipFree(d);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("multiProcessorCount %d  maxSharedMemoryPerMultiProcessor %zu  sharedMemPerBlock %zu  sharedMemPerBlockOptin %zu\n",
           p.multiProcessorCount, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock, p.sharedMemPerBlockOptin);
    run(1024, 4096, 1024);
    run(1024, 65536, 1024);
    run(1024, 100000, 1024);
    run(1024, 135184, 1024);
    run(1024, 160 * 1024, 1024);
    run(512, 77712, 2048);
    run(512, 65536, 2048);
    run(256, 40000, 4096);
    run(64, 8192, 8192);
    return 0;
}
