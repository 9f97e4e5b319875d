// wavemamba_hip.hip - C ABI (include/wavemamba_hip.h) over the gfx950 kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC wavemamba_hip.hip -o libwavemamba_hip.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <array>
#include <atomic>
#include <list>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/wavemamba_hip.h"
#include "haar.hip.h"
#include "selscan.hip.h"
#include "selscan_bwd.hip.h"
#include "dwconv.hip.h"
#include "ss2d_core.hip.h"
#include "lfss.hip.h"
#include "lfss_mfma.hip.h"
#include "gram.hip.h"
#include "conv2d.hip.h"
#include "conv2d_ws.hip.h"
#include "hfe.hip.h"
#include "linear_wgrad.hip.h"
#include "ss2d_core_bwd.hip.h"
#include "imageio.hip.h"
#include "gates.hip.h"
#include "conv_wgrad.hip.h"
#include "patchify.hip.h"
#include "loss.hip.h"

namespace wm {

// ------------------------------------------------------------------------------------------------
// profiling hooks: HIP events on the launch stream around each kernel class
// ------------------------------------------------------------------------------------------------
struct Prof {
    std::mutex mu;
    unsigned mask = 0;                                          // bit k: record kernel class k
    std::vector<hipEvent_t> pool;                               // recycled events
    std::vector<std::pair<hipEvent_t, hipEvent_t>> rec[WM_PROF_NKERNELS];
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; hipEventCreate(&e); return e;
    }
};
static Prof g_prof;

struct ProfScope {
    int id; hipStream_t s; hipEvent_t e0 = nullptr, e1 = nullptr; bool active;
    ProfScope(int id_, hipStream_t s_) : id(id_), s(s_), active((g_prof.mask >> id_) & 1u) {
        if (!active) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        e0 = g_prof.get(); e1 = g_prof.get();
        hipEventRecord(e0, s);
    }
    ~ProfScope() {
        if (!active) return;
        hipEventRecord(e1, s);
        std::lock_guard<std::mutex> lk(g_prof.mu);
        g_prof.rec[id].emplace_back(e0, e1);
    }
};

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? WM_OK : (int)e;
}

// ------------------------------------------------------------------------------------------------
// Zeroing of accumulate-into outputs: a KERNEL, never hipMemsetAsync.  Round 6: a hipMemsetAsync captured into a HIP graph
// (torch.cuda.graph around a training step, trainer.GraphedTrainStep) becomes a memset node whose fill pattern this runtime
// (ROCm 7.0.2 as bundled with torch 2.10) re-reads at every launch of the graph from memory it has meanwhile recycled: with any
// eager kernel launch between two replays the node filled the gradient buffers with 16-byte records of somebody else's kernel
// arguments (tools/repro_graph_memset_node.py: every fourth float of a depth-wise weight gradient = the low half of a temporary's
// address, -1.5e38) - NaN parameters two replays later.  A kernel node carries its arguments by value.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zero_fill_kernel(uint32_t* __restrict__ p, size_t nwords, int vec) {
    const size_t stride = (size_t)gridDim.x * 256;
    if (vec) {                                                   // 16-byte aligned, nwords % 4 == 0
        uint4* q = reinterpret_cast<uint4*>(p);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords / 4; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride) p[i] = 0u;
    }
}

// bytes % 4 == 0 and a 4-byte aligned pointer (every caller zeroes float buffers)
static hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if ((bytes & 3u) || (reinterpret_cast<uintptr_t>(p) & 3u)) return hipErrorInvalidValue;
    const size_t nwords = bytes / 4;
    const int vec = aligned16(p) && (nwords % 4 == 0) ? 1 : 0;
    const size_t items = vec ? nwords / 4 : nwords;
    size_t blocks = (items + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), nwords, vec);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Haar launchers
// ------------------------------------------------------------------------------------------------
template <typename Tf, typename Ts, bool ROUND>
static int launch_analysis(const void* full, void* s0, void* s1, void* s2, void* s3, const HaarGeom& g,
                           bool vec, hipStream_t st) {
    const dim3 block(64, 4);
    const int cols = vec ? (g.w + 3) / 4 : g.w;
    const dim3 grid((unsigned)((g.rows + 3) / 4), (unsigned)((cols + 63) / 64));
    if (vec)
        hipLaunchKernelGGL((haar_analysis_kernel<Tf, Ts, ROUND, true>), grid, block, 0, st,
                           (const Tf*)full, (Ts*)s0, (Ts*)s1, (Ts*)s2, (Ts*)s3, g);
    else
        hipLaunchKernelGGL((haar_analysis_kernel<Tf, Ts, ROUND, false>), grid, block, 0, st,
                           (const Tf*)full, (Ts*)s0, (Ts*)s1, (Ts*)s2, (Ts*)s3, g);
    return launch_status();
}
template <typename Ts, typename Tf, bool ROUND>
static int launch_synthesis(const void* s0, const void* s1, const void* s2, const void* s3, void* full,
                            const HaarGeom& g, bool vec, hipStream_t st) {
    const dim3 block(64, 4);
    const int cols = vec ? (g.w + 3) / 4 : g.w;
    const dim3 grid((unsigned)((g.rows + 3) / 4), (unsigned)((cols + 63) / 64));
    if (vec)
        hipLaunchKernelGGL((haar_synthesis_kernel<Ts, Tf, ROUND, true>), grid, block, 0, st,
                           (const Ts*)s0, (const Ts*)s1, (const Ts*)s2, (const Ts*)s3, (Tf*)full, g);
    else
        hipLaunchKernelGGL((haar_synthesis_kernel<Ts, Tf, ROUND, false>), grid, block, 0, st,
                           (const Ts*)s0, (const Ts*)s1, (const Ts*)s2, (const Ts*)s3, (Tf*)full, g);
    return launch_status();
}

static int haar_geom(HaarGeom& g, int B, int C, int h, int w, int64_t b0, int64_t b1, int64_t b2,
                     int64_t b3) {
    if (B < 0 || C < 0 || h < 0 || w < 0) return WM_EINVAL;
    g.C = C; g.h = h; g.w = w; g.rows = (long long)B * C * h;
    g.bs[0] = b0; g.bs[1] = b1; g.bs[2] = b2; g.bs[3] = b3;
    if ((g.rows + 3) / 4 > 0x7fffffffLL) return WM_EINVAL;
    return WM_OK;
}
// vector path: 4 sub-band columns per thread, every row start / stride 16-byte aligned on the
// full-res side and 16 B (fp32) / 8 B (bf16) aligned on the sub-band side
static bool haar_vec_ok(const HaarGeom& g, const void* full, const void* const s[4]) {
    if (g.w % 4 != 0) return false;
    if (!aligned16(full)) return false;
    for (int k = 0; k < 4; ++k)
        if (!aligned16(s[k]) || (g.bs[k] % 4) != 0) return false;
    return true;
}

// ------------------------------------------------------------------------------------------------
// selective-scan host side
// ------------------------------------------------------------------------------------------------
struct ScanPlan { int NP, wpg, rows, chunk_len, nchunks; size_t ws_bytes; };

static inline long long carry_nsegs(long long nchunks) { return (nchunks + kCarrySegLen - 1) / kCarrySegLen; }

// phase 2 over [nchunks][nchains] summaries of `ndirs` independent scans with the same chain count and the same
// hierarchy depth (all <= 1024 chunks, or all above); d[i].segP / segH = scratch for carry_nsegs(nchunks) * nchains
// floats each
static void launch_carry_batch(CarryBatch cb, int ndirs, long long nchains, hipStream_t st) {
    ProfScope ps(3, st);
    const dim3 cgrid((unsigned)((nchains + 15) / 16), 1, (unsigned)ndirs), cblock(1024);
    int maxchunks = 0;
    for (int i = 0; i < ndirs; ++i) maxchunks = cb.d[i].nchunks > maxchunks ? cb.d[i].nchunks : maxchunks;
    if (maxchunks <= 1024 || !cb.d[0].segP) {
        hipLaunchKernelGGL((selscan_carry_kernel<false>), cgrid, cblock, 0, st, cb, nchains);
        return;
    }
    int maxsegs = 0;
    for (int i = 0; i < ndirs; ++i) {
        cb.d[i].nsegs = (int)carry_nsegs(cb.d[i].nchunks);
        maxsegs = cb.d[i].nsegs > maxsegs ? cb.d[i].nsegs : maxsegs;
    }
    const dim3 grid((unsigned)((nchains + 63) / 64), (unsigned)((maxsegs + 3) / 4), (unsigned)ndirs), block(256);
    hipLaunchKernelGGL((selscan_carry_seg_kernel<false>), grid, block, 0, st, cb, nchains);
    hipLaunchKernelGGL((selscan_carry_kernel<true>), cgrid, cblock, 0, st, cb, nchains);
    hipLaunchKernelGGL((selscan_carry_seg_kernel<true>), grid, block, 0, st, cb, nchains);
}
static void launch_carry(float* wsP, float* wsH, float* seg, long long nchains, int nchunks, hipStream_t st) {
    CarryBatch cb{};
    const int nsegs = (int)carry_nsegs(nchunks);
    cb.d[0] = CarryDir{wsP, wsH, seg, seg ? seg + (size_t)nsegs * nchains : nullptr, nchunks, nsegs};
    launch_carry_batch(cb, 1, nchains, st);
}

static int scan_plan(ScanPlan& pl, int batch, int dim, int L, int N, int G) {
    if (batch <= 0 || dim <= 0 || L <= 0 || N <= 0 || G <= 0) return WM_EINVAL;
    if (N > 32) return WM_EUNSUPPORTED;
    if (dim % G != 0) return WM_EINVAL;
    const int dpg = dim / G;
    pl.NP = N <= 16 ? 16 : 32;
    pl.wpg = (dpg + 63) / 64;
    const long long rows = (long long)batch * G * pl.wpg;
    if (rows > 65535) return WM_EUNSUPPORTED;
    pl.rows = (int)rows;
    // enough chunks to keep every CU's wave slots busy for several rounds (256 CUs x ~12 resident
    // single-wave workgroups), but never shorter than 64 steps
#ifndef WM_TARGET_WAVES
#define WM_TARGET_WAVES (256LL * 12 * 4)
#endif
    const long long target_waves = WM_TARGET_WAVES;
    long long want = (target_waves + rows - 1) / rows;           // chunks wanted per wave-row
    long long cl = ((long long)L + want - 1) / want;
    cl = ((cl + kTile - 1) / kTile) * kTile;
    if (cl < 64) cl = 64;
    pl.chunk_len = (int)cl;
    pl.nchunks = (int)(((long long)L + cl - 1) / cl);
    pl.ws_bytes = pl.nchunks > 1 ? (size_t)2 * (pl.nchunks + carry_nsegs(pl.nchunks)) * batch * dim * pl.NP * sizeof(float) : 0;
    return WM_OK;
}

template <int NP, bool VEC>
static int scan_launch(const ScanArgs& a, const ScanPlan& pl, hipStream_t st) {
    const dim3 grid((unsigned)pl.nchunks, (unsigned)pl.rows), block(64);
    if (pl.nchunks > 1) {
        {
            ProfScope ps(2, st);
            hipLaunchKernelGGL((selscan_chunk_kernel<NP, 1, VEC>), grid, block, 0, st, a);
        }
        const long long nchains = (long long)a.batch * a.dim * NP;
        launch_carry(a.wsP, a.wsH, a.wsH + (size_t)pl.nchunks * nchains, nchains, pl.nchunks, st);
    }
    {
        ProfScope ps(4, st);
        hipLaunchKernelGGL((selscan_chunk_kernel<NP, 3, VEC>), grid, block, 0, st, a);
    }
    return launch_status();
}

// > 64 KB of dynamic LDS is an opt-in per kernel function AND per device: one flag per (instantiation, device).
// `flags` is the caller's function-local static array; returns WM_OK or WM_EHIP.
static int lds_optin(const void* fn, int bytes, bool (&flags)[64]) {
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return WM_EHIP;
    std::lock_guard<std::mutex> lk(mu);
    if (!flags[dev]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return WM_EHIP;
        flags[dev] = true;
    }
    return WM_OK;
}

}  // namespace wm

using namespace wm;

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int wm_abi_version(void) { return 31; }

#ifndef WM_BUILD_ID
#define WM_BUILD_ID "unknown"
#endif
const char* wm_build_id(void) { return WM_BUILD_ID; }

const char* wm_strerror(int code) {
    switch (code) {
        case WM_OK: return "ok";
        case WM_EINVAL: return "invalid shape or size argument";
        case WM_ENULL: return "required pointer is NULL";
        case WM_EALIGN: return "pointer not aligned to its element size";
        case WM_EWORKSPACE: return "workspace too small";
        case WM_EUNSUPPORTED: return "argument combination not supported";
        case WM_EHIP: return "a HIP runtime call made on behalf of the launch failed";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

int wm_dwt2d_fwd(const void* x, void* ll, void* hl, void* lh, void* hh, int B, int C, int H, int W,
                 int dtype, void* stream) {
    if (H % 2 != 0 || W % 2 != 0) return WM_EINVAL;        // reference: RuntimeError on odd sizes
    HaarGeom g;
    const int64_t bs = (int64_t)C * (H / 2) * (W / 2);
    int rc = haar_geom(g, B, C, H / 2, W / 2, bs, bs, bs, bs);
    if (rc) return rc;
    if (g.rows == 0 || g.w == 0) return WM_OK;
    if (!x || !ll || !hl || !lh || !hh) return WM_ENULL;
    const void* s[4] = {ll, hl, lh, hh};
    const bool vec = haar_vec_ok(g, x, s);
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(0, st);
    if (dtype == WM_F32) return launch_analysis<float, float, false>(x, ll, hl, lh, hh, g, vec, st);
    if (dtype == WM_BF16) return launch_analysis<bf16_t, bf16_t, true>(x, ll, hl, lh, hh, g, vec, st);
    return WM_EUNSUPPORTED;
}

int wm_dwt2d_bwd(const void* dll, const void* dhl, const void* dlh, const void* dhh, void* dx, int B,
                 int C, int H, int W, int dtype, void* stream) {
    if (H % 2 != 0 || W % 2 != 0) return WM_EINVAL;
    HaarGeom g;
    const int64_t bs = (int64_t)C * (H / 2) * (W / 2);
    int rc = haar_geom(g, B, C, H / 2, W / 2, bs, bs, bs, bs);
    if (rc) return rc;
    if (g.rows == 0 || g.w == 0) return WM_OK;
    if (!dx || !dll || !dhl || !dlh || !dhh) return WM_ENULL;
    const void* s[4] = {dll, dhl, dlh, dhh};
    const bool vec = haar_vec_ok(g, dx, s);
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(1, st);
    // the 2x2 Haar matrix (with the 1/2 scaling) is orthogonal: d(analysis) = synthesis
    if (dtype == WM_F32) return launch_synthesis<float, float, false>(dll, dhl, dlh, dhh, dx, g, vec, st);
    if (dtype == WM_BF16) return launch_synthesis<bf16_t, bf16_t, false>(dll, dhl, dlh, dhh, dx, g, vec, st);
    return WM_EUNSUPPORTED;
}

int wm_idwt2d_fwd(const void* x1, const void* x2, const void* x3, const void* x4, int64_t bs1,
                  int64_t bs2, int64_t bs3, int64_t bs4, float* out, int B, int C, int h, int w,
                  int dtype, void* stream) {
    HaarGeom g;
    int rc = haar_geom(g, B, C, h, w, bs1, bs2, bs3, bs4);
    if (rc) return rc;
    if (g.rows == 0 || g.w == 0) return WM_OK;
    if (!x1 || !x2 || !x3 || !x4 || !out) return WM_ENULL;
    const void* s[4] = {x1, x2, x3, x4};
    const bool vec = haar_vec_ok(g, out, s);
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(1, st);
    if (dtype == WM_F32) return launch_synthesis<float, float, false>(x1, x2, x3, x4, out, g, vec, st);
    if (dtype == WM_BF16) return launch_synthesis<bf16_t, float, true>(x1, x2, x3, x4, out, g, vec, st);
    return WM_EUNSUPPORTED;
}

int wm_idwt2d_bwd(const float* dout, void* d1, void* d2, void* d3, void* d4, int64_t bs1, int64_t bs2,
                  int64_t bs3, int64_t bs4, int B, int C, int h, int w, int dtype, void* stream) {
    HaarGeom g;
    int rc = haar_geom(g, B, C, h, w, bs1, bs2, bs3, bs4);
    if (rc) return rc;
    if (g.rows == 0 || g.w == 0) return WM_OK;
    if (!dout || !d1 || !d2 || !d3 || !d4) return WM_ENULL;
    const void* s[4] = {d1, d2, d3, d4};
    const bool vec = haar_vec_ok(g, dout, s);
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(0, st);
    if (dtype == WM_F32) return launch_analysis<float, float, false>(dout, d1, d2, d3, d4, g, vec, st);
    if (dtype == WM_BF16) return launch_analysis<float, bf16_t, false>(dout, d1, d2, d3, d4, g, vec, st);
    return WM_EUNSUPPORTED;
}

size_t wm_selscan_fwd_workspace_bytes(int batch, int dim, int L, int N, int G) {
    ScanPlan pl;
    if (scan_plan(pl, batch, dim, L, N, G) != WM_OK) return 0;
    return pl.ws_bytes;
}

int wm_selscan_fwd(const float* u, const float* delta, const float* A, const float* Bm, const float* Cm,
                   const float* D, const float* z, const float* delta_bias, float* out,
                   float* last_state, void* workspace, size_t workspace_bytes, int batch, int dim, int L,
                   int N, int G, int delta_softplus, void* stream) {
    if (batch == 0 || dim == 0 || L == 0) return (batch < 0 || dim < 0 || L < 0) ? WM_EINVAL : WM_OK;
    ScanPlan pl;
    int rc = scan_plan(pl, batch, dim, L, N, G);
    if (rc) return rc;
    if (!u || !delta || !A || !Bm || !Cm || !out) return WM_ENULL;
    if (pl.ws_bytes > 0) {
        if (!workspace) return WM_ENULL;
        if (workspace_bytes < pl.ws_bytes) return WM_EWORKSPACE;
        if (!aligned16(workspace)) return WM_EALIGN;
    }
    ScanArgs a;
    a.u = u; a.delta = delta; a.A = A; a.Bm = Bm; a.Cm = Cm; a.D = D; a.z = z; a.bias = delta_bias;
    a.out = out; a.last_state = last_state;
    a.wsP = (float*)workspace;
    a.wsH = a.wsP ? a.wsP + (size_t)pl.nchunks * batch * dim * pl.NP : nullptr;
    a.batch = batch; a.dim = dim; a.L = L; a.N = N; a.G = G; a.dpg = dim / G; a.wpg = pl.wpg;
    a.chunk_len = pl.chunk_len; a.nchunks = pl.nchunks; a.softplus = delta_softplus ? 1 : 0;
    const bool vec = (L % 4 == 0) && aligned16(u) && aligned16(delta) && aligned16(Bm) && aligned16(Cm) &&
                     aligned16(out) && (!z || aligned16(z));
    hipStream_t st = (hipStream_t)stream;
    if (pl.NP == 16) return vec ? scan_launch<16, true>(a, pl, st) : scan_launch<16, false>(a, pl, st);
    return vec ? scan_launch<32, true>(a, pl, st) : scan_launch<32, false>(a, pl, st);
}

}  // extern "C" (templates below need C++ linkage)
namespace wm {
struct BwdPlan {
    int NP, wpg, rows, nchunks, cpb, nblocks; long long chains;
    size_t blk_bytes, arr_bytes, s_bytes, seg_bytes, part_bytes, total;
    // workspace: [P | H | Pr | G] block summaries (blk_bytes each), per-chunk local states (arr_bytes), per-chunk dt sums
    // (s_bytes), carry scratch for two scans (seg_bytes), per-block parameter-gradient partials (part_bytes)
    size_t off_hl() const { return 4 * blk_bytes; }
    size_t off_s() const { return off_hl() + arr_bytes; }
    size_t off_seg() const { return off_s() + s_bytes; }
    size_t off_part() const { return off_seg() + seg_bytes; }
};
// chunks per block of the two backward kernels: enough single-wave blocks to fill the chip (the reduce kernel holds 9
// per compute unit = 2,304 at once).  The summaries the carry kernels walk are per BLOCK, so longer blocks also mean a
// shorter carry and fewer partial records.  BASELINE config 3 training step on one MI355X (tools/train_breakdown.py,
// gpurun_out r3z): at least 8192 blocks / at most 8 chunks each 101.2 ms, 4096 / 8 99.7, 2048 / 16 95.9, 1024 / 32 97.5.
#ifndef WM_BWD_CPB_MAX
#define WM_BWD_CPB_MAX 16
#endif
#ifndef WM_BWD_MIN_BLOCKS
#define WM_BWD_MIN_BLOCKS 2048
#endif
// blocks of the finish kernel per channel: a thread adds up at most ~4 records per batch item
static int bwd_finish_split(int nblocks, int npp) {
    const int stride = (256 / npp) * npp;
    const long long per = (long long)nblocks * npp;
    long long y = (per + 4LL * stride - 1) / (4LL * stride);
    return (int)(y < 1 ? 1 : (y > 64 ? 64 : y));
}
// The fused core's gradient kernel (core_bwd_chunk_kernel) is a workgroup of NP / 8 waves with 37,952 / 58,944 B of LDS and ~256
// registers per lane: FOUR (N <= 16) / TWO (N <= 32) workgroups are resident per compute unit, 1024 / 512 on the chip, and a
// workgroup pays a prologue (weights, carried states) of ~0.4 chunks before its first chunk.  Its block length is therefore chosen
// per shape: the chunks per block c in [1, 32] that minimise  ceil(workgroups(c) / resident) * (c + 0.4)  - whole dispatch rounds
// of resident workgroups - ties to the longer block (fewer summaries, shorter carry).  Round 5, BASELINE config 3 on one MI355X
// (profiles/r05/core_bwd_block_length_ab.txt): against "at least 2048 blocks, at most 16 chunks" (two rounds at every level)
// 3.93 -> 3.77 / 1.19 -> 1.04 / 0.514 -> 0.368 ms per call at levels 1 / 2 / 3, 59.0 -> 57.1 ms per training step; 512, 1536, 2048
// and 4096 blocks are all slower (1536: one and a half rounds, the worst).
static int bwd_fused_cpb(int nchunks, long long rows, int NP) {
    const long long resident = NP == 16 ? 1024 : 512;
    int best = 1;
    double best_cost = 1e300;
    for (int c = 1; c <= 32; ++c) {
        const long long wgs = (long long)((nchunks + c - 1) / c) * rows;
        const double cost = (double)((wgs + resident - 1) / resident) * (c + 0.4);
        if (cost <= best_cost) { best_cost = cost; best = c; }
    }
    return best;
}
static int bwd_plan(BwdPlan& pl, int batch, int dim, int L, int N, int G, int part_pad = kPartPad) {
    if (batch <= 0 || dim <= 0 || L <= 0 || N <= 0 || G <= 0) return WM_EINVAL;
    if (N > 32) return WM_EUNSUPPORTED;
    if (dim % G != 0) return WM_EINVAL;
    pl.NP = N <= 16 ? 16 : 32;
    pl.wpg = (dim / G + 63) / 64;
    const long long rows = (long long)batch * G * pl.wpg;
    if (rows > 65535) return WM_EUNSUPPORTED;
    pl.rows = (int)rows;
    pl.nchunks = (L + kBT - 1) / kBT;
    {
        const long long blocks1 = (long long)pl.nchunks * pl.rows;
        const int cpb = (int)(blocks1 / (long long)WM_BWD_MIN_BLOCKS);
        pl.cpb = cpb < 1 ? 1 : (cpb > WM_BWD_CPB_MAX ? WM_BWD_CPB_MAX : cpb);
        if (part_pad == kPartPadFused) pl.cpb = bwd_fused_cpb(pl.nchunks, rows, pl.NP);     // the fused core's gradient kernel
    }
    pl.nblocks = (pl.nchunks + pl.cpb - 1) / pl.cpb;
    pl.chains = (long long)batch * dim * pl.NP;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    pl.blk_bytes = up((size_t)pl.nblocks * pl.chains * sizeof(float));
    pl.arr_bytes = up((size_t)pl.nchunks * pl.chains * sizeof(float));
    pl.s_bytes = up((size_t)pl.nchunks * batch * dim * sizeof(float));
    pl.seg_bytes = up((size_t)4 * carry_nsegs(pl.nblocks) * pl.chains * sizeof(float));
    pl.part_bytes = up((size_t)pl.nblocks * batch * dim * (pl.NP + part_pad) * sizeof(float));
    pl.total = pl.off_part() + pl.part_bytes;
    return WM_OK;
}

// forward and adjoint carry of one direction in ONE batch launch (same length, same depth)
static void bwd_launch_carry(const ScanBwdArgs& a, const BwdPlan& pl, float* seg, hipStream_t st) {
    CarryBatch cb{};
    const int nsegs = (int)carry_nsegs(pl.nblocks);
    const size_t one = (size_t)nsegs * pl.chains;
    cb.d[0] = CarryDir{a.wsP, a.wsH, seg, seg + one, pl.nblocks, nsegs};
    cb.d[1] = CarryDir{a.wsPr, a.wsG, seg + 2 * one, seg + 3 * one, pl.nblocks, nsegs};
    launch_carry_batch(cb, 2, pl.chains, st);
}
static void bwd_bind_workspace(ScanBwdArgs& a, const BwdPlan& pl, char* w, float*& seg) {
    a.wsP = (float*)w; a.wsH = (float*)(w + pl.blk_bytes); a.wsPr = (float*)(w + 2 * pl.blk_bytes);
    a.wsG = (float*)(w + 3 * pl.blk_bytes);
    a.wsHl = (float*)(w + pl.off_hl()); a.wsS = (float*)(w + pl.off_s());
    seg = (float*)(w + pl.off_seg());
    a.part = (float*)(w + pl.off_part());
    a.nchunks = pl.nchunks; a.cpb = pl.cpb; a.nblocks = pl.nblocks;
}

template <int NP, bool VEC>
static int bwd_launch(const ScanBwdArgs& a0, const BwdPlan& pl, float* seg, float* dA, float* dD, float* dbias,
                      hipStream_t st) {
    ScanBwdArgs a = a0;
    const dim3 grid((unsigned)pl.nblocks, (unsigned)pl.rows), block(64);
    ProfScope ps(12, st);
    if (pl.nchunks > 1) {
        hipLaunchKernelGGL((selscan_bwd_reduce_kernel<NP, VEC, 0>), grid, block, 0, st, a);
        if (pl.nblocks > 1) bwd_launch_carry(a, pl, seg, st);
    }
    hipLaunchKernelGGL((selscan_bwd_chunk_kernel<NP, VEC, 0>), grid, block, 0, st, a);
    zero_async(dA, (size_t)a.dim * a.N * sizeof(float), st);
    if (dD) zero_async(dD, (size_t)a.dim * sizeof(float), st);
    if (dbias) zero_async(dbias, (size_t)a.dim * sizeof(float), st);
    const int ysplit = bwd_finish_split(pl.nblocks, NP + kPartPad);
    hipLaunchKernelGGL(selscan_bwd_finish_kernel, dim3((unsigned)a.dim, (unsigned)ysplit), dim3(256), 0, st,
                       (const float*)a.part, dA, dD, dbias, a.batch, a.dim, a.N, NP + kPartPad, pl.nblocks,
                       (const float*)nullptr, (float*)nullptr, 0, NP);
    return launch_status();
}

}  // namespace wm
extern "C" {
size_t wm_selscan_bwd_workspace_bytes(int batch, int dim, int L, int N, int G) {
    BwdPlan pl;
    if (bwd_plan(pl, batch, dim, L, N, G) != WM_OK) return 0;
    return pl.total;
}

int wm_selscan_bwd(const float* u, const float* delta, const float* A, const float* Bm, const float* Cm,
                   const float* D, const float* delta_bias, const float* dy, float* du, float* ddelta,
                   float* dA, float* dB, float* dC, float* dD, float* dbias, void* workspace,
                   size_t workspace_bytes, int batch, int dim, int L, int N, int G, int delta_softplus,
                   void* stream) {
    if (batch == 0 || dim == 0 || L == 0) return (batch < 0 || dim < 0 || L < 0) ? WM_EINVAL : WM_OK;
    BwdPlan pl;
    int rc = bwd_plan(pl, batch, dim, L, N, G);
    if (rc) return rc;
    if (!u || !delta || !A || !Bm || !Cm || !dy || !du || !ddelta || !dA || !dB || !dC) return WM_ENULL;
    if (!workspace) return WM_ENULL;
    if (workspace_bytes < pl.total) return WM_EWORKSPACE;
    if (!aligned16(workspace)) return WM_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    ScanBwdArgs a;
    a.u = u; a.delta = delta; a.A = A; a.Bm = Bm; a.Cm = Cm; a.D = D; a.bias = delta_bias; a.dy = dy;
    a.du = du; a.ddelta = ddelta; a.dB = dB; a.dC = dC;
    float* seg = nullptr;
    bwd_bind_workspace(a, pl, (char*)workspace, seg);
    a.batch = batch; a.dim = dim; a.L = L; a.N = N; a.G = G; a.dpg = dim / G; a.wpg = pl.wpg;
    a.softplus = delta_softplus ? 1 : 0; a.atomic_bc = pl.wpg > 1 ? 1 : 0; a.accumulate = 0;
    if (a.atomic_bc) {
        const size_t nb = (size_t)batch * G * N * L * sizeof(float);
        hipError_t e = zero_async(dB, nb, st);
        if (e == hipSuccess) e = zero_async(dC, nb, st);
        if (e != hipSuccess) return (int)e;
    }
    const bool vec = (L % 4 == 0) && aligned16(u) && aligned16(delta) && aligned16(Bm) && aligned16(Cm) &&
                     aligned16(dy) && aligned16(du) && aligned16(ddelta) && aligned16(dB) && aligned16(dC);
    if (pl.NP == 16) return vec ? bwd_launch<16, true>(a, pl, seg, dA, dD, dbias, st)
                                : bwd_launch<16, false>(a, pl, seg, dA, dD, dbias, st);
    return vec ? bwd_launch<32, true>(a, pl, seg, dA, dD, dbias, st) : bwd_launch<32, false>(a, pl, seg, dA, dD, dbias, st);
}

// lanes per strip row of the depth-wise kernels (dwconv.hip.h): narrow maps put 2 / 4 planes side by side in a wave
static int dw_lanes_per_row(int W, bool vec) { return !vec || W > 128 ? 64 : (W > 64 ? 32 : 16); }
// rows per strip: 16 (every input row fetched 18 / 16 times), or 8 / 4 on small problems - a strip is a chain of dependent row
// fetches, and 16-row strips of a 64 x 64 map are 128-384 workgroups for 256 compute units (21 us per weight-gradient launch at
// BASELINE config 3's level 3, whatever the lanes did)
static int dw_strip_rows(int H, long long column_blocks, long long plane_groups) {
    const long long waves16 = column_blocks * ((H + kDwRows - 1) / kDwRows) * plane_groups;
    return waves16 >= 2048 ? kDwRows : (waves16 >= 1024 ? 8 : 4);
}

int wm_dwconv3x3_fwd(const void* x, const float* weight, const float* bias, void* y, int B, int C,
                     int H, int W, int act, int plane_dtype, void* stream) {
    if (B < 0 || C < 0 || H < 0 || W < 0) return WM_EINVAL;
    const int flip = (act >> 2) & 1;                  // act + 4: the taps rotated by 180 degrees (the input gradient's convolution)
    act &= 3;
    if (act < 0 || act > 2 || (plane_dtype != WM_F32 && plane_dtype != WM_BF16)) return WM_EUNSUPPORTED;
    const long long planes = (long long)B * C;
    if (planes == 0 || H == 0 || W == 0) return WM_OK;
    if (!x || !weight || !y) return WM_ENULL;
    const bool vec = (W % 4 == 0) && aligned16(x) && aligned16(y);      // (bf16: 8-byte accesses, covered by the same test)
    const int lpr = dw_lanes_per_row(W, vec);
    const long long pgroups = (planes + 64 / lpr - 1) / (64 / lpr);
    const dim3 block(64, 4);
    const int rows = dw_strip_rows(H, (W + 4 * lpr - 1) / (4 * lpr), pgroups);
    const dim3 grid((unsigned)((W + 4 * lpr - 1) / (4 * lpr)), (unsigned)((H + 4 * rows - 1) / (4 * rows)),
                    (unsigned)(pgroups < 65535 ? pgroups : 65535));
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(act == 1 ? 7 : 17, st);       // + SiLU: SS2D's conv2d (:486-487, hot path); the others belong to the HFE branch / the ffn
#define WM_DW1(ACT, VEC, LPR)                                                                                                 \
    do {                                                                                                                      \
        if (plane_dtype == WM_F32)                                                                                            \
            hipLaunchKernelGGL((dwconv3x3_kernel<ACT, VEC, float, LPR>), grid, block, 0, st, (const float*)x, weight, bias,   \
                               (float*)y, C, H, W, planes, rows, flip);                                                       \
        else                                                                                                                  \
            hipLaunchKernelGGL((dwconv3x3_kernel<ACT, VEC, bf16_t, LPR>), grid, block, 0, st, (const bf16_t*)x, weight, bias, \
                               (bf16_t*)y, C, H, W, planes, rows, flip);                                                      \
    } while (0)
#define WM_DW(ACT)                                                                                                            \
    do {                                                                                                                      \
        if (!vec) WM_DW1(ACT, false, 64); else if (lpr == 64) WM_DW1(ACT, true, 64);                                          \
        else if (lpr == 32) WM_DW1(ACT, true, 32); else WM_DW1(ACT, true, 16);                                                \
    } while (0)
    if (act == 1) WM_DW(1); else if (act == 2) WM_DW(2); else WM_DW(0);
#undef WM_DW
#undef WM_DW1
    return launch_status();
}

}  // extern "C"
namespace wm {
// ------------------------------------------------------------------------------------------------
// SS2D core, second generation (ss2d_core.hip.h): reduce (4 directions) -> carry -> scan (4 directions)
// ------------------------------------------------------------------------------------------------
struct CorePlan {
    int NP, NW;
    int row_chunk, row_nchunks, row_wgs, row_cpw;
    int col_seg, col_nseg, col_tiles, col_wgs;
    long long col_nchunks, max_chunks;
    size_t half_bytes, ytmp_bytes, prep_bytes, total;   // one P (or H) array of one direction; merged-mode y buffers;
                                                        // the prep kernel's output; everything
};

// Length of one core launch (in row-tile times) under the dispatch model of core_plan's comment, replaying
// ss2d_core_kernel's blockIdx -> (direction, slot) mapping.  Costs measured on MI355X (tools/bench_core.py with
// WM_CORE_DIRMASK, gpurun_out r3a): a 16-wave tile round takes the same 15.7 us (chunk-scan) / 13.4 us (chunk-reduce) in
// a column workgroup (68 tiles, one round of 240: 1.065 / 0.919 ms) as in a row workgroup (60 tiles: 0.942 / 0.790 ms),
// so kColTile = 1; a workgroup costs about one tile round on top of its tiles (UHD level 2: 240 workgroups of 34 tiles
// in one round 1.07 ms, 512 of 17 / 15 tiles in two rounds 1.12 ms; level 3: 234 of 9 tiles 0.319 ms, 222 with 10-tile
// row workgroups 0.340 ms) - launch, LDS fill, carried-in state, and above all the drain: a compute unit holds one
// workgroup, and the next one starts only when the slowest of its 16 waves has finished.
// Returns early (a value >= bound) once the launch is known to be no better than `bound`.
#ifndef WM_CORE_COST_COL
#define WM_CORE_COST_COL 1.0
#endif
#ifndef WM_CORE_COST_WG
#define WM_CORE_COST_WG 1.0
#endif
static double core_makespan(const CorePlan& pl, int B, int H, double bound) {
    const double kColTile = WM_CORE_COST_COL, kWgStart = WM_CORE_COST_WG, kCarry = 0.0005;
    const int slots = 32 * (16 / pl.NW);
    double freeat[8][64];
    for (int x = 0; x < 8; ++x) for (int i = 0; i < slots; ++i) freeat[x][i] = 0.0;
    const long long per_b = 2LL * pl.row_wgs + 2LL * pl.col_wgs;
    const long long total = (long long)B * per_b;
    // a row workgroup's waves share its row_cpw chunks: tile rounds = chunks x tiles per chunk / waves
    const int row_tiles = (pl.row_cpw * (pl.row_chunk / 16) + pl.NW - 1) / pl.NW;
    double span = 0.0;
    {   // lower bound: all work spread evenly
        double work = 0.0;
        for (int sg = 0; sg < pl.col_nseg; ++sg) {
            const int rows = (H - sg * pl.col_seg) < pl.col_seg ? (H - sg * pl.col_seg) : pl.col_seg;
            work += 2.0 * B * pl.col_tiles * (((rows + 15) / 16) * kColTile + kWgStart);
        }
        work += 2.0 * B * pl.row_wgs * (row_tiles + kWgStart);
        if (work / (8.0 * slots) >= bound) return work / (8.0 * slots);
    }
    for (long long i = 0; i < total; ++i) {
        const long long r = i % per_b;
        double cost;
        if (r < 2LL * pl.col_wgs) {
            const int idx = (int)(r >> 1);
            const int wg = (((idx >> 3) << 2) + (idx & 3)) * 2 + ((idx >> 2) & 1);
            if (wg >= pl.col_tiles * pl.col_nseg) continue;
            const int sg = wg / pl.col_tiles;
            const int rows = (H - sg * pl.col_seg) < pl.col_seg ? (H - sg * pl.col_seg) : pl.col_seg;
            cost = ((rows + 15) / 16) * kColTile + kWgStart;
        } else {
            const long long wg = (r - 2LL * pl.col_wgs) >> 1;
            long long c0 = wg * pl.row_cpw;                                   // first chunk of the workgroup
            if (c0 >= pl.row_nchunks) continue;
            cost = row_tiles + kWgStart;                                       // (the very last chunk may be shorter)
        }
        double* f = freeat[i & 7];
        int best = 0;
        for (int j = 1; j < slots; ++j) if (f[j] < f[best]) best = j;
        f[best] += cost;
        if (f[best] > span) { span = f[best]; if (span >= bound) return span; }
    }
    const long long chunks = pl.col_nchunks > pl.row_nchunks ? pl.col_nchunks : pl.row_nchunks;
    return span + kCarry * (double)chunks;
}

static int core_plan(CorePlan& pl, int B, int D, int H, int W, int N, int R, int merged) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || N <= 0 || R <= 0) return WM_EINVAL;
    if (N > 32 || R > 4 || D > 64) return WM_EUNSUPPORTED;
    const long long L = (long long)H * W;
    if (L * D > 0x7fffffffLL) return WM_EUNSUPPORTED;          // 32-bit element offsets inside one batch item
    pl.NP = N <= 16 ? 16 : 32;
    pl.NW = pl.NP == 16 ? 16 : 8;
#ifdef WM_CORE_NW
    if (pl.NP == 16) pl.NW = WM_CORE_NW;
#endif
    const int NW = pl.NW;
    pl.col_tiles = (W + NW - 1) / NW;
    // Work split.  Column workgroups: one per (NW-column tile, segment of `col_seg` rows); row workgroups: NW waves with
    // one chunk of `row_chunk` steps each.  Workgroup i goes to XCD i % 8 and there to the first of 32 compute units
    // (x 2 workgroups at NW = 8) that frees up - so a launch is list scheduling in workgroup-id order, and its length
    // is decided by how the LAST workgroups fill the machine.  Round 2 gave the row workgroups the column workgroups'
    // length: 480 equal workgroups on 256 compute units at UHD level 1 = 2 rounds, the second 7/8 full, and halving or
    // quartering the row chunks changes nothing (480 k / 256 is never whole).  Now the plan is searched: every
    // (segments per column, tiles per row chunk) pair is replayed through that dispatch model with measured per-tile
    // costs (core_makespan) and the shortest launch wins, e.g. level 1: 240 column workgroups of 68 tiles + 272 row
    // workgroups of 60.  Plans are cached per shape (the search replays ~10^5 workgroups).
    const long long RT = ((L + 15) / 16 + NW - 1) / NW;             // row workgroup-tiles per direction
    // Row workgroups hand their chunks to their waves on demand (ss2d_core.hip.h), kRowSplit chunks per wave.  Measured
    // (gpurun_out r3h, UHD levels 1 / 2 / 3, ms per call): 1 chunk per wave 3.564 / 1.034 / 0.323, 2: 3.541 / 1.046 /
    // 0.339, 3: 3.498 / 1.081 / 0.344, 4: 3.584 / 1.062 / 0.366 - shorter chunks do even out the waves' lifetimes (566-918 us
    // instead of 370-857 at level 1) but the launch does not get shorter (the remaining waves of a SIMD were using the
    // issue slots of the finished ones) and the carry pays for the extra chunks: 1.
#ifndef WM_CORE_ROW_SPLIT
#define WM_CORE_ROW_SPLIT 1
#endif
    constexpr int kRowSplit = WM_CORE_ROW_SPLIT;
    struct Cand { int seg, nseg, row_chunk; };
    auto fill = [&](const Cand& c) {
        pl.col_seg = c.seg; pl.col_nseg = c.nseg;
        pl.col_nchunks = (long long)W * pl.col_nseg;
        pl.col_wgs = ((pl.col_tiles * pl.col_nseg + 7) / 8) * 8;
        pl.row_chunk = c.row_chunk;
        pl.row_nchunks = (int)((L + c.row_chunk - 1) / c.row_chunk);
        pl.row_cpw = NW * kRowSplit;
        pl.row_wgs = (pl.row_nchunks + pl.row_cpw - 1) / pl.row_cpw;
    };
    {
        // shape -> plan, most recently used first; the least recently used entry is evicted at kPlanCache entries (a
        // serving process meets an open-ended set of image sizes - three pyramid levels each; a cache that stopped
        // inserting when full re-ran the 5-18 ms search on every call for every later shape)
        constexpr size_t kPlanCache = 1024;
        static std::mutex mu;
        static std::list<std::pair<std::array<int, 6>, Cand>> lru;
        static std::map<std::array<int, 6>, std::list<std::pair<std::array<int, 6>, Cand>>::iterator> index;
        const std::array<int, 6> key = {B, D, H, W, pl.NP, NW};
        std::lock_guard<std::mutex> lk(mu);
        const Cand* hit = nullptr;
        {
            const auto it = index.find(key);
            if (it != index.end()) { lru.splice(lru.begin(), lru, it->second); hit = &lru.front().second; }
        }
        if (hit) fill(*hit);
        else {
            double best = 1e300; Cand bc{((H + 15) / 16) * 16, 1, 32};
            int last_seg = -1;
            for (int n = 1; n <= 64; ++n) {
                int sg = (H + n - 1) / n;
                sg = ((sg + 15) / 16) * 16;
                if (n > 1 && sg < 32) break;
                if (sg == last_seg) continue;
                last_seg = sg;
                const int nn = (H + sg - 1) / sg;
                int last_chunk = -1;
                for (long long ct = RT < 128 ? RT : 128; ct >= 2; --ct) {      // tile rounds per row workgroup
                    const long long m = (RT + ct - 1) / ct;                    // row workgroups per direction
                    long long c = (L + m * NW * kRowSplit - 1) / (m * NW * kRowSplit);
                    c = ((c + 15) / 16) * 16;
                    if (c < 32) c = 32;
                    if ((int)c == last_chunk) continue;
                    last_chunk = (int)c;
                    const Cand cand{sg, nn, (int)c};
                    fill(cand);
                    const double t = core_makespan(pl, B, H, best);
                    if (t < best * 0.995) { best = t; bc = cand; }
                }
            }
            fill(bc);
            lru.emplace_front(key, bc);
            index[key] = lru.begin();
            if (lru.size() > kPlanCache) { index.erase(lru.back().first); lru.pop_back(); }
        }
    }
    pl.max_chunks = pl.col_nchunks > pl.row_nchunks ? pl.col_nchunks : pl.row_nchunks;
    if ((long long)B * (2LL * pl.row_wgs + 2LL * pl.col_wgs) > 0x7fffffffLL) return WM_EUNSUPPORTED;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    pl.half_bytes = up((size_t)pl.max_chunks * B * D * pl.NP * sizeof(float));
    pl.ytmp_bytes = merged ? up((size_t)B * D * L * sizeof(float)) : 0;
    pl.prep_bytes = up((size_t)4 * (pl.NP == 16 ? CoreCfg<16>::PREP : CoreCfg<32>::PREP) * sizeof(float));
    pl.total = 8 * pl.half_bytes + 3 * pl.ytmp_bytes + pl.prep_bytes;
    return WM_OK;
}

template <int NP, int NW, bool RHI, typename TP, bool VEC>
static int core_launch(const CoreArgs& a, const CorePlan& pl, bool do_prep, hipStream_t st) {
    constexpr int lds = core_lds_bytes<NP, NW>();
    // > 64 KB of dynamic LDS is an opt-in per function AND per device
    static bool configured[64] = {};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return WM_EHIP;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || !configured[dev]) {
            hipError_t e = hipFuncSetAttribute((const void*)ss2d_core_kernel<NP, NW, 1, RHI, TP, VEC>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)ss2d_core_kernel<NP, NW, 3, RHI, TP, VEC>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return WM_EHIP;
            if (dev >= 0 && dev < 64) configured[dev] = true;
        }
    }
    const dim3 grid((unsigned)(a.B * (2 * pl.row_wgs + 2 * pl.col_wgs))), block(64 * NW);
    const bool split = pl.row_nchunks > 1 || pl.col_nchunks > 1;
    if (do_prep) {   // parameters -> bf16 weight fragments, A * log2(e), per-channel constants (four small blocks)
        ProfScope ps(10, st);
        hipLaunchKernelGGL((ss2d_core_prep_kernel<NP, true>), dim3(4), dim3(256), 0, st, a.Wx, a.Wdt, a.dtb, a.A_logs, a.Ds,
                           const_cast<float*>(a.prep), a.D, a.N, a.R);
    }
    if (split) {
        {
            ProfScope ps(10, st);
            hipLaunchKernelGGL((ss2d_core_kernel<NP, NW, 1, RHI, TP, VEC>), grid, block, lds, st, a);
        }
        ProfScope ps(3, st);
        CarryBatch cb{};
        const int order[4] = {0, 2, 1, 3};
        for (int i = 0; i < 4; ++i) {
            const int k = order[i];
            cb.d[i] = CarryDir{a.wsP[k], a.wsH[k], nullptr, nullptr, (k & 1) ? (int)pl.col_nchunks : pl.row_nchunks, 0};
        }
        const long long nchains = (long long)a.B * a.D * NP;
        hipLaunchKernelGGL((selscan_carry_kernel<false>), dim3((unsigned)((nchains + 15) / 16), 1, 4), dim3(1024), 0,
                           st, cb, nchains);
    }
    {
        ProfScope ps(8, st);
        hipLaunchKernelGGL((ss2d_core_kernel<NP, NW, 3, RHI, TP, VEC>), grid, block, lds, st, a);
    }
    return launch_status();
}
}  // namespace wm

extern "C" {
size_t wm_ss2d_core_fwd_workspace_bytes(int B, int D, int H, int W, int N, int R, int merged) {
    // (sized for fp32 planes; bf16 planes need less for the merged mode's temporaries)
    CorePlan pl;
    if (core_plan(pl, B, D, H, W, N, R, merged == 1) != WM_OK) return 0;
    return pl.total;
}

size_t wm_ss2d_core_prep_bytes(int N) {
    if (N <= 0 || N > 32) return 0;
    return (size_t)4 * (N <= 16 ? CoreCfg<16>::PREP : CoreCfg<32>::PREP) * sizeof(float);
}

int wm_ss2d_core_prep(const float* x_proj_weight, const float* dt_projs_weight, const float* dt_projs_bias,
                      const float* A_logs, const float* Ds, void* prepared, int D, int N, int R, void* stream) {
    if (D <= 0 || N <= 0 || R <= 0) return WM_EINVAL;
    if (N > 32 || R > 4 || D > 64) return WM_EUNSUPPORTED;
    if (!x_proj_weight || !dt_projs_weight || !dt_projs_bias || !A_logs || !Ds || !prepared) return WM_ENULL;
    if (!aligned16(prepared)) return WM_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (N <= 16)
        hipLaunchKernelGGL((ss2d_core_prep_kernel<16, true>), dim3(4), dim3(256), 0, st, x_proj_weight, dt_projs_weight,
                           dt_projs_bias, A_logs, Ds, (float*)prepared, D, N, R);
    else
        hipLaunchKernelGGL((ss2d_core_prep_kernel<32, true>), dim3(4), dim3(256), 0, st, x_proj_weight, dt_projs_weight,
                           dt_projs_bias, A_logs, Ds, (float*)prepared, D, N, R);
    return launch_status();
}

int wm_ss2d_core_plan(int B, int D, int H, int W, int N, int R, int* out10) {
    if (!out10) return WM_ENULL;
    CorePlan pl;
    const int rc = core_plan(pl, B, D, H, W, N, R, 0);
    if (rc) return rc;
    out10[0] = pl.NW; out10[1] = pl.col_seg; out10[2] = pl.col_nseg; out10[3] = pl.col_tiles; out10[4] = pl.col_wgs;
    out10[5] = pl.row_chunk; out10[6] = pl.row_nchunks; out10[7] = pl.row_wgs;      // (row_wgs x 2 NW chunks, taken on demand)
    out10[8] = B * (2 * pl.row_wgs + 2 * pl.col_wgs);
    out10[9] = (int)(100.0 * core_makespan(pl, B, H, 1e300));
    return WM_OK;
}

int wm_ss2d_core_fwd(const void* x, const float* x_proj_weight, const float* dt_projs_weight,
                     const float* dt_projs_bias, const float* A_logs, const float* Ds, void* y_row_fwd,
                     void* y_row_rev, void* y_col_fwd, void* y_col_rev, int merged, void* workspace,
                     size_t workspace_bytes, const void* prepared, int B, int D, int H, int W, int N, int R, int plane_dtype,
                     void* stream) {
    if (B == 0 || D == 0 || H == 0 || W == 0) return (B < 0 || D < 0 || H < 0 || W < 0) ? WM_EINVAL : WM_OK;
    if (plane_dtype != WM_F32 && plane_dtype != WM_BF16) return WM_EUNSUPPORTED;
    CorePlan pl;
    int rc = core_plan(pl, B, D, H, W, N, R, merged == 1);
    if (rc) return rc;
    if (!x || !x_proj_weight || !dt_projs_weight || !dt_projs_bias || !A_logs || !Ds || !y_row_fwd) return WM_ENULL;
    if (merged < 0 || merged > 1) return WM_EINVAL;       // (2 was round 4's two-plane mode: measured slower, deleted in round 5)
    if (merged == 0 && (!y_row_rev || !y_col_fwd || !y_col_rev)) return WM_ENULL;
    if (!workspace) return WM_ENULL;
    if (workspace_bytes < pl.total) return WM_EWORKSPACE;
    if (!aligned16(workspace)) return WM_EALIGN;
    // 16-byte tile accesses when the map width allows them (every size the network itself produces: it pads its input
    // to multiples of 8); otherwise the same kernels with element-wise tile accesses (fp32 planes only).
    const bool planes16 = aligned16(x) && aligned16(y_row_fwd) &&
                          (merged || (aligned16(y_row_rev) && aligned16(y_col_fwd) && aligned16(y_col_rev)));
    const bool vec = (W % 4 == 0) && planes16;
    if (!vec && plane_dtype != WM_F32) return (W % 4 == 0) ? WM_EALIGN : WM_EUNSUPPORTED;
    {
        auto mis4 = [](const void* q) { return q && ((uintptr_t)q & 3) != 0; };
        if (mis4(x) || mis4(y_row_fwd) || (!merged && (mis4(y_row_rev) || mis4(y_col_fwd) || mis4(y_col_rev)))) return WM_EALIGN;
    }
    hipStream_t st = (hipStream_t)stream;
    CoreArgs a;
    a.x = x; a.Wx = x_proj_weight; a.Wdt = dt_projs_weight; a.dtb = dt_projs_bias; a.A_logs = A_logs; a.Ds = Ds;
    char* w = (char*)workspace;
    for (int k = 0; k < 4; ++k) {
        a.wsP[k] = (float*)(w + (size_t)(2 * k) * pl.half_bytes);
        a.wsH[k] = (float*)(w + (size_t)(2 * k + 1) * pl.half_bytes);
    }
    char* yt = w + 8 * pl.half_bytes;
    a.prep = prepared ? (const float*)prepared : (const float*)(yt + 3 * pl.ytmp_bytes);
    if (prepared && !aligned16(prepared)) return WM_EALIGN;
    // direction k: 0 row forward, 1 column forward, 2 row reversed, 3 column reversed (the reference's xs order, :451-452)
    a.y[0] = y_row_fwd;
    a.y[1] = merged ? (void*)yt : y_col_fwd;
    a.y[2] = merged ? (void*)(yt + pl.ytmp_bytes) : y_row_rev;
    a.y[3] = merged ? (void*)(yt + 2 * pl.ytmp_bytes) : y_col_rev;
    a.B = B; a.D = D; a.H = H; a.W = W; a.L = H * W; a.N = N; a.R = R;
    a.row_chunk = pl.row_chunk; a.row_nchunks = pl.row_nchunks; a.row_wgs = pl.row_wgs; a.row_cpw = pl.row_cpw;
    a.stamps = nullptr;
#if WM_CORE_STAMP
    {   // diagnostics build: WM_CORE_STAMPS=<device pointer, decimal> receives [workgroups][waves][8] cycle totals of the scan launch
        const char* e = getenv("WM_CORE_STAMPS");
        a.stamps = e ? (unsigned long long*)strtoull(e, nullptr, 10) : nullptr;
    }
#endif
    {   // WM_CORE_DIRMASK (tools only): run a subset of the four directions, e.g. 1 = row forward alone
        static const int mask = [] { const char* e = getenv("WM_CORE_DIRMASK"); return e ? atoi(e) : 15; }();
        a.dirmask = mask;
    }
    a.col_seg = pl.col_seg; a.col_nseg = pl.col_nseg; a.col_tiles = pl.col_tiles; a.col_wgs = pl.col_wgs;
#ifndef WM_CORE_NW
#define WM_CORE_NW 16
#endif
#define WM_CORE_GO(TP, VEC)                                                                                                   \
    do {                                                                                                                      \
        const bool dp = prepared == nullptr;                                                                                  \
        if (pl.NP == 16) rc = R > 2 ? core_launch<16, WM_CORE_NW, true, TP, VEC>(a, pl, dp, st) : core_launch<16, WM_CORE_NW, false, TP, VEC>(a, pl, dp, st); \
        else rc = R > 2 ? core_launch<32, 8, true, TP, VEC>(a, pl, dp, st) : core_launch<32, 8, false, TP, VEC>(a, pl, dp, st); \
    } while (0)
    if (!vec) WM_CORE_GO(float, false);
    else if (plane_dtype == WM_F32) WM_CORE_GO(float, true);
    else WM_CORE_GO(bf16_t, true);
#undef WM_CORE_GO
    if (rc) return rc;
    if (merged) {
        const long long n = (long long)B * D * a.L, n4 = n / 4;
        ProfScope ps(8, st);
        if (!vec || (n & 3))
            hipLaunchKernelGGL((ss2d_sum4_kernel<float, false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (float*)a.y[0],
                               (const float*)a.y[2], (const float*)a.y[1], (const float*)a.y[3], n);
        else if (plane_dtype == WM_F32)
            hipLaunchKernelGGL((ss2d_sum4_kernel<float, true>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (float*)a.y[0],
                               (const float*)a.y[2], (const float*)a.y[1], (const float*)a.y[3], n4);
        else
            hipLaunchKernelGGL((ss2d_sum4_kernel<bf16_t, true>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (bf16_t*)a.y[0],
                               (const bf16_t*)a.y[2], (const bf16_t*)a.y[1], (const bf16_t*)a.y[3], n4);
    }
    return launch_status();
}

}  // extern "C"
namespace wm {
// ---- second generation (ss2d_core_bwd.hip.h) ---------------------------------------------------------------------------
struct CoreBwdPlan2 {
    BwdPlan scan; long long L; int NP, NWT, slices;
    size_t prep_bytes, wt_bytes, map_bytes, scan_bytes, wpart_bytes, wsum_bytes, total;
};
static int core_bwd_plan2(CoreBwdPlan2& pl, int B, int D, int H, int W, int N, int R) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || N <= 0 || R <= 0) return WM_EINVAL;
    if (N > 32 || R > kRecPad || D > 64) return WM_EUNSUPPORTED;
    pl.L = (long long)H * W;
    if (pl.L > 0x7fffffffLL) return WM_EUNSUPPORTED;
    if (B > 65535) return WM_EUNSUPPORTED;
    int rc = bwd_plan(pl.scan, B, D, (int)pl.L, N, 1, kPartPadFused);
    if (rc) return rc;
    pl.NP = N <= 16 ? 16 : 32;
    pl.NWT = pl.NP == 16 ? BwdCfg<16>::NWT : BwdCfg<32>::NWT;
    const long long nb = (long long)B * pl.scan.nblocks;
    pl.slices = (int)(nb < 64 ? 1 : (nb / 32 > 64 ? 64 : nb / 32));           // >= 32 partials per slice, <= 64 slices
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    pl.prep_bytes = up((size_t)4 * (pl.NP == 16 ? CoreCfg<16>::PREP : CoreCfg<32>::PREP) * sizeof(float));
    pl.wt_bytes = up((size_t)4 * (pl.NP == 16 ? BwdCfg<16>::WT_U4 : BwdCfg<32>::WT_U4) * sizeof(uint4));
    pl.map_bytes = up((size_t)B * D * pl.L * sizeof(float));
    pl.scan_bytes = up(pl.scan.total);
    pl.wpart_bytes = up((size_t)nb * pl.NWT * 256 * sizeof(float));
    pl.wsum_bytes = up((size_t)4 * pl.slices * pl.NWT * 256 * sizeof(float));
    pl.total = pl.prep_bytes + pl.wt_bytes + 4 * pl.map_bytes + 4 * pl.scan_bytes + 4 * pl.wpart_bytes + pl.wsum_bytes;
    return WM_OK;
}

template <int NP>
static int core_bwd_v2(const CoreBwdPlan2& pl, const float* x, const float* x_proj_weight, const float* dt_projs_weight,
                       const float* dt_projs_bias, const float* A_logs, const float* Ds, const float* dy_row_fwd,
                       const float* dy_row_rev, const float* dy_col_fwd, const float* dy_col_rev, float* dx,
                       float* dx_proj_weight, float* ddt_projs_weight, float* ddt_projs_bias, float* dA_logs, float* dDs,
                       void* workspace, int B, int D, int H, int W, int N, int R, hipStream_t st) {
    using Cfg = BwdCfg<NP>;
    const long long L = pl.L;
    char* w = (char*)workspace;
    float* prep = (float*)w; w += pl.prep_bytes;
    uint4* wT = (uint4*)w; w += pl.wt_bytes;
    float* xT = (float*)w; w += pl.map_bytes;
    float* dyTa = (float*)w; w += pl.map_bytes;
    float* dyTb = (float*)w; w += pl.map_bytes;
    float* dxT = (float*)w; w += pl.map_bytes;
    char* scan_ws[4];
    for (int k = 0; k < 4; ++k) { scan_ws[k] = w; w += pl.scan_bytes; }
    float* wpart[4];
    for (int k = 0; k < 4; ++k) { wpart[k] = (float*)w; w += pl.wpart_bytes; }
    float* wsum = (float*)w;

    ProfScope ps(12, st);
    // parameters -> forward-style fragments / constants, and the transposed fragments of the dx product
    hipLaunchKernelGGL((ss2d_core_prep_kernel<NP>), dim3(4), dim3(256), 0, st, x_proj_weight, dt_projs_weight, dt_projs_bias,
                       A_logs, Ds, prep, D, N, R);
    hipLaunchKernelGGL((core_bwd_prep_kernel<NP>), dim3(4), dim3(256), 0, st, x_proj_weight, wT, D, N, R);
    {   // the column directions scan the transposed map
        const dim3 tg((unsigned)((W + 31) / 32), (unsigned)((H + 31) / 32), (unsigned)(B * D)), tb(32, 8);
        hipLaunchKernelGGL(transpose_planes_kernel, tg, tb, 0, st, x, xT, H, W, 0);
        hipLaunchKernelGGL(transpose_planes_kernel, tg, tb, 0, st, dy_col_fwd, dyTa, H, W, 0);
        if (dy_col_rev != dy_col_fwd) hipLaunchKernelGGL(transpose_planes_kernel, tg, tb, 0, st, dy_col_rev, dyTb, H, W, 0);
    }
    const float* dyT_rev = dy_col_rev != dy_col_fwd ? dyTb : dyTa;
    CoreBwdArgs a[4];
    float* seg[4];
    const int CP = R + 2 * N;
    for (int k = 0; k < 4; ++k) {
        const bool col = k & 1;
        CoreBwdArgs& q = a[k];
        q.x = col ? xT : x;
        q.dy = k == 0 ? dy_row_fwd : k == 2 ? dy_row_rev : k == 1 ? (const float*)dyTa : dyT_rev;
        q.dx = col ? dxT : dx;
        q.prep = prep + (size_t)k * CoreCfg<NP>::PREP;
        q.wT = wT + (size_t)k * Cfg::WT_U4;
        q.WxR = x_proj_weight + (size_t)k * CP * D;
        ScanBwdArgs t{};
        bwd_bind_workspace(t, pl.scan, scan_ws[k], seg[k]);
        q.wsP = t.wsP; q.wsH = t.wsH; q.wsPr = t.wsPr; q.wsG = t.wsG; q.wsHl = t.wsHl; q.wsS = t.wsS; q.part = t.part;
        q.wpart = wpart[k];
        q.batch = B; q.dim = D; q.L = (int)L; q.N = N; q.R = R;
        q.nchunks = pl.scan.nchunks; q.cpb = pl.scan.cpb; q.nblocks = pl.scan.nblocks;
        q.accumulate = k >= 2;                           // a layout's first direction writes dx, its second adds
    }
    const bool vec = (L % 4 == 0) && aligned16(x) && aligned16(dx) && aligned16(dy_row_fwd) && aligned16(dy_row_rev) &&
                     aligned16(xT) && aligned16(dyTa) && aligned16(dyTb) && aligned16(dxT);
    const dim3 grid((unsigned)pl.scan.nblocks, (unsigned)B);
    if (pl.scan.nchunks > 1) {
        const dim3 g4(grid.x, grid.y, 4);
        if (vec) hipLaunchKernelGGL((core_bwd_reduce_kernel<NP, true>), g4, dim3(64), 0, st, a[0], a[1], a[2], a[3]);
        else hipLaunchKernelGGL((core_bwd_reduce_kernel<NP, false>), g4, dim3(64), 0, st, a[0], a[1], a[2], a[3]);
        if (pl.scan.nblocks > 1) {                       // all eight carries (four forward, four adjoint) in one batch
            CarryBatch cb{};
            const int nsegs = (int)carry_nsegs(pl.scan.nblocks);
            const size_t one = (size_t)nsegs * pl.scan.chains;
            for (int k = 0; k < 4; ++k) {
                cb.d[2 * k] = CarryDir{a[k].wsP, a[k].wsH, seg[k], seg[k] + one, pl.scan.nblocks, nsegs};
                cb.d[2 * k + 1] = CarryDir{a[k].wsPr, a[k].wsG, seg[k] + 2 * one, seg[k] + 3 * one, pl.scan.nblocks, nsegs};
            }
            launch_carry_batch(cb, 8, pl.scan.chains, st);
        }
    }
    const int order[4] = {0, 2, 1, 3};
    static const int dirmask = [] { const char* e = getenv("WM_CORE_BWD_DIRMASK"); return e ? atoi(e) : 15; }();   // tools only
    for (int i = 0; i < 4; ++i) {
        const int k = order[i];
        if (!((dirmask >> k) & 1)) continue;
        const dim3 blk(64 * Cfg::NW);
        if (k < 2) { if (vec) hipLaunchKernelGGL((core_bwd_chunk_kernel<NP, true, false>), grid, blk, 0, st, a[k]);
                     else hipLaunchKernelGGL((core_bwd_chunk_kernel<NP, false, false>), grid, blk, 0, st, a[k]); }
        else       { if (vec) hipLaunchKernelGGL((core_bwd_chunk_kernel<NP, true, true>), grid, blk, 0, st, a[k]);
                     else hipLaunchKernelGGL((core_bwd_chunk_kernel<NP, false, true>), grid, blk, 0, st, a[k]); }
    }
    {
        const dim3 tg((unsigned)((H + 31) / 32), (unsigned)((W + 31) / 32), (unsigned)(B * D)), tb(32, 8);
        hipLaunchKernelGGL(transpose_planes_kernel, tg, tb, 0, st, (const float*)dxT, dx, W, H, 1);   // dx += (dx^T)^T
    }
    CoreBwdFinishArgs f;
    for (int k = 0; k < 4; ++k) { f.part[k] = a[k].part; f.wpart[k] = wpart[k]; }
    f.wsum = wsum; f.A_logs = A_logs; f.dA_logs = dA_logs; f.dDs = dDs; f.dbias = ddt_projs_bias; f.dWdt = ddt_projs_weight;
    f.dWx = dx_proj_weight; f.batch = B; f.dim = D; f.N = N; f.R = R; f.NP = NP; f.nblocks = pl.scan.nblocks; f.slices = pl.slices;
    hipLaunchKernelGGL(core_bwd_finish_kernel, dim3((unsigned)D, 1, 4), dim3(256), 0, st, f);
    hipLaunchKernelGGL(core_bwd_wsum_kernel, dim3((unsigned)Cfg::NWT, (unsigned)pl.slices, 4), dim3(256), 0, st, f, (int)Cfg::NWT);
    hipLaunchKernelGGL(core_bwd_wfin_kernel, dim3((unsigned)Cfg::NWT, 1, 4), dim3(256), 0, st, f, (int)Cfg::NWT);
    return launch_status();
}

}  // namespace wm
extern "C" {

size_t wm_ss2d_core_bwd_workspace_bytes(int B, int D, int H, int W, int N, int R) {
    CoreBwdPlan2 pl;
    if (core_bwd_plan2(pl, B, D, H, W, N, R) != WM_OK) return 0;
    return pl.total;
}

int wm_ss2d_core_bwd(const float* x, const float* x_proj_weight, const float* dt_projs_weight,
                     const float* dt_projs_bias, const float* A_logs, const float* Ds, const float* dy_row_fwd,
                     const float* dy_row_rev, const float* dy_col_fwd, const float* dy_col_rev, float* dx,
                     float* dx_proj_weight, float* ddt_projs_weight, float* ddt_projs_bias, float* dA_logs, float* dDs,
                     void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int N, int R, void* stream) {
    if (B == 0 || D == 0 || H == 0 || W == 0) return (B < 0 || D < 0 || H < 0 || W < 0) ? WM_EINVAL : WM_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool nul = !x || !x_proj_weight || !dt_projs_weight || !dt_projs_bias || !A_logs || !Ds || !dy_row_fwd || !dy_row_rev ||
                     !dy_col_fwd || !dy_col_rev || !dx || !dx_proj_weight || !ddt_projs_weight || !ddt_projs_bias || !dA_logs ||
                     !dDs || !workspace;
    CoreBwdPlan2 pl;
    int rc = core_bwd_plan2(pl, B, D, H, W, N, R);
    if (rc) return rc;
    if (nul) return WM_ENULL;
    if (workspace_bytes < pl.total) return WM_EWORKSPACE;
    if (!aligned16(workspace)) return WM_EALIGN;
    if (pl.NP == 16)
        return core_bwd_v2<16>(pl, x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, dy_row_fwd, dy_row_rev, dy_col_fwd,
                               dy_col_rev, dx, dx_proj_weight, ddt_projs_weight, ddt_projs_bias, dA_logs, dDs, workspace, B, D, H, W, N, R, st);
    return core_bwd_v2<32>(pl, x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, dy_row_fwd, dy_row_rev, dy_col_fwd,
                           dy_col_rev, dx, dx_proj_weight, ddt_projs_weight, ddt_projs_bias, dA_logs, dDs, workspace, B, D, H, W, N, R, st);
}

#define WM_LFSS_DISPATCH(PROFCLASS, KERNEL, ...)                                                   \
    do {                                                                                           \
        const long long total = (long long)B * L;                                                  \
        if (total == 0) return WM_OK;                                                              \
        const dim3 grid((unsigned)((total + 255) / 256)), block(256);                              \
        hipStream_t st = (hipStream_t)stream;                                                      \
        ProfScope ps(PROFCLASS, st);                                                               \
        if (C == 32) hipLaunchKernelGGL((KERNEL<32>), grid, block, 0, st, __VA_ARGS__);            \
        else if (C == 16) hipLaunchKernelGGL((KERNEL<16>), grid, block, 0, st, __VA_ARGS__);       \
        else if (C == 8) hipLaunchKernelGGL((KERNEL<8>), grid, block, 0, st, __VA_ARGS__);         \
        else return WM_EUNSUPPORTED;                                                               \
        return launch_status();                                                                    \
    } while (0)

int wm_lfss_in_fwd(const float* tok, int tok_nchw, const float* ln_w, const float* ln_b, float ln_eps,
                   const float* in_proj_weight, void* x_, void* z_, int B, int64_t L, int C, int plane_dtype, void* stream) {
    if (B < 0 || L < 0) return WM_EINVAL;
    if (plane_dtype != WM_F32 && !(plane_dtype == WM_BF16 && C == 32)) return WM_EUNSUPPORTED;    // bf16 planes: C = 32 kernels
    float* x = (float*)x_; float* z = (float*)z_;
    // z == NULL (C == 32 only): the gate half is not written - the block's wm_lfss_mid_rz_fwd recomputes it
    if (B && L && (!tok || !ln_w || !ln_b || !in_proj_weight || !x || (!z && C != 32))) return WM_ENULL;
    if (!tok_nchw && !aligned16(tok)) return WM_EALIGN;
    if (C == 32 && B && L) {
        const int ngl = (int)((L + 63) / 64);
        const long long ngroups = (long long)B * ngl;
        const int gpw = lfss_groups_per_wave(ngroups, 1024 * WM_LFSS_IN_WAVES);
        const long long waves = (ngroups + gpw - 1) / gpw;
        hipStream_t st = (hipStream_t)stream;
        ProfScope ps(5, st);
        if (plane_dtype == WM_F32)
            hipLaunchKernelGGL(lfss_in_mfma_kernel<float>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, tok, tok_nchw, ln_w,
                               ln_b, ln_eps, in_proj_weight, x, z, B, (long long)L, ngl, ngroups, gpw);
        else
            hipLaunchKernelGGL(lfss_in_mfma_kernel<bf16_t>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, tok, tok_nchw, ln_w,
                               ln_b, ln_eps, in_proj_weight, (bf16_t*)x_, (bf16_t*)z_, B, (long long)L, ngl, ngroups, gpw);
        return launch_status();
    }
    WM_LFSS_DISPATCH(5, lfss_in_kernel, tok, tok_nchw, ln_w, ln_b, ln_eps, in_proj_weight, x, z, B, (long long)L);
}

int wm_lfss_mid_fwd(const void* ysum_, int ny, int64_t ystride, const void* z_, const float* tok, int tok_nchw, const float* out_norm_w,
                    const float* out_norm_b, float out_norm_eps, const float* out_proj_weight,
                    const float* skip_scale, const float* ln2_w, const float* ln2_b, float ln2_eps,
                    const float* conv1_weight, const float* conv1_bias, float* tok1, void* f_, int B, int64_t L,
                    int C, int plane_dtype, void* stream) {
    if (B < 0 || L < 0 || (ny != 1 && ny != 4)) return WM_EINVAL;
    if (plane_dtype != WM_F32 && !(plane_dtype == WM_BF16 && C == 32)) return WM_EUNSUPPORTED;
    const float* ysum = (const float*)ysum_; const float* z = (const float*)z_; float* f = (float*)f_;
    if (B && L && (!ysum || !z || !tok || !out_norm_w || !out_norm_b || !out_proj_weight || !skip_scale || !ln2_w ||
                   !ln2_b || !conv1_weight || !conv1_bias || !tok1 || !f)) return WM_ENULL;
    if ((!tok_nchw && !aligned16(tok)) || !aligned16(tok1)) return WM_EALIGN;
    if (C == 32 && B && L) {                                 // the shipped width: projections on the matrix cores
        const int ngl = (int)((L + 63) / 64);
        const long long ngroups = (long long)B * ngl;
        const int gpw = lfss_groups_per_wave(ngroups, 1024 * WM_LFSS_MID_WAVES);
        const long long waves = (ngroups + gpw - 1) / gpw;
        hipStream_t st = (hipStream_t)stream;
        ProfScope ps(9, st);
#define WM_MID(NY, TP) hipLaunchKernelGGL((lfss_mid_mfma_kernel<NY, TP>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, \
                           (const TP*)ysum_, (long long)ystride, (const TP*)z_, tok, tok_nchw,                                     \
                           out_norm_w, out_norm_b, out_norm_eps, out_proj_weight, skip_scale, ln2_w, ln2_b, ln2_eps,               \
                           conv1_weight, conv1_bias, tok1, (TP*)f_, B, (long long)L, ngl, ngroups, gpw)
        if (plane_dtype == WM_F32) { if (ny == 4) WM_MID(4, float); else WM_MID(1, float); }
        else { if (ny == 4) WM_MID(4, bf16_t); else WM_MID(1, bf16_t); }
#undef WM_MID
        return launch_status();
    }
    WM_LFSS_DISPATCH(9, lfss_mid_kernel, ysum, ny, (long long)ystride, z, tok, tok_nchw, out_norm_w, out_norm_b, out_norm_eps, out_proj_weight,
                     skip_scale, ln2_w, ln2_b, ln2_eps, conv1_weight, conv1_bias, tok1, f, B, (long long)L);
}

// wm_lfss_mid_fwd with the gate z RECOMPUTED from `tok` (ln_1 + in_proj rows [D, 2D) on the matrix cores, bit-identical to
// wm_lfss_in_fwd's z in fp32 planes) instead of read: C == 32 only (WM_EUNSUPPORTED otherwise: callers keep z and wm_lfss_mid_fwd).
int wm_lfss_mid_rz_fwd(const void* ysum_, int ny, int64_t ystride, const float* tok, int tok_nchw, const float* ln1_w,
                       const float* ln1_b, float ln1_eps, const float* in_proj_weight, const float* out_norm_w,
                       const float* out_norm_b, float out_norm_eps, const float* out_proj_weight,
                       const float* skip_scale, const float* ln2_w, const float* ln2_b, float ln2_eps,
                       const float* conv1_weight, const float* conv1_bias, float* tok1, void* f_, int B, int64_t L,
                       int C, int plane_dtype, void* stream) {
    if (B < 0 || L < 0 || (ny != 1 && ny != 4)) return WM_EINVAL;
    if (C != 32 || (plane_dtype != WM_F32 && plane_dtype != WM_BF16)) return WM_EUNSUPPORTED;
    if (B == 0 || L == 0) return WM_OK;
    if (!ysum_ || !tok || !ln1_w || !ln1_b || !in_proj_weight || !out_norm_w || !out_norm_b || !out_proj_weight || !skip_scale ||
        !ln2_w || !ln2_b || !conv1_weight || !conv1_bias || !tok1 || !f_) return WM_ENULL;
    if ((!tok_nchw && !aligned16(tok)) || !aligned16(tok1)) return WM_EALIGN;
    const int ngl = (int)((L + 63) / 64);
    const long long ngroups = (long long)B * ngl;
    const int gpw = lfss_groups_per_wave(ngroups, 1024 * WM_LFSS_MID_RZ_WAVES);
    const long long waves = (ngroups + gpw - 1) / gpw;
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(9, st);
#define WM_MIDZ(NY, TP) hipLaunchKernelGGL((lfss_mid_mfma_kernel<NY, TP, true>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, \
                           (const TP*)ysum_, (long long)ystride, (const TP*)nullptr, tok, tok_nchw,                                  \
                           out_norm_w, out_norm_b, out_norm_eps, out_proj_weight, skip_scale, ln2_w, ln2_b, ln2_eps,                 \
                           conv1_weight, conv1_bias, tok1, (TP*)f_, B, (long long)L, ngl, ngroups, gpw, ln1_w, ln1_b, ln1_eps,       \
                           in_proj_weight)
    if (plane_dtype == WM_F32) { if (ny == 4) WM_MIDZ(4, float); else WM_MIDZ(1, float); }
    else { if (ny == 4) WM_MIDZ(4, bf16_t); else WM_MIDZ(1, bf16_t); }
#undef WM_MIDZ
    return launch_status();
}

int wm_lfss_out_fwd(const void* fc_, const float* tok1, const float* conv3_weight, const float* conv3_bias,
                    const float* skip_scale2, float* out, int out_nchw, int B, int64_t L, int C, int plane_dtype, void* stream) {
    if (B < 0 || L < 0) return WM_EINVAL;
    if (plane_dtype != WM_F32 && !(plane_dtype == WM_BF16 && C == 32)) return WM_EUNSUPPORTED;
    const float* fc = (const float*)fc_;
    if (B && L && (!fc || !tok1 || !conv3_weight || !conv3_bias || !skip_scale2 || !out)) return WM_ENULL;
    if (!aligned16(tok1) || (!out_nchw && !aligned16(out))) return WM_EALIGN;
    if (C == 32 && B && L) {
        const int ngl = (int)((L + 63) / 64);
        const long long ngroups = (long long)B * ngl;
        const int gpw = lfss_groups_per_wave(ngroups, 2048);
        const long long waves = (ngroups + gpw - 1) / gpw;
        hipStream_t st = (hipStream_t)stream;
        ProfScope ps(11, st);
        if (plane_dtype == WM_F32)
            hipLaunchKernelGGL(lfss_out_mfma_kernel<float>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, fc, tok1,
                               conv3_weight, conv3_bias, skip_scale2, out, out_nchw, B, (long long)L, ngl, ngroups, gpw);
        else
            hipLaunchKernelGGL(lfss_out_mfma_kernel<bf16_t>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, (const bf16_t*)fc_,
                               tok1, conv3_weight, conv3_bias, skip_scale2, out, out_nchw, B, (long long)L, ngl, ngroups, gpw);
        return launch_status();
    }
    WM_LFSS_DISPATCH(11, lfss_out_kernel, fc, tok1, conv3_weight, conv3_bias, skip_scale2, out, out_nchw, B, (long long)L);
}

int wm_lfss_out_conv_fwd(const void* f_, const float* conv2_weight, const float* conv2_bias, const float* tok1,
                         const float* conv3_weight, const float* conv3_bias, const float* skip_scale2, float* out,
                         int out_nchw, int B, int H, int W, int C, int plane_dtype, void* stream) {
    if (B < 0 || H < 0 || W < 0) return WM_EINVAL;
    if (C != 32 || W % 32 != 0) return WM_EUNSUPPORTED;          // callers fall back to wm_dwconv3x3_fwd + wm_lfss_out_fwd
    if (plane_dtype != WM_F32 && plane_dtype != WM_BF16) return WM_EUNSUPPORTED;
    const long long L = (long long)H * W;
    if (B == 0 || L == 0) return WM_OK;
    if (L > 0x1fffffffLL) return WM_EUNSUPPORTED;                // 32-bit byte offsets inside one channel plane
    if (!f_ || !conv2_weight || !tok1 || !conv3_weight || !conv3_bias || !skip_scale2 || !out) return WM_ENULL;
    if (!aligned16(tok1) || (!out_nchw && !aligned16(out))) return WM_EALIGN;
    const int ngl = (int)((L + 63) / 64);
    const long long ngroups = (long long)B * ngl;
    const int gpw = lfss_groups_per_wave(ngroups, 2048);
    const long long waves = (ngroups + gpw - 1) / gpw;
    // accumulating row-window form (round 6, lfss_out_conv_acc_kernel<4>: four output rows of a 64-column strip per wave pass, the
    // closing product accumulated in registers over groups of eight gated channels, one coalesced load per tap row + lane shifts):
    // W % 64 == 0 and >= 2^18 positions.  Measured (profiles/r06/lfss_out_conv_forms.txt, ms per call at UHD levels 1 / 2): banded
    // one-row form 0.464 / 0.087 (round 4), row windows in LDS R = 2 0.405 / 0.088 (round 4-5; deleted), this form R = 2 0.401 /
    // 0.084, R = 4 0.346 / 0.069; its double-buffered variant 0.338-0.351 / 0.074 (not kept).  Bit-identical outputs in all forms.
    if (W % 64 == 0 && (long long)B * L >= (1ll << 18)) {
        constexpr int R = 4;
        const int nstrips = W / 64, nbands = (H + R - 1) / R;
        // bands per walk (a wave walks consecutive bands of its strip): enough walks for two rounds of the 2,048 resident waves
        int bpw = (int)(((long long)B * nbands * nstrips + 4095) / 4096);
        if (bpw < 1) bpw = 1;
        if (bpw > 8) bpw = 8;
        const int nchunks = (nbands + bpw - 1) / bpw;
        const long long nwalks = (long long)B * nchunks * nstrips;
        hipStream_t st3 = (hipStream_t)stream;
        ProfScope ps3(11, st3);
#define WM_ACC(TP) hipLaunchKernelGGL((lfss_out_conv_acc_kernel<R, TP>), dim3((unsigned)((nwalks + 3) / 4)), dim3(256), 0, st3, \
                                      (const TP*)f_, conv2_weight, conv2_bias, tok1, conv3_weight, conv3_bias, skip_scale2, out,   \
                                      out_nchw, B, H, W, nstrips, nbands, bpw, nchunks, nwalks)
        if (plane_dtype == WM_F32) WM_ACC(float); else WM_ACC(bf16_t);
#undef WM_ACC
        return launch_status();
    }
    // groups per image row for the kernel's banded (column-major) group order (0: linear order, maps whose width is not a multiple of 64)
    const int gpr = (W % 64 == 0) ? W / 64 : 0;
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(11, st);
    if (plane_dtype == WM_F32)
        hipLaunchKernelGGL(lfss_out_conv_mfma_kernel<float>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, (const float*)f_,
                           conv2_weight, conv2_bias, tok1, conv3_weight, conv3_bias, skip_scale2, out, out_nchw, B, H, W, ngl,
                           ngroups, gpw, gpr);
    else
        hipLaunchKernelGGL(lfss_out_conv_mfma_kernel<bf16_t>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st,
                           (const bf16_t*)f_, conv2_weight, conv2_bias, tok1, conv3_weight, conv3_bias, skip_scale2, out, out_nchw,
                           B, H, W, ngl, ngroups, gpw, gpr);
    return launch_status();
}

// nn.Sequential(nn.PixelUnshuffle(r), nn.Conv2d(r r Cin, Cout, 1)) of the UNet's image inputs (reference :1014-1025, :1043-1045) in one
// kernel: an r x r / stride r convolution read straight from the image (patchify.hip.h).
int wm_patchify_conv_fwd(const float* img, const float* weight, const float* bias, float* y, int B, int Cin, int Cout, int H, int W,
                         int r, void* stream) {
    if (B < 0 || Cin <= 0 || Cout <= 0 || H < 0 || W < 0) return WM_EINVAL;
    if (r != 2 && r != 4 && r != 8) return WM_EUNSUPPORTED;
    if (H % r || W % r) return WM_EINVAL;
    if (Cout != 16 && Cout != 32 && Cout != 48 && Cout != 64) return WM_EUNSUPPORTED;
    const size_t lds = (size_t)Cin * r * r * Cout * sizeof(float);
    if (lds > 64 * 1024) return WM_EUNSUPPORTED;
    if (B == 0 || H == 0 || W == 0) return WM_OK;
    if (!img || !weight || !y) return WM_ENULL;
    if (!aligned16(img)) return WM_EALIGN;
    if ((long long)B * Cin * H * W >= (1ll << 40)) return WM_EUNSUPPORTED;
    const int Ho = H / r, Wo = W / r;
    const int spr = (Wo + 255) / 256;
    const long long nsegs = (long long)B * Ho * spr;
    const int spb = (int)((nsegs + 4095) / 4096);
    const long long blocks = (nsegs + spb - 1) / spb;
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(18, st);
#define WM_PF(R, CO) hipLaunchKernelGGL((patchify_conv_kernel<R, CO>), dim3((unsigned)blocks), dim3(256), lds, st, img, weight, bias, y, \
                                        B, Cin, H, W, spr, nsegs, spb)
#define WM_PFR(R) do { if (Cout == 16) WM_PF(R, 16); else if (Cout == 32) WM_PF(R, 32); else if (Cout == 48) WM_PF(R, 48); else WM_PF(R, 64); } while (0)
    if (r == 2) WM_PFR(2); else if (r == 4) WM_PFR(4); else WM_PFR(8);
#undef WM_PFR
#undef WM_PF
    return launch_status();
}

int wm_layernorm2d_fwd(const float* x, const float* weight, const float* bias, float eps, float* y, int B,
                       int64_t L, int C, void* stream) {
    if (B < 0 || L < 0) return WM_EINVAL;
    if (B && L && (!x || !weight || !bias || !y)) return WM_ENULL;
    if (C == 64) {                                       // SS2D.out_norm on (B, D, L) planes (NCHW training path)
        const long long total = (long long)B * L;
        if (total == 0) return WM_OK;
        hipStream_t st = (hipStream_t)stream;
        ProfScope ps(16, st);
        hipLaunchKernelGGL((layernorm2d_kernel<64>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, weight, bias,
                           eps, y, B, (long long)L);
        return launch_status();
    }
    WM_LFSS_DISPATCH(16, layernorm2d_kernel, x, weight, bias, eps, y, B, (long long)L);
}

}  // extern "C"
namespace wm {
// waves / blocks / slice of a Gram launch
static void gram_plan(int64_t L, long long& nblk, long long& slice) {
    // >= 512 positions per wave on large maps; small maps are latency-bound (one round trip per 32 positions of a
    // wave), so they get down to 128 positions per wave, up to 2048 waves
    long long waves = (L + 511) / 512, wsmall = (L + 127) / 128;
    if (wsmall > 2048) wsmall = 2048;
    if (waves < wsmall) waves = wsmall;
    if (waves > 4096) waves = 4096;
    if (waves < 1) waves = 1;
    waves = ((waves + kGramWaves - 1) / kGramWaves) * kGramWaves;
    slice = (L + waves - 1) / waves;
    slice = ((slice + 31) / 32) * 32;
    if (slice < 32) slice = 32;
    nblk = waves / kGramWaves;
}
}  // namespace wm
extern "C" {

// Zero-initialised accumulator outputs (parameter gradients that kernels add into with atomics, the two running maxima of
// wm_conv2d_f16_steps).  A caller that hands out such buffers from memory it has ALREADY zeroed registers that memory
// (wm_zero_arena_register): a buffer that lies inside a registered range is taken as zero and the memset node is skipped - a
// BASELINE config-3 training step issued 323 memsets of a few hundred bytes, 4.2 us of stream time each (round 5).
struct ZeroArenas {
    std::mutex mu;
    std::atomic<int> count{0};
    std::vector<std::pair<uintptr_t, uintptr_t>> ranges;            // [begin, end)
};
static ZeroArenas g_zero_arenas;

static bool prezeroed(const void* p, size_t bytes) {
    if (g_zero_arenas.count.load(std::memory_order_acquire) == 0) return false;
    const uintptr_t b = reinterpret_cast<uintptr_t>(p), e = b + bytes;
    std::lock_guard<std::mutex> lk(g_zero_arenas.mu);
    for (const auto& r : g_zero_arenas.ranges)
        if (b >= r.first && e <= r.second) return true;
    return false;
}

static hipError_t zero_out(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0 || prezeroed(p, bytes)) return hipSuccess;
    return zero_async(p, bytes, st);
}

// Zero two small gradient buffers: ONE memset node when the caller allocated them back to back (ops.py does: a training step
// issued 440 memsets of a few hundred bytes, ~4 us of GPU time each).
static hipError_t zero_pair(float* a, size_t na, float* b, size_t nb, hipStream_t st) {
    if (b && b == a + na) return zero_out(a, (na + nb) * sizeof(float), st);
    hipError_t e = zero_out(a, na * sizeof(float), st);
    if (e == hipSuccess && b) e = zero_out(b, nb * sizeof(float), st);
    return e;
}

int wm_zero_arena_register(void* base, size_t bytes) {
    if (!base) return WM_ENULL;
    if (bytes == 0) return WM_EINVAL;
    const uintptr_t b = reinterpret_cast<uintptr_t>(base);
    std::lock_guard<std::mutex> lk(g_zero_arenas.mu);
    for (const auto& r : g_zero_arenas.ranges)
        if (b < r.second && b + bytes > r.first) return WM_EINVAL;  // overlaps a registered range
    g_zero_arenas.ranges.emplace_back(b, b + bytes);
    g_zero_arenas.count.store((int)g_zero_arenas.ranges.size(), std::memory_order_release);
    return WM_OK;
}

int wm_zero_arena_unregister(void* base) {
    const uintptr_t b = reinterpret_cast<uintptr_t>(base);
    std::lock_guard<std::mutex> lk(g_zero_arenas.mu);
    for (size_t i = 0; i < g_zero_arenas.ranges.size(); ++i)
        if (g_zero_arenas.ranges[i].first == b) {
            g_zero_arenas.ranges.erase(g_zero_arenas.ranges.begin() + (long)i);
            g_zero_arenas.count.store((int)g_zero_arenas.ranges.size(), std::memory_order_release);
            return WM_OK;
        }
    return WM_EINVAL;
}

size_t wm_gram_workspace_bytes(int B, int C, int64_t L) {
    if (B <= 0 || C <= 0 || C > 32 || L < 0) return 0;
    long long nblk, slice;
    gram_plan(L, nblk, slice);
    return (size_t)B * nblk * kGramPart * sizeof(float);
}

int wm_gram_fwd(const float* X, const float* Y, float* G, float* nx, float* ny, void* workspace, size_t workspace_bytes,
                int B, int C, int64_t L, void* stream) {
    if (B < 0 || C < 0 || L < 0) return WM_EINVAL;
    if (C > 32) return WM_EUNSUPPORTED;
    if (B == 0 || C == 0) return WM_OK;
    if (!X || !Y || !G || !nx || !ny || !workspace) return WM_ENULL;
    if (!aligned16(X) || !aligned16(Y) || !aligned16(workspace)) return WM_EALIGN;
    long long nblk, slice;
    gram_plan(L, nblk, slice);
    if (workspace_bytes < (size_t)B * nblk * kGramPart * sizeof(float)) return WM_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    if (L % 4 == 0)
        hipLaunchKernelGGL(gram32_kernel<true>, dim3((unsigned)nblk, (unsigned)B), dim3(64 * kGramWaves), 0, st, X, Y, part, C,
                           (long long)L, slice);
    else
        hipLaunchKernelGGL(gram32_kernel<false>, dim3((unsigned)nblk, (unsigned)B), dim3(64 * kGramWaves), 0, st, X, Y, part, C,
                           (long long)L, slice);
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(kGramPart / 64, (unsigned)B), dim3(256), 0, st, (const float*)part, G, nx, ny,
                       C, (int)nblk);
    return launch_status();
}


int wm_dwconv3x3_wgrad(const float* x, const float* gy, float* dW, float* db, int B, int C, int H, int W,
                       void* stream) {
    if (B < 0 || C < 0 || H < 0 || W < 0) return WM_EINVAL;
    if (C == 0) return WM_OK;
    if (!dW) return WM_ENULL;
    hipStream_t st = (hipStream_t)stream;
    const hipError_t e = zero_pair(dW, (size_t)C * 9, db, (size_t)C, st);
    if (e != hipSuccess) return (int)e;
    const long long planes = (long long)B * C;
    if (planes == 0 || H == 0 || W == 0) return WM_OK;
    if (!x || !gy) return WM_ENULL;
    const bool vec = (W % 4 == 0) && aligned16(x) && aligned16(gy);
    const int lpr = dw_lanes_per_row(W, vec);
    const long long pgroups = (planes + 64 / lpr - 1) / (64 / lpr);
    const dim3 block(64, 4);
    const int rows = kDwRows;                  // (shorter strips on small maps: more atomics - twice the time at config 3's level 3)
    const dim3 grid((unsigned)((W + 4 * lpr - 1) / (4 * lpr)), (unsigned)((H + 4 * rows - 1) / (4 * rows)),
                    (unsigned)(pgroups < 65535 ? pgroups : 65535));
    if (!vec) hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<false, 64>), grid, block, 0, st, x, gy, dW, db, C, H, W, planes, rows);
    else if (lpr == 64) hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<true, 64>), grid, block, 0, st, x, gy, dW, db, C, H, W, planes, rows);
    else if (lpr == 32) hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<true, 32>), grid, block, 0, st, x, gy, dW, db, C, H, W, planes, rows);
    else hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<true, 16>), grid, block, 0, st, x, gy, dW, db, C, H, W, planes, rows);
    return launch_status();
}

int wm_layernorm2d_bwd(const float* x, const float* weight, const float* gy, float eps, float* gx, float* dweight,
                       float* dbias, int B, int64_t L, int C, void* stream) {
    if (B < 0 || L < 0) return WM_EINVAL;
    if (C != 8 && C != 16 && C != 32 && C != 64) return WM_EUNSUPPORTED;
    if (!dweight || !dbias) return WM_ENULL;
    hipStream_t st = (hipStream_t)stream;
    const hipError_t e = zero_pair(dweight, (size_t)C, dbias, (size_t)C, st);
    if (e != hipSuccess) return (int)e;
    const long long total = (long long)B * L;
    if (total == 0) return WM_OK;
    if (!x || !weight || !gy || !gx) return WM_ENULL;
    const int tpp = C >= 32 ? 2 : 1;                     // threads per pixel (layernorm2d_bwd_pair_kernel)
    long long blocks = (total * tpp + 255) / 256;
    if (blocks > 512) blocks = 512;                      // grid-stride: few blocks -> few atomics per channel
    const dim3 grid((unsigned)blocks), block(256);
    if (C == 64) hipLaunchKernelGGL((layernorm2d_bwd_pair_kernel<64>), grid, block, 0, st, x, weight, gy, eps, gx, dweight, dbias, B, (long long)L);
    else if (C == 32) hipLaunchKernelGGL((layernorm2d_bwd_pair_kernel<32>), grid, block, 0, st, x, weight, gy, eps, gx, dweight, dbias, B, (long long)L);
    else if (C == 16) hipLaunchKernelGGL((layernorm2d_bwd_kernel<16>), grid, block, 0, st, x, weight, gy, eps, gx, dweight, dbias, B, (long long)L);
    else hipLaunchKernelGGL((layernorm2d_bwd_kernel<8>), grid, block, 0, st, x, weight, gy, eps, gx, dweight, dbias, B, (long long)L);
    return launch_status();
}

int wm_layernorm_tok_fwd(const float* x, const float* weight, const float* bias, float eps, float* y, int64_t T, int C,
                         void* stream) {
    if (T < 0) return WM_EINVAL;
    if (C != 8 && C != 16 && C != 32 && C != 64) return WM_EUNSUPPORTED;
    if (T == 0) return WM_OK;
    if (!x || !weight || !bias || !y) return WM_ENULL;
    if (!aligned16(x) || !aligned16(y) || !aligned16(weight) || !aligned16(bias)) return WM_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int tpb = 256 / (C / 4);
    long long blocks = (T + tpb - 1) / tpb;
    if (blocks > 256 * 32) blocks = 256 * 32;
    const dim3 grid((unsigned)blocks), block(256);
#define WM_LNT(CC) hipLaunchKernelGGL((layernorm_tok_kernel<CC>), grid, block, 0, st, (const float4*)x, (const float4*)weight, \
                                      (const float4*)bias, eps, (float4*)y, (long long)T)
    if (C == 64) WM_LNT(64); else if (C == 32) WM_LNT(32); else if (C == 16) WM_LNT(16); else WM_LNT(8);
#undef WM_LNT
    return launch_status();
}

int wm_layernorm_tok_bwd(const float* x, const float* weight, const float* gy, float eps, float* gx, float* dweight,
                         float* dbias, int64_t T, int C, void* stream) {
    if (T < 0) return WM_EINVAL;
    if (C != 8 && C != 16 && C != 32 && C != 64) return WM_EUNSUPPORTED;
    if (!dweight || !dbias) return WM_ENULL;
    hipStream_t st = (hipStream_t)stream;
    const hipError_t e = zero_pair(dweight, (size_t)C, dbias, (size_t)C, st);
    if (e != hipSuccess) return (int)e;
    if (T == 0) return WM_OK;
    if (!x || !weight || !gy || !gx) return WM_ENULL;
    if (!aligned16(x) || !aligned16(gy) || !aligned16(gx) || !aligned16(weight)) return WM_EALIGN;
    const int tpb = 256 / (C / 4);
    long long blocks = (T + tpb - 1) / tpb;
    if (blocks > 1024) blocks = 1024;                    // grid-stride: few blocks -> few atomics per channel
    const dim3 grid((unsigned)blocks), block(256);
#define WM_LNTB(CC) hipLaunchKernelGGL((layernorm_tok_bwd_kernel<CC>), grid, block, 0, st, (const float4*)x, (const float4*)weight, \
                                       (const float4*)gy, eps, (float4*)gx, dweight, dbias, (long long)T)
    if (C == 64) WM_LNTB(64); else if (C == 32) WM_LNTB(32); else if (C == 16) WM_LNTB(16); else WM_LNTB(8);
#undef WM_LNTB
    return launch_status();
}

int wm_image_pre_u8(const uint8_t* image, float* out, int h, int w, int Hp, int Wp, int swap_rb, void* stream) {
    if (h < 0 || w < 0 || Hp < h || Wp < w) return WM_EINVAL;
    if (Hp == 0 || Wp == 0) return WM_OK;
    if (h == 0 || w == 0) return WM_EINVAL;
    if (Hp - h > h - 1 || Wp - w > w - 1) return WM_EINVAL;            // reflect padding needs pad < size
    if (!image || !out) return WM_ENULL;
    if (Hp > 65535) return WM_EUNSUPPORTED;
    hipLaunchKernelGGL(image_pre_kernel, dim3((unsigned)((Wp + 255) / 256), (unsigned)Hp), dim3(256), 0, (hipStream_t)stream,
                       image, out, h, w, Hp, Wp, swap_rb);
    return launch_status();
}

int wm_image_post_u8(const float* in, uint8_t* image, int h, int w, int Hp, int Wp, int swap_rb, void* stream) {
    if (h < 0 || w < 0 || Hp < h || Wp < w) return WM_EINVAL;
    if (h == 0 || w == 0) return WM_OK;
    if (!in || !image) return WM_ENULL;
    if (h > 65535) return WM_EUNSUPPORTED;
    hipLaunchKernelGGL(image_post_kernel, dim3((unsigned)((w + 255) / 256), (unsigned)h), dim3(256), 0, (hipStream_t)stream,
                       in, image, h, w, Hp, Wp, swap_rb);
    return launch_status();
}

}  // extern "C"
template <int OT, int IT>
static void linear_wgrad_launch(const float* gy, const float* x, float* dW, long long T, hipStream_t st) {
    long long waves = (T + 511) / 512;                                  // >= 512 tokens per wave
    if (waves > 4096) waves = 4096;
    waves = ((waves + wm::kLwWaves - 1) / wm::kLwWaves) * wm::kLwWaves;
    long long slice = (T + waves - 1) / waves;
    slice = ((slice + 3) / 4) * 4;
    hipLaunchKernelGGL((wm::linear_wgrad_kernel<OT, IT>), dim3((unsigned)(waves / wm::kLwWaves)), dim3(64 * wm::kLwWaves), 0,
                       st, gy, x, dW, T, slice);
}
extern "C" {

int wm_linear_wgrad(const float* gy, const float* x, float* dW, int64_t T, int O, int I, void* stream) {
    if (T < 0 || O <= 0 || I <= 0) return WM_EINVAL;
    if (!dW) return WM_ENULL;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = zero_out(dW, (size_t)O * I * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    if (T == 0) return WM_OK;
    if (!gy || !x) return WM_ENULL;
    if (O % 16 != 0 || I % 16 != 0 || O * I > 8192) return WM_EUNSUPPORTED;
#define WM_LW(OT, IT) if (O == 16 * OT && I == 16 * IT) { linear_wgrad_launch<OT, IT>(gy, x, dW, (long long)T, st); return launch_status(); }
    WM_LW(8, 2) WM_LW(2, 4) WM_LW(4, 1) WM_LW(1, 2) WM_LW(2, 1) WM_LW(1, 1) WM_LW(4, 2) WM_LW(2, 2) WM_LW(1, 4)
#undef WM_LW
    return WM_EUNSUPPORTED;
}

// ---- dense convolution weight gradient (conv_wgrad.hip.h) -----------------------------------------------------------
// position sub-ranges (= partials per input tile) of the weight-gradient launch; *blocks = workgroups along x
static int conv_wgrad_parts(long long nunits, int ntiles_in, int* upw, int* blocks) {
    // About one 4-wave workgroup per compute unit: a wave's fixed cost - its OT x TAPS KB partial and the finish kernel's
    // pass over it - is what more of them buy (tools/bench_conv_wgrad.py).  A workgroup holds tpw input tiles x gpw
    // position sub-ranges.
#ifndef WM_CW_TARGET
#define WM_CW_TARGET 256
#endif
#ifndef WM_CW_MINUNITS
#define WM_CW_MINUNITS 4
#endif
    const int tpw = cw_tiles_per_wg(ntiles_in), gpw = kCwWaves / tpw;
    const int ygroups = (ntiles_in + tpw - 1) / tpw;
    long long wgs = WM_CW_TARGET / ygroups;
    if (wgs < 1) wgs = 1;
    long long parts = wgs * gpw;
    const long long most = nunits / WM_CW_MINUNITS;
    if (parts > most) parts = most;
    if (parts < 1) parts = 1;
    const long long per = (nunits + parts - 1) / parts;
    *upw = (int)per;
    parts = (nunits + per - 1) / per;
    *blocks = (int)((parts + gpw - 1) / gpw);
    return (int)parts;
}
size_t wm_conv2d_wgrad_workspace_bytes(int B, int Cin, int Cout, int H, int W, int ks) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (ks != 1 && ks != 3) || W % 32 != 0) return 0;
    const int OT = (Cout + 15) / 16;
    if (OT > 6 || (OT != 1 && OT != 2 && OT != 4 && OT != 6)) return 0;
    int upw, blocks;
    const int np = conv_wgrad_parts((long long)B * H * (W / 32), (Cin + 15) / 16, &upw, &blocks);
    return ((size_t)((Cin + 15) / 16) * np * OT * ks * ks * 256 + (size_t)np * kCwBiasRow) * sizeof(float);
}
int wm_conv2d_wgrad(const float* gy, const float* x, float* dW, float* db, void* workspace, size_t workspace_bytes, int B, int Cin,
                    int Cout, int H, int W, int ks, void* stream) {
    if (B < 0 || Cin <= 0 || Cout <= 0 || H < 0 || W < 0) return WM_EINVAL;
    if (ks != 1 && ks != 3) return WM_EUNSUPPORTED;
    if (!dW) return WM_ENULL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0 || H == 0 || W == 0) {
        hipError_t e = zero_async(dW, (size_t)Cout * Cin * ks * ks * sizeof(float), st);
        if (e == hipSuccess && db) e = zero_async(db, (size_t)Cout * sizeof(float), st);
        return e == hipSuccess ? WM_OK : (int)e;
    }
    const size_t need = wm_conv2d_wgrad_workspace_bytes(B, Cin, Cout, H, W, ks);
    if (need == 0) return WM_EUNSUPPORTED;                        // W % 32 != 0, more than 96 output channels, ...
    if ((long long)B * (Cin > Cout ? Cin : Cout) * H * W > 0x7fffffffffLL || (long long)B * H * (W / 32) > 0x7fffff00LL) return WM_EUNSUPPORTED;
    if (!gy || !x || !workspace) return WM_ENULL;
    if (workspace_bytes < need) return WM_EWORKSPACE;
    if (!aligned16(gy) || !aligned16(x) || !aligned16(workspace)) return WM_EALIGN;
    ConvWgradArgs a;
    a.gy = gy; a.x = x; a.part = (float*)workspace; a.dW = dW; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.nunits = (long long)B * H * (W / 32);
    int blocks = 1;
    a.nparts = conv_wgrad_parts(a.nunits, (Cin + 15) / 16, &a.upw, &blocks);
    const int OT = (Cout + 15) / 16, ITN = (Cin + 15) / 16;
    a.bpart = (float*)workspace + (size_t)ITN * a.nparts * OT * ks * ks * 256;
    a.db = db;
    const int ygroups = (ITN + cw_tiles_per_wg(ITN) - 1) / cw_tiles_per_wg(ITN);
#define WM_CW(KS, OTV, CO0, NCO)                                                                                         \
    do {                                                                                                                 \
        a.co0 = (CO0); a.nco = (NCO);                                                                                    \
        hipLaunchKernelGGL((conv_wgrad_kernel<KS, OTV>), dim3((unsigned)blocks, (unsigned)ygroups), dim3(64 * kCwWaves), 0, st, a); \
        hipLaunchKernelGGL((conv_wgrad_finish_kernel<KS, OTV>), dim3((unsigned)(OTV * KS * KS * 16), (unsigned)ITN), dim3(256), 0, st, a); \
    } while (0)
    if (ks == 3) {
        if (OT == 1) WM_CW(3, 1, 0, Cout); else if (OT == 2) WM_CW(3, 2, 0, Cout); else if (OT == 4) WM_CW(3, 4, 0, Cout);
        else { WM_CW(3, 4, 0, 64); WM_CW(3, 2, 64, Cout - 64); }     // 65 .. 96 output channels: two passes (same workspace, stream order)
    } else {
        if (OT == 1) WM_CW(1, 1, 0, Cout); else if (OT == 2) WM_CW(1, 2, 0, Cout); else if (OT == 4) WM_CW(1, 4, 0, Cout); else WM_CW(1, 6, 0, Cout);
    }
#undef WM_CW
    return launch_status();
}

int wm_plane_sums(const float* x, float* sums, int B, int C, int H, int W, void* stream) {
    if (B < 0 || C < 0 || H < 0 || W < 0) return WM_EINVAL;
    if (C == 0) return WM_OK;
    if (!sums) return WM_ENULL;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = zero_out(sums, (size_t)C * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    const long long HW = (long long)H * W, planes = (long long)B * C;
    if (planes == 0 || HW == 0) return WM_OK;
    if (!x) return WM_ENULL;
    if (planes > 65535) return WM_EUNSUPPORTED;
    const bool vec = (HW % 4 == 0) && aligned16(x);
    long long bpp = (HW / 4 + 256 * 8 - 1) / (256 * 8);
    const long long cap = (256 * 16 + planes - 1) / planes;
    if (bpp > cap) bpp = cap;
    if (bpp < 1) bpp = 1;
    hipLaunchKernelGGL(plane_sums_kernel, dim3((unsigned)bpp, (unsigned)planes), dim3(256), 0, st, x, sums, C, HW, vec);
    return launch_status();
}

// mean |a - b| over n elements -> out[0] (zeroed here, by a kernel); ga = gout[0] * sign(a - b) / n
int wm_l1_mean_fwd(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (n < 0) return WM_EINVAL;
    if (!out) return WM_ENULL;
    hipStream_t st = (hipStream_t)stream;
    const hipError_t e = zero_out(out, sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return WM_OK;                                     // (torch returns nan for an empty mean; callers never ask)
    if (!a || !b) return WM_ENULL;
    const int vec = (n % 4 == 0) && aligned16(a) && aligned16(b) ? 1 : 0;
    long long blocks = ((vec ? n / 4 : n) + 256 * 8 - 1) / (256 * 8);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(l1_mean_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, out, (long long)n, 1.0f / (float)n, vec);
    return launch_status();
}
int wm_l1_mean_bwd(const float* a, const float* b, const float* gout, float* ga, int64_t n, void* stream) {
    if (n < 0) return WM_EINVAL;
    if (n == 0) return WM_OK;
    if (!a || !b || !gout || !ga) return WM_ENULL;
    const int vec = (n % 4 == 0) && aligned16(a) && aligned16(b) && aligned16(ga) ? 1 : 0;
    long long blocks = ((vec ? n / 4 : n) + 256 * 4 - 1) / (256 * 4);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(l1_mean_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, gout, ga, (long long)n,
                       1.0f / (float)n, vec);
    return launch_status();
}

// act: 1 = SiLU, 2 = GELU (erf).  a / b / out (and g / ga / gb) are (B, per_b) with batch strides in elements.
int wm_gate_fwd(const float* a, const float* b, float* out, int act, int B, int64_t per_b, int64_t stride_a, int64_t stride_b,
                int64_t stride_out, void* stream) {
    if (B < 0 || per_b < 0) return WM_EINVAL;
    if (act < 1 || act > 3) return WM_EUNSUPPORTED;
    if (B == 0 || per_b == 0) return WM_OK;
    if (!a || !b || !out) return WM_ENULL;
    if (B > 65535) return WM_EUNSUPPORTED;
    GateArgs p{a, b, nullptr, out, nullptr, nullptr, per_b, stride_a, stride_b, 0, stride_out, 0, 0};
    const bool vec = per_b % 4 == 0 && stride_a % 4 == 0 && stride_b % 4 == 0 && stride_out % 4 == 0 && aligned16(a) &&
                     aligned16(b) && aligned16(out);
    const dim3 grid((unsigned)((per_b + 1023) / 1024), (unsigned)B), block(256);
    hipStream_t st = (hipStream_t)stream;
#define WM_GATE(ACT) do { if (vec) hipLaunchKernelGGL((gate_kernel<ACT, false, true>), grid, block, 0, st, p);          \
                          else hipLaunchKernelGGL((gate_kernel<ACT, false, false>), grid, block, 0, st, p); } while (0)
    if (act == 1) WM_GATE(1); else if (act == 2) WM_GATE(2); else WM_GATE(3);
#undef WM_GATE
    return launch_status();
}

int wm_gate_bwd(const float* a, const float* b, const float* g, float* ga, float* gb, int act, int B, int64_t per_b,
                int64_t stride_a, int64_t stride_b, int64_t stride_g, int64_t stride_ga, int64_t stride_gb, void* stream) {
    if (B < 0 || per_b < 0) return WM_EINVAL;
    if (act < 1 || act > 3) return WM_EUNSUPPORTED;
    if (B == 0 || per_b == 0) return WM_OK;
    if (!a || !b || !g || !ga || !gb) return WM_ENULL;
    if (B > 65535) return WM_EUNSUPPORTED;
    GateArgs p{a, b, g, nullptr, ga, gb, per_b, stride_a, stride_b, stride_g, 0, stride_ga, stride_gb};
    const bool vec = per_b % 4 == 0 && stride_a % 4 == 0 && stride_b % 4 == 0 && stride_g % 4 == 0 && stride_ga % 4 == 0 &&
                     stride_gb % 4 == 0 && aligned16(a) && aligned16(b) && aligned16(g) && aligned16(ga) && aligned16(gb);
    const dim3 grid((unsigned)((per_b + 1023) / 1024), (unsigned)B), block(256);
    hipStream_t st = (hipStream_t)stream;
#define WM_GATE(ACT) do { if (vec) hipLaunchKernelGGL((gate_kernel<ACT, true, true>), grid, block, 0, st, p);           \
                          else hipLaunchKernelGGL((gate_kernel<ACT, true, false>), grid, block, 0, st, p); } while (0)
    if (act == 1) WM_GATE(1); else if (act == 2) WM_GATE(2); else WM_GATE(3);
#undef WM_GATE
    return launch_status();
}

int wm_scale_add_fwd(const float* x, const float* scale, const float* o, float* out, int B, int C, int64_t L, void* stream) {
    if (B < 0 || C < 0 || L < 0) return WM_EINVAL;
    if (B == 0 || C == 0 || L == 0) return WM_OK;
    if (!x || !scale || !o || !out) return WM_ENULL;
    if ((long long)B * C > 65535) return WM_EUNSUPPORTED;
    const bool vec = L % 4 == 0 && aligned16(x) && aligned16(o) && aligned16(out);
    const dim3 grid((unsigned)((L + 1023) / 1024), (unsigned)(B * C)), block(256);
    if (vec) hipLaunchKernelGGL(scale_add_fwd_kernel<true>, grid, block, 0, (hipStream_t)stream, x, scale, o, out, C, (long long)L);
    else hipLaunchKernelGGL(scale_add_fwd_kernel<false>, grid, block, 0, (hipStream_t)stream, x, scale, o, out, C, (long long)L);
    return launch_status();
}

int wm_scale_add_bwd(const float* g, const float* x, const float* scale, float* gx, float* gscale, int B, int C, int64_t L,
                     void* stream) {
    if (B < 0 || C < 0 || L < 0) return WM_EINVAL;
    if (C == 0) return WM_OK;
    if (!gscale) return WM_ENULL;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = zero_out(gscale, (size_t)C * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    if (B == 0 || L == 0) return WM_OK;
    if (!g || !x || !scale || !gx) return WM_ENULL;
    if ((long long)B * C > 65535) return WM_EUNSUPPORTED;
    const bool vec = L % 4 == 0 && aligned16(g) && aligned16(x) && aligned16(gx);
    long long bpp = (L + 1023) / 1024;                       // blocks per plane: <= 8 (scale_add_bwd_kernel), >= ~2048 in all if the map allows
    const long long want = (2048 + (long long)B * C - 1) / ((long long)B * C);
    const long long cap = want > 8 ? want : 8;
    if (bpp > cap) bpp = cap;
    const dim3 grid((unsigned)bpp, (unsigned)(B * C)), block(256);
    if (vec) hipLaunchKernelGGL(scale_add_bwd_kernel<true>, grid, block, 0, st, g, x, scale, gx, gscale, C, (long long)L);
    else hipLaunchKernelGGL(scale_add_bwd_kernel<false>, grid, block, 0, st, g, x, scale, gx, gscale, C, (long long)L);
    return launch_status();
}

int wm_match_index(const float* G, const float* nx, const float* ny, int* index, int B, int C, void* stream) {
    if (B < 0 || C < 0) return WM_EINVAL;
    if (B == 0 || C == 0) return WM_OK;
    if (!G || !nx || !ny || !index) return WM_ENULL;
    hipLaunchKernelGGL(match_argmin_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, G, nx, ny, index, C);
    return launch_status();
}

int wm_attn_fold(const float* G, const float* nq, const float* nk, const float* temperature, const float* Wpo,
                 float* Wout, int B, int C, int heads, void* stream) {
    if (B < 0 || C <= 0 || heads <= 0 || C % heads != 0) return WM_EINVAL;
    if (C > 64) return WM_EUNSUPPORTED;
    if (B == 0) return WM_OK;
    if (!G || !nq || !nk || !temperature || !Wpo || !Wout) return WM_ENULL;
    hipLaunchKernelGGL(attn_fold_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, G, nq, nk, temperature,
                       Wpo, Wout, C, heads);
    return launch_status();
}

}  // extern "C"
// blocks per plane of the SKFF reduction / apply kernels: ~16 blocks per compute unit in flight over all planes
static long long skff_bpp_cap(long long planes) { return (256 * 16 + planes - 1) / planes; }
extern "C" {

size_t wm_skff_workspace_bytes(int B, int C) {
    if (B <= 0 || C <= 0) return 0;
    const long long planes = (long long)B * C;
    return (size_t)(planes * skff_bpp_cap(planes) + 3 * planes) * sizeof(float);
}

int wm_skff_fwd(const float* x0, const float* x1, const float* x2, const float* Wdu, const float* prelu,
                const float* Wfc, float* out, void* workspace, size_t workspace_bytes, int B, int C, int d, int H, int W,
                void* stream) {
    if (B < 0 || C <= 0 || d <= 0 || H < 0 || W < 0) return WM_EINVAL;
    if (C > 64 || d > 16) return WM_EUNSUPPORTED;
    if (B == 0 || H == 0 || W == 0) return WM_OK;
    if (!x0 || !x1 || !x2 || !Wdu || !prelu || !Wfc || !out || !workspace) return WM_ENULL;
    if (workspace_bytes < wm_skff_workspace_bytes(B, C)) return WM_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long long HW = (long long)H * W, planes = (long long)B * C;
    if (planes > 65535) return WM_EUNSUPPORTED;
    const bool vec = (HW % 4 == 0) && aligned16(x0) && aligned16(x1) && aligned16(x2) && aligned16(out);
    long long bpp = (HW / 4 + 256 * 8 - 1) / (256 * 8);              // >= 8 float4 per thread
    const long long cap = skff_bpp_cap(planes);
    if (bpp > cap) bpp = cap;
    if (bpp < 1) bpp = 1;
    float* wts = (float*)workspace;                      // (B, 3, C)
    float* part = wts + (size_t)3 * planes;              // (B, C, bpp) block partials of the plane sums
    const dim3 grid((unsigned)bpp, (unsigned)planes), block(256);
    ProfScope ps(15, st);
    hipLaunchKernelGGL(chansum3_kernel, grid, block, 0, st, x0, x1, x2, part, HW, vec);
    hipLaunchKernelGGL(skff_weights_kernel, dim3((unsigned)B), dim3(64), 0, st, (const float*)part, (int)bpp, Wdu, prelu, Wfc,
                       wts, C, d, (float)(1.0 / (double)HW));
    hipLaunchKernelGGL(skff_apply_kernel, grid, block, 0, st, x0, x1, x2, wts, out, C, HW, vec);
    return launch_status();
}

size_t wm_conv2d_wfrag_bytes(int Cout, int Cin, int ks) {
    if (Cout <= 0 || Cin <= 0 || (ks != 1 && ks != 3)) return 0;
    return (size_t)((Cin + 15) / 16) * ks * ks * ((Cout + 31) / 32) * 2 * 64 * 16;
}

int wm_conv2d_prep(const float* weight, void* wfrag, int Cout, int Cin, int ks, void* stream) {
    if (Cout <= 0 || Cin <= 0) return WM_EINVAL;
    if (ks != 1 && ks != 3) return WM_EUNSUPPORTED;
    if (!weight || !wfrag) return WM_ENULL;
    if (!aligned16(wfrag)) return WM_EALIGN;
    const int nch = (Cin + 15) / 16, mtot = (Cout + 31) / 32;
    const long long total = (long long)nch * ks * ks * mtot * 128;
    hipLaunchKernelGGL(conv2d_prep_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       weight, (uint4*)wfrag, Cout, Cin, ks * ks, nch, mtot, (const float*)nullptr);
    return launch_status();
}

}  // extern "C"

template <int KS, int RW, int MT, bool G1X1 = false, bool F16 = false, bool LNIN = false>
static int conv2d_launch(const wm::Conv2dArgs& a, int B, hipStream_t st) {
    constexpr int PAD = KS / 2;
    constexpr int smem = ((4 * RW + 2 * PAD) * (wm::kCvTW + 2 * PAD) * 4 + (KS * KS + (G1X1 ? 1 : 0)) * MT * 2 * 64) * 16;   // input planes + weights
    static bool configured[64] = {};
    if (smem > 65536) {
        const int rc = wm::lds_optin((const void*)wm::conv2d_mfma_kernel<KS, RW, MT, G1X1, F16, LNIN>, smem, configured);
        if (rc) return rc;
    }
    const int ntiles = ((a.W + wm::kCvTW - 1) / wm::kCvTW) * ((a.H + 4 * RW - 1) / (4 * RW));
    const dim3 grid((unsigned)(((ntiles + 7) / 8) * 8), (unsigned)B);
    hipLaunchKernelGGL((wm::conv2d_mfma_kernel<KS, RW, MT, G1X1, F16, LNIN>), grid, dim3(256), smem, st, a);
    return launch_status();
}

// Which 3x3 kernel: the persistent wave-specialised one (conv2d_ws.hip.h, one workgroup per compute unit) where it
// pays - enough 64 x 8 tiles that every compute unit pipelines a few (UHD levels 1 and 2 and full resolution; at level 3
// a workgroup gets one or two tiles and the first-generation kernel is 20-30 % faster), at most one epilogue operand and
// then a single 32-channel row tile (two launches re-reading the input lose to the first-generation kernel's one) - and
// where its 32-bit offsets hold.  wm_conv2d_select() pins the choice (parity tests run both on the same inputs: the
// accumulation order per output element is the same, so the results are bit-identical).
// rows per consumer wave (tile = 64 x 2 RW pixels) of the wave-specialised launches: two row tiles, one row tile, gated
#ifndef WM_CONV_WS_RW2
#define WM_CONV_WS_RW2 4
#endif
#ifndef WM_CONV_WS_RW1
#define WM_CONV_WS_RW1 4
#endif
#ifndef WM_CONV_WS_RWG
#define WM_CONV_WS_RWG 2
#endif
#ifndef WM_CONV_WS_NPW1
#define WM_CONV_WS_NPW1 4              // producer waves of the one-row-tile launches
#endif
static std::atomic<int> g_conv_select{0};
static int conv_select_mode() { return g_conv_select.load(std::memory_order_relaxed); }
// th: tile rows of the launch that would run (2 x row tiles per workgroup)
static bool conv_ws_enabled(const wm::Conv2dArgs& a, int B, int th) {
    const int mode = conv_select_mode();
    if (mode == 1) return false;
    // 32-bit byte offsets inside one batch element of every tensor; gather indices in two registers
    const long long cmax = std::max(std::max(a.Ca, a.xb ? a.Cbsrc : 0), a.Cout);
    if (cmax * a.H * a.W * 4 >= (1ll << 32) || (a.xb_idx && a.Cb > 128)) return false;
    if (a.gate && a.res) return false;
    if (mode == 2) return true;
    if ((a.gate || a.res) && a.mtot > 1) return false;
    const long long ntiles = (long long)B * ((a.W + wm::kWsTW - 1) / wm::kWsTW) * ((a.H + th - 1) / th);
    return ntiles >= 768;
}

template <int RW, int MT, bool G1X1 = false, bool EPI = false, int NPW = 4, bool F16 = false>
static int conv2d_ws_launch(const wm::Conv2dArgs& a, int B, hipStream_t st) {
    using Cfg = wm::ConvWsCfg<RW, MT, G1X1, NPW>;
    static bool configured[64] = {};
    static int ncu[64] = {};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return WM_EHIP;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!configured[dev]) {
            if (hipFuncSetAttribute((const void*)wm::conv3x3_ws_kernel<RW, MT, G1X1, EPI, NPW, F16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    Cfg::LDS_BYTES) != hipSuccess) return WM_EHIP;
            if (hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return WM_EHIP;
            configured[dev] = true;
        }
    }
    const long long ntiles = (long long)B * ((a.W + wm::kWsTW - 1) / wm::kWsTW) * ((a.H + Cfg::TH - 1) / Cfg::TH);
    if (ntiles >= (1ll << 31)) return WM_EUNSUPPORTED;
    const int cus = std::max(8, ncu[dev] & ~7);
    const int G = (int)std::min<long long>(cus, ((ntiles + 7) / 8) * 8);
    hipLaunchKernelGGL((wm::conv3x3_ws_kernel<RW, MT, G1X1, EPI, NPW, F16>), dim3((unsigned)G), dim3(256 + 64 * NPW), Cfg::LDS_BYTES, st, a, B);
    return launch_status();
}

extern "C" {

int wm_conv2d_fwd(const float* xa, const float* xb, const int* xb_index, const void* wfrag, const float* bias,
                  const float* gate, const float* residual, float* y, int B, int Ca, int Cb, int Cb_src, int Cout,
                  int H, int W, int ks, void* stream) {
    if (B < 0 || Ca <= 0 || Cb < 0 || Cout <= 0 || H < 0 || W < 0 || Cb_src < 0) return WM_EINVAL;
    if (ks != 1 && ks != 3) return WM_EUNSUPPORTED;
    if (B == 0 || H == 0 || W == 0) return WM_OK;
    if (!xa || !wfrag || !y || (Cb > 0 && !xb)) return WM_ENULL;
    if (Cb > 0 && Ca % 8 != 0) return WM_EUNSUPPORTED;   // an 8-channel fragment never straddles the two sources
    if (Cb > 0 && !xb_index && Cb_src != Cb) return WM_EINVAL;
    if (B > 65535 || (long long)H * W >= (1ll << 31)) return WM_EUNSUPPORTED;
    if (!aligned16(wfrag)) return WM_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    Conv2dArgs a;
    a.xa = xa; a.xb = Cb > 0 ? xb : nullptr; a.xb_idx = Cb > 0 ? xb_index : nullptr; a.wfrag = (const uint4*)wfrag;
    a.bias = bias; a.gate = gate; a.res = residual; a.y = y; a.wfrag1 = nullptr; a.bias1 = nullptr;
    a.Ca = Ca; a.Cb = Cb; a.Cbsrc = Cb_src; a.Cout = Cout; a.H = H; a.W = W;
    a.nch = (Ca + Cb + 15) / 16; a.mtot = (Cout + 31) / 32; a.amax = nullptr; a.ln_w = nullptr; a.ln_b = nullptr; a.ln_eps = 0.0f;
    ProfScope ps(ks == 3 ? 13 : 14, st);
    for (int mb = 0; mb < a.mtot;) {
        a.mbase = mb;
        const int left = a.mtot - mb;
        int rc;
        if (ks == 3) {
#ifndef WM_CONV_RW1
#define WM_CONV_RW1 3
#endif
            // 32 output channels: 12-row tiles (49 KB of LDS: three workgroups per compute unit, staging slots 93 % used)
            // beat 16-row tiles (two workgroups, 80 %) by 4-13 %; 64 channels keep 16 rows (two accumulator sets)
            if (conv_ws_enabled(a, B, 8)) {
                if (gate || residual) { rc = conv2d_ws_launch<WM_CONV_WS_RW1, 1, false, true>(a, B, st); mb += 1; }
                else if (left >= 2) { rc = conv2d_ws_launch<WM_CONV_WS_RW2, 2>(a, B, st); mb += 2; }
                else { rc = conv2d_ws_launch<WM_CONV_WS_RW1, 1, false, false, WM_CONV_WS_NPW1>(a, B, st); mb += 1; }
            } else if (left >= 2) { rc = conv2d_launch<3, 4, 2>(a, B, st); mb += 2; }
            else { rc = conv2d_launch<3, WM_CONV_RW1, 1>(a, B, st); mb += 1; }
        } else {
            // 1x1 is bandwidth-bound: never read the input twice (3 row tiles in one launch on an 8-row tile)
            if (left >= 3) { rc = conv2d_launch<1, 2, 3>(a, B, st); mb += 3; }
            else if (left == 2) { rc = conv2d_launch<1, 4, 2>(a, B, st); mb += 2; }
            else { rc = conv2d_launch<1, 4, 1>(a, B, st); mb += 1; }
        }
        if (rc) return rc;
    }
    return WM_OK;
}

// y = conv1x1(LayerNorm2d(x)) + bias (+ residual): the LayerNorm of a 32-channel map inside the 1x1 kernel's staging (conv2d.hip.h, LNIN).
int wm_conv2d_ln_fwd(const float* x, const float* ln_weight, const float* ln_bias, float ln_eps, const void* wfrag, const float* bias,
                     const float* residual, float* y, int B, int Cin, int Cout, int H, int W, void* stream) {
    if (B < 0 || Cin <= 0 || Cout <= 0 || H < 0 || W < 0) return WM_EINVAL;
    if (Cin != 32) return WM_EUNSUPPORTED;                 // callers run wm_layernorm2d_fwd + wm_conv2d_fwd
    if (B == 0 || H == 0 || W == 0) return WM_OK;
    if (!x || !ln_weight || !ln_bias || !wfrag || !y) return WM_ENULL;
    if (B > 65535 || (long long)H * W >= (1ll << 31)) return WM_EUNSUPPORTED;
    if (!aligned16(wfrag)) return WM_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    Conv2dArgs a;
    a.xa = x; a.xb = nullptr; a.xb_idx = nullptr; a.wfrag = (const uint4*)wfrag;
    a.bias = bias; a.gate = nullptr; a.res = residual; a.y = y; a.wfrag1 = nullptr; a.bias1 = nullptr;
    a.Ca = Cin; a.Cb = 0; a.Cbsrc = 0; a.Cout = Cout; a.H = H; a.W = W;
    a.nch = 2; a.mtot = (Cout + 31) / 32; a.amax = nullptr; a.ln_w = ln_weight; a.ln_b = ln_bias; a.ln_eps = ln_eps;
    ProfScope ps(14, st);
    for (int mb = 0; mb < a.mtot;) {
        a.mbase = mb;
        const int left = a.mtot - mb;
        int rc;
        if (left >= 3) { rc = conv2d_launch<1, 2, 3, false, false, true>(a, B, st); mb += 3; }
        else if (left == 2) { rc = conv2d_launch<1, 4, 2, false, false, true>(a, B, st); mb += 2; }
        else { rc = conv2d_launch<1, 4, 1, false, false, true>(a, B, st); mb += 1; }
        if (rc) return rc;
    }
    return WM_OK;
}

// The training form (conv2d.hip.h, fp16 split with per-tensor power-of-two scales): y = conv(x, w) + bias, ks in {1, 3}; wfrag from
// cv_amax_prep_kernel with the SAME amax buffer {max |x|, max |w|} (device floats).
static int conv2d_fwd_f16(const float* x, const void* wfrag, const float* amax, const float* bias, float* y, int B, int Cin, int Cout,
                          int H, int W, int ks, void* stream) {
    if (B < 0 || Cin <= 0 || Cout <= 0 || H < 0 || W < 0) return WM_EINVAL;
    if (ks != 1 && ks != 3) return WM_EUNSUPPORTED;
    if (B == 0 || H == 0 || W == 0) return WM_OK;
    if (!x || !wfrag || !y || !amax) return WM_ENULL;
    if (B > 65535 || (long long)H * W >= (1ll << 31)) return WM_EUNSUPPORTED;
    if (!aligned16(wfrag)) return WM_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    Conv2dArgs a;
    a.xa = x; a.xb = nullptr; a.xb_idx = nullptr; a.wfrag = (const uint4*)wfrag;
    a.bias = bias; a.gate = nullptr; a.res = nullptr; a.y = y; a.wfrag1 = nullptr; a.bias1 = nullptr;
    a.Ca = Cin; a.Cb = 0; a.Cbsrc = 0; a.Cout = Cout; a.H = H; a.W = W;
    a.nch = (Cin + 15) / 16; a.mtot = (Cout + 31) / 32; a.amax = amax; a.ln_w = nullptr; a.ln_b = nullptr; a.ln_eps = 0.0f;
    ProfScope ps(ks == 3 ? 13 : 14, st);
    for (int mb = 0; mb < a.mtot;) {
        a.mbase = mb;
        const int left = a.mtot - mb;
        int rc;
        if (ks == 3) {
            if (conv_ws_enabled(a, B, 8)) {
                if (left >= 2) { rc = conv2d_ws_launch<WM_CONV_WS_RW2, 2, false, false, 4, true>(a, B, st); mb += 2; }
                else { rc = conv2d_ws_launch<WM_CONV_WS_RW1, 1, false, false, WM_CONV_WS_NPW1, true>(a, B, st); mb += 1; }
            } else if (left >= 2) { rc = conv2d_launch<3, 4, 2, false, true>(a, B, st); mb += 2; }
            else { rc = conv2d_launch<3, WM_CONV_RW1, 1, false, true>(a, B, st); mb += 1; }
        } else {
            if (left >= 3) { rc = conv2d_launch<1, 2, 3, false, true>(a, B, st); mb += 3; }
            else if (left == 2) { rc = conv2d_launch<1, 4, 2, false, true>(a, B, st); mb += 2; }
            else { rc = conv2d_launch<1, 4, 1, false, true>(a, B, st); mb += 1; }
        }
        if (rc) return rc;
    }
    return WM_OK;
}

// The training step's convolution (fp16 split): `amax` (two floats; a slot of the caller's zeroed arena skips the memset node) and the
// fragments in separate buffers, magnitudes + weight preparation in ONE launch (cv_amax_prep_kernel), then the convolution; dgrad: the
// input-gradient convolution of the forward weight `weight` (conv2d.hip.h: cv_prep_item).  Two launches per convolution where round 4 had
// four (memset, magnitudes, preparation, convolution) - and six with autograd's flipped copy of the weight.
int wm_conv2d_f16_steps(const float* x, const float* weight, const float* bias, float* y, float* amax, void* wfrag, int B, int Cin,
                        int Cout, int H, int W, int ks, int dgrad, void* stream) {
    if (B < 0 || Cin <= 0 || Cout <= 0 || H < 0 || W < 0) return WM_EINVAL;
    if (ks != 1 && ks != 3) return WM_EUNSUPPORTED;
    if (wm_conv2d_wfrag_bytes(Cout, Cin, ks) == 0) return WM_EUNSUPPORTED;
    if (B == 0 || H == 0 || W == 0) return WM_OK;
    if (!x || !weight || !y || !amax || !wfrag) return WM_ENULL;
    if (!aligned16(wfrag)) return WM_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (zero_out(amax, 2 * sizeof(float), st) != hipSuccess) return WM_EHIP;
    const long long nx = (long long)B * Cin * H * W, nw = (long long)Cout * Cin * ks * ks;
    long long blocks = (nx / 4 + 256 * 8 - 1) / (256 * 8);           // >= 8 float4 per thread
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
    const int nch = (Cin + 15) / 16, mtot = (Cout + 31) / 32;
    const long long items = (long long)nch * ks * ks * mtot * 128;
    const int nprep = (int)(items <= 512 ? 1 : (items >= 32 * 512 ? 32 : (items + 511) / 512));   // ~2 fragment items per thread
    hipLaunchKernelGGL(cv_amax_prep_kernel, dim3((unsigned)blocks + nprep), dim3(256), 0, st, x, nx, weight, nw, (unsigned*)amax,
                       (uint4*)wfrag, Cout, Cin, ks * ks, nch, mtot, dgrad ? 1 : 0, nprep);
    int rc = launch_status();
    if (rc) return rc;
    return conv2d_fwd_f16(x, wfrag, amax, bias, y, B, Cin, Cout, H, W, ks, stream);
}

int wm_conv2d_gated_fwd(const float* xa, const float* xb, const int* xb_index, const void* wfrag3, const void* wfrag1,
                        const float* bias1, float* y, int B, int Ca, int Cb, int Cb_src, int Cout, int H, int W,
                        void* stream) {
    if (B < 0 || Ca <= 0 || Cb < 0 || Cout <= 0 || H < 0 || W < 0 || Cb_src < 0) return WM_EINVAL;
    if (B == 0 || H == 0 || W == 0) return WM_OK;
    if (!xa || !wfrag3 || !wfrag1 || !y || (Cb > 0 && !xb)) return WM_ENULL;
    if (Cb > 0 && Ca % 8 != 0) return WM_EUNSUPPORTED;
    if (Cb > 0 && !xb_index && Cb_src != Cb) return WM_EINVAL;
    if (B > 65535 || (long long)H * W >= (1ll << 31)) return WM_EUNSUPPORTED;
    if (!aligned16(wfrag3) || !aligned16(wfrag1)) return WM_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    Conv2dArgs a;
    a.xa = xa; a.xb = Cb > 0 ? xb : nullptr; a.xb_idx = Cb > 0 ? xb_index : nullptr; a.wfrag = (const uint4*)wfrag3;
    a.bias = nullptr; a.gate = nullptr; a.res = nullptr; a.y = y; a.wfrag1 = (const uint4*)wfrag1; a.bias1 = bias1;
    a.Ca = Ca; a.Cb = Cb; a.Cbsrc = Cb_src; a.Cout = Cout; a.H = H; a.W = W;
    a.nch = (Ca + Cb + 15) / 16; a.mtot = (Cout + 31) / 32; a.amax = nullptr; a.ln_w = nullptr; a.ln_b = nullptr; a.ln_eps = 0.0f;
    ProfScope ps(13, st);
    for (int mb = 0; mb < a.mtot;) {
        // two accumulator sets per wave: 64 channels x 8-row tiles read the input once (0.79 ms against 0.90 ms for
        // two 32-channel x 16-row launches at UHD level 1, 64 -> 64); a last odd row tile takes the 16-row form
        a.mbase = mb;
        int rc;
        const bool two = a.mtot - mb >= 2;                   // an odd tail runs one 32-channel tile on 8-row tiles
        if (conv_ws_enabled(a, B, two ? 2 * WM_CONV_WS_RWG : 8)) {
            if (two) { rc = conv2d_ws_launch<WM_CONV_WS_RWG, 2, true>(a, B, st); mb += 2; }
            else { rc = conv2d_ws_launch<4, 1, true>(a, B, st); mb += 1; }
        } else if (a.mtot - mb >= 2) { rc = conv2d_launch<3, 2, 2, true>(a, B, st); mb += 2; }
        else { rc = conv2d_launch<3, 4, 1, true>(a, B, st); mb += 1; }
        if (rc) return rc;
    }
    return WM_OK;
}

int wm_conv2d_select(int mode) {
    if (mode < 0 || mode > 2) return WM_EINVAL;
    g_conv_select.store(mode, std::memory_order_relaxed);
    return WM_OK;
}

#if WM_BWD_STAMP
int wm_debug_bwd_stamps(unsigned long long* out, int reset) {      // host buffer of 44 values
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wm::g_bwd_stamps), sizeof(unsigned long long) * 44);
    if (rc == 0 && reset) {
        unsigned long long z[44] = {};
        rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(wm::g_bwd_stamps), z, sizeof(z));
    }
    return rc;
}
#endif

#if WM_CV_STAMP
int wm_debug_conv_stamps(unsigned long long* out) {      // host buffer of 2 * 128 * 8 values
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wm::g_cv_stamps), sizeof(unsigned long long) * 2 * 128 * 8);
}
#endif

void wm_prof_enable(unsigned mask) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.mask = mask;
    if (mask)
        for (auto& v : g_prof.rec) {
            for (auto& pr : v) { g_prof.pool.push_back(pr.first); g_prof.pool.push_back(pr.second); }
            v.clear();
        }
}

int wm_prof_collect(int* launches, double* total_ms) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (int k = 0; k < WM_PROF_NKERNELS; ++k) {
        double tot = 0.0;
        for (auto& pr : g_prof.rec[k]) {
            hipError_t e = hipEventSynchronize(pr.second);
            if (e != hipSuccess) return (int)e;
            float ms = 0.f;
            e = hipEventElapsedTime(&ms, pr.first, pr.second);
            if (e != hipSuccess) return (int)e;
            tot += ms;
        }
        launches[k] = (int)g_prof.rec[k].size();
        total_ms[k] = tot;
    }
    return WM_OK;
}

int wm_event_synchronize_relaxed(void* event) {
    if (!event) return WM_EINVAL;
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    if (hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) return WM_EHIP;
    const hipError_t e = hipEventSynchronize(static_cast<hipEvent_t>(event));
    hipThreadExchangeStreamCaptureMode(&mode);
    return e == hipSuccess ? WM_OK : WM_EHIP;
}

}  // extern "C"
