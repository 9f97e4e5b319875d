// ss2d_core.hip.h - SS2D.forward_core (/root/reference/basicsr/archs/wavemamba_arch.py:446-478) as three launches
// for gfx950: chunk-reduce (all four directions) -> carry -> chunk-scan (all four directions).
//
//   xs = [x row-major | x column-major | their flips]                                              (:451-452)
//   x_dbl[k] = x_proj_weight[k] . xs[k] -> (dt_r | B | C);  dts[k] = dt_projs_weight[k] . dt_r       (:453-455)
//   y[k] = selective_scan(xs[k], dts[k], -exp(A_logs), B, C, Ds, dt_projs_bias, softplus)           (:465-471)
//   flips / transposes back to row-major                                                             (:474-478)
//
// One kernel body serves every direction.  lane = channel (D <= 64), a wave walks ONE sequence chunk in tiles of 16
// scan steps, a workgroup is NW waves:
//   * the x tile [64 channels][16 steps] of a wave is BOTH its `u` operand and the B operand of the x_proj GEMM:
//     (dt_r | B | C)[16 steps] = Wx[k] (R + 2N rows, padded to 16-row tiles) x tile.  Both operands are split into two
//     bf16 terms (v = hi + lo, |v - hi - lo| <= 2^-17 |v|) and the product is accumulated in fp32 as hi*hi + hi*lo +
//     lo*hi on v_mfma_f32_16x16x32_bf16 (two K-steps of 32 channels, 6 matrix instructions per 16-row tile): ~4e-6
//     relative on (dt_r | B | C) against the contract's 1e-4, the same split every dense convolution of this library
//     uses.  The fp32-input matrix instructions (v_mfma_f32_16x16x4_f32, exact products: round 2's kernel) run at the fp32
//     VECTOR rate and their time ADDS to the recurrence's VALU time (2.1 of 14.1 ms per UHD step); the bf16 ones are 16x
//     faster per FLOP and run beside the VALU.  Weight fragments
//     (pre-split, fragment-ordered), A * log2(e) and the per-channel constants come from a once-per-call prep kernel
//     (ss2d_core_prep_kernel): a workgroup's prologue is one 12-KB copy into LDS.  x_dbl / dts / xs never exist in
//     HBM - the first version wrote 576 B of projection records per position and re-read them in 16 launches.
//   * row directions (k = 0, 2): a tile is 16 consecutive row-major positions (64 B per channel row), wave-private
//     staging, no workgroup barrier in the loop; k = 2 walks the same tiles backwards (no flip).
//   * column directions (k = 1, 3): the NW waves of a workgroup own NW ADJACENT COLUMNS and the same 16 rows: the
//     workgroup fetches [64 ch][16 rows][NW columns] with 4*NW-byte runs (one 64-B DRAM burst at NW = 16) and
//     scatters the columns to the waves' tiles through LDS - no transposed copy of x or y ever exists, and the
//     recurrence code is the row directions' code.
//   * L-split: chunk-reduce (P = prod a, H = end state from 0) -> carry over chunks -> chunk-scan from H_in.  Every
//     exponential is evaluated twice (once per pass): with lane = channel and N states per lane there is nothing
//     else to repeat.  Ragged tails are masked steps (dt := 0 => a = 1, b = 0: the state passes through).
//   * y[k] goes to the direction's own (B, D, L) buffer in row-major positions; the consumer adds the four
//     (y1 + y2 + y3 + y4 of :490) while it reads them, so the scan launch has no read-modify-write and the four
//     directions run concurrently, bit-reproducibly.  (Round 4 built the two-plane alternative - the reversed directions as a
//     second, read-modify-write scan launch - and measured it 3.5 ms per UHD image slower, profiles/r04/core_ab_paired_planes.txt:
//     launch granularity, and the bytes are the same - write + read-modify-write + two planes read = four written + four read.
//     Round 5 built the column directions as wave-private strips of four columns (no LDS transposition, no barrier) and measured
//     them equal at UHD level 1 and 10-13 % slower below: profiles/r05/core_column_strips_experiment.txt.  Both deleted.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "selscan.hip.h"
#include "haar.hip.h"          // bf16_t and its conversions

#ifndef WM_CORE_ABLATE
#define WM_CORE_ABLATE 0          // timing experiments only (wrong results): 1 = no MFMA, 2 = no scan steps, 4 = no y store
#endif
#ifndef WM_CORE_STAMP
#define WM_CORE_STAMP 0           // diagnostics: per-wave s_memtime phase totals into CoreArgs::stamps (tools/core_stamps.py)
#endif
#ifndef WM_CORE_ROW_SYNC
#define WM_CORE_ROW_SYNC 0        // row directions: workgroup barrier every this many tiles (0 = never), see core_body
#endif
#ifndef WM_CORE_STEP_FENCE
#define WM_CORE_STEP_FENCE 1      // scheduling barrier after every scan step
#endif

namespace wm {

typedef float core_f4 __attribute__((ext_vector_type(4)));

// Storage type of the x / y planes: float, or bf16_t in the bf16-storage mode (four elements per lane access either way;
// everything inside the kernel - LDS tiles, projection, state - is fp32).
template <typename TP> struct CoreIO;
template <> struct CoreIO<float> {
    typedef float4 raw;
    static __device__ __forceinline__ raw load(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ float4 cvt(raw r) { return r; }
    static __device__ __forceinline__ void store(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
};
template <> struct CoreIO<bf16_t> {
    typedef uint2 raw;
    static __device__ __forceinline__ raw load(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
    static __device__ __forceinline__ float4 cvt(raw r) {
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                           __uint_as_float(r.y & 0xffff0000u));
    }
    static __device__ __forceinline__ void store(bf16_t* p, float4 v) {
        uint2 a;
        a.x = (uint32_t)float_to_bf16_bits(v.x) | ((uint32_t)float_to_bf16_bits(v.y) << 16);
        a.y = (uint32_t)float_to_bf16_bits(v.z) | ((uint32_t)float_to_bf16_bits(v.w) << 16);
        *reinterpret_cast<uint2*>(p) = a;
    }
};

struct CoreArgs {
    const void* x;           // (B, D, H, W), fp32 or bf16
    const float* Wx;         // (4, R + 2N, D)    x_proj_weight
    const float* Wdt;        // (4, D, R)         dt_projs_weight
    const float* dtb;        // (4, D)            dt_projs_bias
    const float* A_logs;     // (4 D, N)
    const float* Ds;         // (4 D)
    void* y[4];              // y of direction k, (B, D, L), row-major positions, fp32 or bf16
    float* wsP[4];           // chunk summaries of direction k: [chunk][b * D + d][NP]
    float* wsH[4];
    int B, D, H, W, L, N, R;
    int row_chunk, row_nchunks, row_wgs;        // steps per row chunk (multiple of 16), chunks, workgroups per direction
    int row_cpw;                                // row chunks per workgroup (handed to its waves on demand)
    unsigned long long* stamps;                 // WM_CORE_STAMP builds: [workgroup][wave][12] cycle totals / stamps, else unused
    int dirmask;                                // bit k set: direction k runs (tools: time one direction alone)
    const float* prep;                          // ss2d_core_prep_kernel's output: 4 x CoreCfg<NP>::PREP floats
    int col_seg, col_nseg, col_tiles, col_wgs;  // rows per column segment (multiple of 16), segments, column tiles,
                                                // workgroup slots per direction (col_tiles * col_nseg rounded up to 8)
};

template <int NP> struct CoreCfg {
    static constexpr int NTB = NP / 16;                  // 16-row tiles of B (and of C)
    static constexpr int NT3 = 2 * NTB + 1;              // row tiles of x_proj: dt_r | B.. | C..
    static constexpr int RS = 2 * NP + 4;                // record: [dt_r (4) | B (NP) | C (NP)] floats
    static constexpr int ROW = 20;                       // x tile row stride (floats): conflict-free per-lane float4
    static constexpr int XT = 64 * ROW + 4;              // x tile stride: the column scatter is 2-way at worst
    static constexpr int WF = NT3 * 1024 + 512;          // [tile][K-step (2)][hi | lo][lane] x 8 bf16 = 1 KB each, + the dt_r tile's
                                                         // third term [K-step (2)][lane] (P_W3)
    // prep buffer of ONE direction (floats): bf16 weight fragments | A * log2(e) as [n / 2][lane] pairs |
    // [wdt0 wdt1 wdt2 wdt3 bias D][lane] | third bf16 term of the dt_r row tile's fragments [K-step (2)][lane][4 dwords]
    static constexpr int P_WF = NT3 * 1024;
    static constexpr int P_A2 = P_WF;
    static constexpr int P_LC = P_A2 + NP * 64;
    static constexpr int P_W3 = P_LC + 6 * 64;
    static constexpr int PREP = P_W3 + 512;
};
#ifndef WM_CORE_LDS_PAD
#define WM_CORE_LDS_PAD 0         // experiments: extra dynamic LDS per workgroup (bytes), e.g. to keep a second workgroup off the compute unit
#endif
template <int NP, int NW> constexpr int core_lds_bytes() {
    return (CoreCfg<NP>::WF + NW * CoreCfg<NP>::XT + NW * 16 * CoreCfg<NP>::RS + 4 /* row-chunk counter */) * 4 + WM_CORE_LDS_PAD;
}

typedef __bf16 core_bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 core_bf8 __attribute__((ext_vector_type(8)));
typedef float core_f2 __attribute__((ext_vector_type(2)));

// v -> (hi, lo) bf16 with v ~ hi + lo (round to nearest even twice): v_cvt_pk_bf16_f32, two bit operations, one
// packed subtract, v_cvt_pk_bf16_f32 per pair of values
__device__ __forceinline__ void core_split2(float a, float b, core_bf2& hi, core_bf2& lo) {
    const core_f2 p = {a, b};
    hi = __builtin_convertvector(p, core_bf2);
    lo = __builtin_convertvector(p - __builtin_convertvector(hi, core_f2), core_bf2);
}

// Three terms (24 significant bits): v ~ hi + mid + lo with (hi, mid) = core_split2's pair.  The dt_r rows of x_proj take it
// (round 6): dt = softplus(Wdt . dt_r + bias) enters exp(dt A), which amplifies an error of the dt pre-activation by |A| dt - with
// trained-like parameters (|A| up to e^6, projections several times their init) or activations 100x the usual the two-term
// product's 2^-17 put the core's outputs 1e-4 .. 8e-4 from the float64 truth where the reference's fp32 arithmetic is at 4e-6 ..
// 4e-5 (tools/core_ood_report.py, profiles/r06/core_ood_report_before.txt).  B and C enter linearly and keep the two-term form.
__device__ __forceinline__ void core_split3(float a, float b, core_bf2& hi, core_bf2& mid, core_bf2& lo) {
    const core_f2 p = {a, b};
    hi = __builtin_convertvector(p, core_bf2);
    const core_f2 r = p - __builtin_convertvector(hi, core_f2);
    mid = __builtin_convertvector(r, core_bf2);
    lo = __builtin_convertvector(r - __builtin_convertvector(mid, core_f2), core_bf2);
}

// Once per call: everything a workgroup's prologue used to compute from the parameters (22 k cycles per workgroup and
// pass in round 2: 7 % of a UHD level-3 workgroup's life).  Block k = direction k.
//   fragments: lane l = (r16 = l & 15, g4 = l >> 4) of (tile t, K-step s2) holds W[row(t, r16)][32 s2 + 4 j + g4],
//   j = 0..7 - the K order of a matrix product is free, and this one lets the B operands be read from the x tile
//   [channel][step] with the bank-conflict-free stride of the fp32 kernel (g4 -> one tile row apart).
// L2U ("log2 units", the FORWARD kernels' form since round 5): the scan kernels work with dt' = dt / ln 2 = log2(1 + 2^x'),
// x' = (Wdt . dt_r + bias) log2(e) - so the prepared dt weights and bias carry the factor log2(e), A is stored WITHOUT it
// (exp(dt A) = exp2(dt' A)) and the B rows of x_proj carry ln 2 (dt u B = dt' u (ln 2 B)): the per-step softplus loses its
// two scale multiplications and its threshold selects (core_softplus_l2), nothing else changes.  The backward kernels
// (ss2d_core_bwd.hip.h) take the natural-unit form (L2U = false).
template <int NP, bool L2U = false>
__global__ __launch_bounds__(256) void ss2d_core_prep_kernel(const float* __restrict__ Wx, const float* __restrict__ Wdt,
                                                             const float* __restrict__ dtb, const float* __restrict__ A_logs,
                                                             const float* __restrict__ Ds, float* __restrict__ prep, int D, int N,
                                                             int R) {
    using Cfg = CoreCfg<NP>;
    constexpr int NTB = Cfg::NTB;
    const int k = blockIdx.x;
    float* out = prep + (long long)k * Cfg::PREP;
    const int Cx = R + 2 * N;
    uint32_t* wf = reinterpret_cast<uint32_t*>(out);
    for (int e = threadIdx.x; e < Cfg::NT3 * 2 * 64 * 4; e += 256) {        // one (hi, lo) bf16 pair-of-pairs per item
        const int jp = e & 3, l = (e >> 2) & 63, s2 = (e >> 8) & 1, t = e >> 9;
        const int r16 = l & 15, g4 = l >> 4;
        int row = -1;
        if (t == 0) { if (r16 < R) row = r16; }
        else if (t <= NTB) { const int n = 16 * (t - 1) + r16; if (n < N) row = R + n; }
        else { const int n = 16 * (t - 1 - NTB) + r16; if (n < N) row = R + N + n; }
        float v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int d = 32 * s2 + 4 * (2 * jp + i) + g4;
            v[i] = (row >= 0 && d < D) ? Wx[((long long)k * Cx + row) * D + d] : 0.0f;
            if (L2U && t >= 1 && t <= NTB) v[i] *= 0.6931471805599453f;       // B rows: ln 2 (see above)
        }
        core_bf2 hi, lo, l3;
        core_split3(v[0], v[1], hi, lo, l3);                             // (hi, lo) = core_split2's pair
        const int base = ((t * 2 + s2) * 2) * 256 + l * 4 + jp;          // [tile][K-step][split][lane][4 dwords]
        wf[base] = *reinterpret_cast<uint32_t*>(&hi);
        wf[base + 256] = *reinterpret_cast<uint32_t*>(&lo);
        if (t == 0) reinterpret_cast<uint32_t*>(out + Cfg::P_W3)[s2 * 256 + l * 4 + jp] = *reinterpret_cast<uint32_t*>(&l3);
    }
    for (int e = threadIdx.x; e < NP * 64; e += 256) {
        const int lane = e & 63, n = e >> 6;
        float a = 0.0f;
        if (lane < D && n < N) a = -expf(A_logs[((long long)k * D + lane) * N + n]) * (L2U ? 1.0f : 1.4426950408889634f);
        out[Cfg::P_A2 + ((n >> 1) * 64 + lane) * 2 + (n & 1)] = a;
    }
    for (int e = threadIdx.x; e < 6 * 64; e += 256) {
        const int lane = e & 63, c = e >> 6;
        float v = 0.0f;
        if (lane < D) {
            const long long kd = (long long)k * D + lane;
            if (c < 4) v = c < R ? Wdt[kd * R + c] : 0.0f;
            else v = c == 4 ? dtb[kd] : Ds[kd];
            if (L2U && c <= 4) v *= 1.4426950408889634f;
        }
        out[Cfg::P_LC + e] = v;
    }
}

// softplus in log2 units: x' = x log2(e) -> softplus(x) / ln 2 = log2(1 + 2^x') = max(x', 0) + log2(1 + 2^-|x'|).
// e = 2^-|x'| <= 1, so w = fl(1 + e) <= 2 and the log1p correction of selscan.hip.h's softplus2 (d = (w - 1) - e, exact) applies
// everywhere: log2(w - d) = log2(w) - d log2(e) / w, the 1 / w dropped as there.  No overflow for any x', hence no threshold
// select (F.softplus's `x > 20 ? x : ...` differs from this by e^-20 = 2e-9 of x), no w < 2 select, no scale multiplications:
// per PAIR of steps 4 transcendentals + 6 packed + 2 plain VALU operations (softplus2 + its scale: 4 + 5 + 8).
__device__ __forceinline__ v2f core_softplus_l2(v2f x) {
    const v2f m = (v2f){fmaxf(x.x, 0.0f), fmaxf(x.y, 0.0f)};
    const v2f e = exp2_2(x - m * 2.0f);                                  // 2^-|x'|
    const v2f w = e + 1.0f;
    const v2f d = (w - 1.0f) - e;
    const v2f lg = (v2f){__builtin_amdgcn_logf(w.x), __builtin_amdgcn_logf(w.y)};      // v_log_f32: base 2
    return m + (lg - d * 1.4426950408889634f);
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() also drains the vector-memory counter
// (s_waitcnt vmcnt(0)): in the column directions that made every barrier wait for the next tile's prefetch and for the
// y stores of the tile just written - ~5,000 cycles per tile (tools/core_stamps.py).  Nothing in these kernels hands
// global data from one wave to another, so the LDS counter is all a barrier has to wait for.
__device__ __forceinline__ void core_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void core_lds_fence() {       // LDS hand-off between the lanes of ONE wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// PHASE 1: chunk summaries.  PHASE 3: scan from the carried-in state, emits y.
// RHI: dt_rank > 2 (the dt projection reads four record slots instead of two).
// VEC: 16-byte tile accesses (W % 4 == 0, 16-byte aligned planes) or element-wise ones with per-element masks (any W, fp32
// planes: odd map widths are rare - the network pads its input to multiples of 8 - and take the same kernel, slower).
template <int NP, int NW, int PHASE, bool RHI, typename TP, bool COL, bool REV, bool VEC>
__device__ __forceinline__ void core_body(const CoreArgs& p, const int k, const int b, const int wg, float* smem) {
    using Cfg = CoreCfg<NP>;
    constexpr int NTB = Cfg::NTB, RS = Cfg::RS, ROW = Cfg::ROW, XT = Cfg::XT;
    constexpr int NT = (PHASE == 3 ? 2 * NTB : NTB) + 1;           // MFMA row tiles this phase needs: dt, B.., (C..)
    constexpr int QPR = NW / 4;                                    // float4 per tile row of a column-mode fetch
#if WM_CORE_STAMP
    const unsigned long long st_entry = wall_clock64();          // 100 MHz, chip-wide (the cycle counter is per compute unit)
#endif
    float* s_w = smem;
    float* s_x = smem + Cfg::WF;
    float* s_rec = s_x + NW * XT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = p.D, H = p.H, W = p.W;
    const long long L = p.L;
    const float* prep = p.prep + (long long)k * Cfg::PREP;

    // ---- the bf16 weight fragments of the row tiles this pass needs (dt_r, B.. and, in the scan pass, C..): one
    // 16-byte copy per thread
    {
        const uint4* gw = reinterpret_cast<const uint4*>(prep);
        uint4* sw4 = reinterpret_cast<uint4*>(s_w);
        for (int e = tid; e < NT * 256; e += 64 * NW) sw4[e] = gw[e];
        const uint4* g3 = reinterpret_cast<const uint4*>(prep + Cfg::P_W3);          // third term of the dt_r tile: behind the tiles
        for (int e = tid; e < 128; e += 64 * NW) sw4[Cfg::NT3 * 256 + e] = g3[e];
    }

    // ---- per-lane (= per-channel) constants, prepared once per call
    const bool live = lane < D;
    const int d = live ? lane : 0;
    v2f A2r[NP / 2];                                     // A per state pair (log2 units: exp(dt A) = exp2(dt' A); in LDS instead: 4 % slower)
#pragma unroll
    for (int i = 0; i < NP / 2; ++i) A2r[i] = *reinterpret_cast<const v2f*>(prep + Cfg::P_A2 + (i * 64 + lane) * 2);
#define WM_A2(i) A2r[i]
    float wdt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wdt[r] = prep[Cfg::P_LC + r * 64 + lane];
    const float bias = prep[Cfg::P_LC + 4 * 64 + lane];
    const float Dd = prep[Cfg::P_LC + 5 * 64 + lane];

    // ---- the wave's sequence chunk(s)
    // Column mode: one chunk per wave - the NW waves of the workgroup own NW adjacent columns and move in step (barriers).
    // Row mode: the workgroup owns `row_cpw` consecutive chunks and its waves TAKE them one at a time from a counter in
    // LDS (summary slots are indexed by chunk, so which wave ran a chunk changes nothing: results stay bit-reproducible).
    // The free-running waves of a SIMD drift apart (the oldest wave wins the issue arbitration): in the UHD level-1
    // chunk-scan launch the waves of a row workgroup finished between 370 and 857 us (tools/core_stamps.py).  Handing
    // out several shorter chunks per wave evens that out (566 - 918 us) but does not shorten the launch - the host plans
    // one chunk per wave (core_plan: WM_CORE_ROW_SPLIT).
    int* s_next = reinterpret_cast<int*>(s_rec + NW * 16 * RS);
    if (!COL && tid == 0) *s_next = 0;
    __syncthreads();                                     // weight fragments (and the counter) visible

    using IO = CoreIO<TP>;
    const TP* xb = static_cast<const TP*>(p.x) + (long long)b * D * L;
    TP* yb = (PHASE == 3) ? static_cast<TP*>(p.y[k]) + (long long)b * D * L : nullptr;
    float* sx = s_x + wv * XT;                           // the wave's x / y tile  [64][ROW]
    float* srec = s_rec + wv * (16 * RS);                // the wave's record tile [16][RS]
    const int trow = lane >> 2, tq = lane & 3;

    for (int rep = 0;; ++rep) {
    int t_begin, t_end;            // scan steps [t_begin, t_end) of the wave's line (row mode: l; column mode: tau)
    long long chunk;               // summary slot, in scan order
    bool active;                   // the wave has a sequence at all
    int wlo = 0;                   // column mode: first image column of the workgroup's tile
    if (!COL) {
        int c = 0;
        if (lane == 0) c = atomicAdd(s_next, 1);
        c = __builtin_amdgcn_readfirstlane(c) + wg * p.row_cpw;
        if (c >= min((wg + 1) * p.row_cpw, p.row_nchunks)) break;
        active = true;
        chunk = c;
        t_begin = c * p.row_chunk;
        t_end = (int)min(L, (long long)t_begin + p.row_chunk);
    } else {
        if (rep > 0) break;
        const int ct = wg % p.col_tiles, sg = wg / p.col_tiles;
        const int omega = ct * NW + wv;                  // column in scan order
        active = omega < W;
        chunk = (long long)omega * p.col_nseg + sg;
        t_begin = sg * p.col_seg;
        t_end = min(H, t_begin + p.col_seg);
        wlo = REV ? W - NW - ct * NW : ct * NW;
    }
    const int ntiles = (t_end - t_begin + 15) >> 4;

    v2f h[NP / 2];
    const long long wsrow = ((chunk * p.B + b) * D + d) * NP;
    if (PHASE == 3 && active && chunk > 0) {
#pragma unroll
        for (int q = 0; q < NP / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(p.wsH[k] + wsrow + 4 * q);
            h[2 * q] = (v2f){v.x, v.y}; h[2 * q + 1] = (v2f){v.z, v.w};
        }
    } else {
#pragma unroll
        for (int n = 0; n < NP / 2; ++n) h[n] = splat(0.0f);
    }
    float sum_dt = 0.0f;

    // tile ti covers steps t0 .. t0+15 of the line; in memory that is positions / rows lo .. lo+15 ascending, and
    // step tt sits at tile column tt (forward) or 15 - tt (reversed).  Valid tile columns: [v_lo, v_hi).
    typename IO::raw xp[4];                              // the next tile, in flight (raw bits: converted when staged)
    // Loads are UNCONDITIONAL with clamped offsets (an `ok ? load : 0` compiles to a branch around the load plus
    // register copies behind it, i.e. a wait for the load right where it was issued); invalid elements are zeroed when
    // the tile is staged, one tile later.  Per thread and float4 i: element offset = tbase[i] + ti * tdelta (D * L < 2^31,
    // host check), valid iff the static mask bit i is set and the thread's tile column (row mode) / tile row (column
    // mode) `vc` lies in the tile's valid range - only a chunk's last tile is ever partial.
    // Element-wise form (VEC == false): the four elements of a quad are checked one by one (tile column in row mode,
    // image column in column mode) and nothing is prefetched - the tile is loaded when it is staged, which exposes one
    // load latency per tile (16 more offsets in flight on top of the scan's registers would spill at N <= 16, where a
    // 16-wave workgroup has 128 registers per lane).
    unsigned tbase[4];
    int smask = 0, vc, tdelta;
    int wq0[4] = {0, 0, 0, 0};                           // column mode: image column of the quad's first element
    if (!COL) {
        vc = 4 * tq;                                     // VEC: L % 4 == 0, chunk % 16 == 0: quads are all-in or all-out
        tdelta = REV ? -16 : 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = 16 * i + trow;
            tbase[i] = (unsigned)ch * (unsigned)L + (unsigned)((REV ? (int)L - 16 - t_begin : t_begin) + 4 * tq);
            smask |= (ch < D) << i;
        }
    } else {
        vc = (tid / QPR) & 15;                           // the same tile row for every i (64 * NW / QPR is a multiple of 16)
        tdelta = REV ? -16 * W : 16 * W;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * 64 * NW + tid;
            const int q = e % QPR, ch = e / (QPR * 16);
            const int wq = wlo + 4 * q;                                              // VEC: W % 4 == 0
            wq0[i] = wq;
            tbase[i] = ((unsigned)ch * (unsigned)H + (unsigned)((REV ? H - 16 - t_begin : t_begin) + vc)) * (unsigned)W + (unsigned)wq;
            smask |= (ch < D && (VEC ? (wq >= 0 && wq < W) : true)) << i;
        }
    }
    auto tile_ok = [&](int ti, int i) -> bool {
        const int tl = min(16, t_end - (t_begin + 16 * ti));
        const int v_lo = REV ? 16 - tl : 0, v_hi = REV ? 16 : tl;
        return ((smask >> i) & 1) && (COL || VEC ? (vc >= v_lo && vc < v_hi) : true);
    };
    auto elem_ok = [&](int ti, int i, int j) -> bool {    // element-wise form only
        if (!tile_ok(ti, i)) return false;
        if (COL) return wq0[i] + j >= 0 && wq0[i] + j < W;
        const int tl = min(16, t_end - (t_begin + 16 * ti));
        const int v_lo = REV ? 16 - tl : 0, v_hi = REV ? 16 : tl;
        return vc + j >= v_lo && vc + j < v_hi;
    };
    auto tile_off = [&](int ti, int i) -> unsigned { return tbase[i] + (unsigned)(ti * tdelta); };
    auto fetch = [&](int ti) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (VEC) {
                const unsigned off = tile_ok(ti, i) ? tile_off(ti, i) : 0u;
                xp[i] = IO::load(xb + off);
            }
        }
    };
    auto stage = [&](int ti) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v4;
            if constexpr (VEC) v4 = tile_ok(ti, i) ? IO::cvt(xp[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
            else {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = ld1(xb + (elem_ok(ti, i, j) ? tile_off(ti, i) + (unsigned)j : 0u));
                v4 = make_float4(elem_ok(ti, i, 0) ? e[0] : 0.f, elem_ok(ti, i, 1) ? e[1] : 0.f,
                                 elem_ok(ti, i, 2) ? e[2] : 0.f, elem_ok(ti, i, 3) ? e[3] : 0.f);
            }
            if (!COL) {
                *reinterpret_cast<float4*>(&sx[(16 * i + trow) * ROW + 4 * tq]) = v4;
            } else {
                const int e = i * 64 * NW + tid;
                const int q = e % QPR, r = (e / QPR) & 15, ch = e / (QPR * 16);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cc = 4 * q + j;                            // tile column = image column wlo + cc
                    const int owner = REV ? NW - 1 - cc : cc;            // the wave that scans it
                    s_x[owner * XT + ch * ROW + r] = v[j];
                }
            }
        }
    };
    // a quad of the y tile back to memory
    auto put = [&](int ti, int i, float4 v4) {
        if constexpr (VEC) {
            if (tile_ok(ti, i)) IO::store(yb + tile_off(ti, i), v4);
        } else {
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (elem_ok(ti, i, j)) {
                    TP* q = yb + (tile_off(ti, i) + (unsigned)j);
                    st1(q, v[j]);
                }
        }
    };
#if WM_CORE_STAMP == 1
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long st_kernel = st_entry;
    unsigned long long st_prev = __builtin_readcyclecounter();
    const unsigned long long st_begin = st_prev;
#define WM_STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_prev; st_prev = now_; }
#else
#define WM_STAMP(i)
#endif
    fetch(0);
    for (int ti = 0; ti < ntiles; ++ti) {
        stage(ti);
        if (COL) core_barrier(); else core_lds_fence();
        WM_STAMP(0)                                      // stage (incl. the wait for the tile's loads) + barrier
        // The next tile's loads are issued after the first four scan steps, not here: on gfx950 loads and stores share
        // one counter, and the compiler guards the reuse of the prefetch registers with waits that assume the loads
        // are the youngest vector-memory operations - issued here, right behind the previous tile's y stores, they made
        // every wave sit out the stores' latency (~2,300 cycles per tile in the column directions).
        if (!active && ti + 1 < ntiles) fetch(ti + 1);   // (a wave without a sequence still moves its share of the tile)
        WM_STAMP(1)
        // (wave-uniform: the per-step mask below is then a scalar compare + one v_cndmask, not a vector compare)
        const int tl = __builtin_amdgcn_readfirstlane(min(16, t_end - (t_begin + 16 * ti)));

        if (active) {
            // ---- projection: records of the 16 steps (tile columns) ----
            core_f4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = (core_f4){0.f, 0.f, 0.f, 0.f};
            const int g4 = lane >> 4, c16 = lane & 15;
            // B operands: lane (c16 = step, g4) of K-step s2 holds channels 32 s2 + 4 j + g4, j = 0..7 (the prep kernel's K
            // order) of tile column c16, split into bf16 hi / lo.  Channels >= D: zero-filled tile rows, zero weights.
            {
                const uint4* sw4 = reinterpret_cast<const uint4*>(s_w) + lane;
                constexpr int S2 = (WM_CORE_ABLATE & 1) ? 0 : 2;
#pragma unroll
                for (int s2 = 0; s2 < S2; ++s2) {
                    float xf[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) xf[j] = sx[(32 * s2 + 4 * j + g4) * ROW + c16];
                    core_bf8 xh, xl, x3;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        core_bf2 h2, l2, t2;
                        core_split3(xf[j], xf[j + 1], h2, l2, t2);
                        xh[j] = h2[0]; xh[j + 1] = h2[1]; xl[j] = l2[0]; xl[j + 1] = l2[1]; x3[j] = t2[0]; x3[j + 1] = t2[1];
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const uint4 wh4 = sw4[((t * 2 + s2) * 2 + 0) * 64], wl4 = sw4[((t * 2 + s2) * 2 + 1) * 64];
                        const core_bf8 wh = *reinterpret_cast<const core_bf8*>(&wh4);
                        const core_bf8 wl = *reinterpret_cast<const core_bf8*>(&wl4);
                        if (t == 0) {            // dt_r rows: the three further products of the 24-bit split, smallest first
                            const uint4 w34 = sw4[Cfg::NT3 * 256 + s2 * 64];
                            const core_bf8 w3 = *reinterpret_cast<const core_bf8*>(&w34);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3, xh, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, x3, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xl, acc[t], 0, 0, 0);
                        }
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[t], 0, 0, 0);
                    }
                }
            }
            // D layout: lane holds rows 4 g4 .. 4 g4 + 3 of tile column c16
            {
                float* rc = srec + c16 * RS;
                if (g4 == 0) *reinterpret_cast<core_f4*>(rc) = acc[0];
#pragma unroll
                for (int t = 1; t < NT; ++t) {
                    *reinterpret_cast<core_f4*>(rc + 4 + 16 * (t - 1) + 4 * g4) = acc[t];
                }
            }
            core_lds_fence();
            WM_STAMP(2)                                  // projection + record write

            // ---- 16 scan steps ----
            // The record addresses are wave-uniform; left to itself the compiler forms each of the ~40 per quad in an
            // SGPR and copies it to a VGPR for its ds_read.  One opaque per-lane base per quad + compile-time offsets
            // puts them in the instructions' offset fields instead.
#pragma unroll 1
            for (int q = 0; q < ((WM_CORE_ABLATE & 2) ? 0 : 4); ++q) {
                if (q == 1 && ti + 1 < ntiles) fetch(ti + 1);            // uniform; see the note at the top of the tile loop
                const int cq = REV ? 3 - q : q;
                const float4 u4 = *reinterpret_cast<const float4*>(&sx[lane * ROW + 4 * cq]);
                const float uu[4] = {REV ? u4.w : u4.x, REV ? u4.z : u4.y, REV ? u4.y : u4.z, REV ? u4.x : u4.w};
                int roff = 4 * cq * RS;                                   // records of tile columns 4 cq .. 4 cq + 3
                asm volatile("" : "+v"(roff));                            // (an opaque OFFSET: the pointer stays an LDS pointer)
                const float* rq = srec + roff;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {                          // two steps at a time (softplus on a float pair)
                    float dtr[2];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * hf + jj;
                        const float* rc = rq + (REV ? 3 - j : j) * RS;
                        if constexpr (RHI) {
                            const float4 dr = *reinterpret_cast<const float4*>(rc);
                            dtr[jj] = fmaf(wdt[3], dr.w, fmaf(wdt[2], dr.z, fmaf(wdt[1], dr.y, fmaf(wdt[0], dr.x, bias))));
                        } else {
                            const float2 dr = *reinterpret_cast<const float2*>(rc);
                            dtr[jj] = fmaf(wdt[1], dr.y, fmaf(wdt[0], dr.x, bias));
                        }
                    }
                    const v2f sp = core_softplus_l2((v2f){dtr[0], dtr[1]});             // dt / ln 2 (log2 units, see the prep kernel)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * hf + jj;
                        const float dt = (4 * q + j < tl) ? (jj ? sp.y : sp.x) : 0.0f;      // masked step: a = 1, b = 0
                        const float ut = uu[j];
                        const v2f dt2 = splat(dt), du2 = splat(dt * ut);
                        if (PHASE == 1) sum_dt += dt;
                        v2f y2 = splat(0.0f);
                        const float* rc = rq + (REV ? 3 - j : j) * RS + 4;
#pragma unroll
                        for (int r = 0; r < NP / 4; ++r) {
                            const v2f a0 = exp2_2(dt2 * WM_A2(2 * r));
                            const v2f a1 = exp2_2(dt2 * WM_A2(2 * r + 1));
                            const float4 bv = *reinterpret_cast<const float4*>(rc + 4 * r);
                            h[2 * r] = a0 * h[2 * r] + du2 * (v2f){bv.x, bv.y};
                            h[2 * r + 1] = a1 * h[2 * r + 1] + du2 * (v2f){bv.z, bv.w};
                            if (PHASE == 3) {
                                const float4 cv = *reinterpret_cast<const float4*>(rc + NP + 4 * r);
                                y2 = (v2f){cv.x, cv.y} * h[2 * r] + y2;
                                y2 = (v2f){cv.z, cv.w} * h[2 * r + 1] + y2;
                            }
                        }
                        if (PHASE == 3)      // y overwrites the consumed u (same lane, same row)
                            sx[lane * ROW + 4 * cq + (REV ? 3 - j : j)] = fmaf(Dd, ut, y2.x + y2.y);
#if WM_CORE_STEP_FENCE == 1
                        __builtin_amdgcn_sched_barrier(0);   // keep the next step's record reads out of this step's registers
#elif WM_CORE_STEP_FENCE == 2
                        if (jj == 1) __builtin_amdgcn_sched_barrier(0);      // ... per pair of steps
#endif
                    }
                }
            }
        }

        WM_STAMP(3)                                      // 16 scan steps
        if (PHASE == 3 && !(WM_CORE_ABLATE & 4)) {
            if (!COL) {
                core_lds_fence();
#pragma unroll
                for (int i = 0; i < 4; ++i) put(ti, i, *reinterpret_cast<const float4*>(&sx[(16 * i + trow) * ROW + 4 * tq]));
                core_lds_fence();                        // the y tile is read before the next stage() overwrites it
            } else {
                core_barrier();
                float v[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = i * 64 * NW + tid;
                    const int q = e % QPR, r = (e / QPR) & 15, ch = e / (QPR * 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int cc = 4 * q + j;
                        const int owner = REV ? NW - 1 - cc : cc;
                        v[i][j] = s_x[owner * XT + ch * ROW + r];
                    }
                }
                // a fully valid tile (the usual case) stores without per-element branches: the 16 LDS reads above are
                // in flight together instead of four read-wait-store rounds
                const bool full = VEC && tl == 16 && D == 64 && wlo >= 0 && wlo + NW <= W;     // uniform
                if (full) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        IO::store(yb + tile_off(ti, i), make_float4(v[i][0], v[i][1], v[i][2], v[i][3]));
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) put(ti, i, make_float4(v[i][0], v[i][1], v[i][2], v[i][3]));
                }
                core_barrier();
            }
        } else if (COL) {
            core_barrier();                             // every wave is done with its tile before the next stage()
        }
        WM_STAMP(4)                                      // y store (+ barriers)
        // Row directions need no barrier (wave-private tiles).  Free-running, the four waves of a SIMD drift apart (the
        // scheduler favours the oldest: wave lifetimes 500 .. 1,150 us inside one workgroup, tools/core_stamps.py), but
        // bounding the drift with a barrier every 1 / 2 / 4 tiles changed the launch time by < 1 %: off by default.
        // (Waves that have ended - shorter last chunk, no chunk at all - no longer count towards s_barrier.)
        if (!COL && WM_CORE_ROW_SYNC > 0 && (ti % (WM_CORE_ROW_SYNC > 0 ? WM_CORE_ROW_SYNC : 1)) == WM_CORE_ROW_SYNC - 1)
            core_barrier();
    }
#if WM_CORE_STAMP == 2
    if (lane == 0 && p.stamps) {          // light mode: workgroup entry / exit only (no per-phase accumulators, no scratch)
        unsigned long long* o = p.stamps + ((unsigned long long)blockIdx.x * NW + wv) * 12;
        o[6] = (unsigned long long)ntiles; o[7] = (unsigned long long)k; o[8] = st_entry; o[10] = wall_clock64();
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); o[11] = xcc & 15;
    }
#endif
#if WM_CORE_STAMP == 1
    if (lane == 0 && p.stamps) {
        unsigned long long* o = p.stamps + ((unsigned long long)blockIdx.x * NW + wv) * 12;
        for (int i = 0; i < 5; ++i) o[i] = st_acc[i];
        const unsigned long long st_end = __builtin_readcyclecounter();
        o[5] = st_end - st_begin; o[6] = (unsigned long long)ntiles; o[7] = (unsigned long long)k;
        o[8] = st_kernel; o[9] = st_begin; o[10] = st_end;
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); o[11] = xcc & 15;
    }
#endif
#undef WM_STAMP

    if (PHASE == 1 && active && live) {
#pragma unroll
        for (int q = 0; q < NP / 4; ++q) {
            *reinterpret_cast<float4*>(p.wsH[k] + wsrow + 4 * q) =
                make_float4(h[2 * q].x, h[2 * q].y, h[2 * q + 1].x, h[2 * q + 1].y);
            const v2f p0 = exp2_2(splat(sum_dt) * WM_A2(2 * q));
            const v2f p1 = exp2_2(splat(sum_dt) * WM_A2(2 * q + 1));
            *reinterpret_cast<float4*>(p.wsP[k] + wsrow + 4 * q) = make_float4(p0.x, p0.y, p1.x, p1.y);
        }
    }
    if (!COL) core_lds_fence();                          // the wave's tiles are its own: nothing of this chunk is pending
    }                                                    // next chunk (row mode)
}

#undef WM_A2

// Workgroup -> (batch, direction, slot).  Per batch: 2 * col_wgs column slots first (k = 1 / 3 interleaved), then
// 2 * row_wgs row slots (k = 0 / 2 interleaved): the column workgroups are the long ones (a whole column segment, three
// barriers per tile), the row workgroups are cut to about half their length by the host and fill in behind them
// (longest-first list scheduling: with equal-length workgroups in id order the second round of the UHD level-1
// launches ran 224 of 256 compute units for a whole workgroup lifetime).
// Column tiles 2 i and 2 i + 1 share every 128-byte line of x and y: their slots are 8 workgroup ids apart, i.e. on the
// same XCD (workgroup id -> XCD id % 8), whose L2 then holds the line.
// second launch-bound: minimum waves per SIMD (N <= 16: four, i.e. <= 128 registers - one 16-wave or two 8-wave
// workgroups per compute unit; N = 32: two)
template <int NP, int NW, int PHASE, bool RHI, typename TP = float, bool VEC = true>
__global__ __launch_bounds__(64 * NW, NP == 16 ? 4 : 2) void ss2d_core_kernel(CoreArgs p) {
    extern __shared__ __attribute__((aligned(16))) float core_smem[];
    const int ncol = p.col_wgs;
    const int per_b = 2 * (p.row_wgs + ncol);
    const int b = blockIdx.x / per_b;
    int r = blockIdx.x - b * per_b;
    const bool col = r < 2 * ncol;
    if (!col) r -= 2 * ncol;
    if (col) {
        const int idx = r >> 1;
        const int wg = (((idx >> 3) << 2) + (idx & 3)) * 2 + ((idx >> 2) & 1);
        if (wg >= p.col_tiles * p.col_nseg || !((p.dirmask >> ((r & 1) * 2 + 1)) & 1)) return;
        if (r & 1) core_body<NP, NW, PHASE, RHI, TP, true, true, VEC>(p, 3, b, wg, core_smem);
        else core_body<NP, NW, PHASE, RHI, TP, true, false, VEC>(p, 1, b, wg, core_smem);
    } else {
        const int wg = r >> 1;
        if (!((p.dirmask >> ((r & 1) * 2)) & 1)) return;
        if (r & 1) core_body<NP, NW, PHASE, RHI, TP, false, true, VEC>(p, 2, b, wg, core_smem);
        else core_body<NP, NW, PHASE, RHI, TP, false, false, VEC>(p, 0, b, wg, core_smem);
    }
}

// merged output for callers of the plain operator (training): y0 <- ((y0 + y2) + y1) + y3, the reference's order of
// y1 + y2 + y3 + y4 (:490: out_y[:, 0], inv_y[:, 0], wh_y, invwh_y)
template <typename TP, bool VEC>
__global__ __launch_bounds__(256) void ss2d_sum4_kernel(TP* __restrict__ y0, const TP* __restrict__ y2,
                                                        const TP* __restrict__ y1, const TP* __restrict__ y3,
                                                        long long n) {          // n: quads (VEC) or elements
    using IO = CoreIO<TP>;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if constexpr (VEC) {
        float4 a = IO::cvt(IO::load(y0 + 4 * i));
        const float4 c = IO::cvt(IO::load(y2 + 4 * i));
        const float4 bq = IO::cvt(IO::load(y1 + 4 * i));
        const float4 e = IO::cvt(IO::load(y3 + 4 * i));
        a.x = ((a.x + c.x) + bq.x) + e.x; a.y = ((a.y + c.y) + bq.y) + e.y;
        a.z = ((a.z + c.z) + bq.z) + e.z; a.w = ((a.w + c.w) + bq.w) + e.w;
        IO::store(y0 + 4 * i, a);
    } else {
        st1(y0 + i, ((ld1(y0 + i) + ld1(y2 + i)) + ld1(y1 + i)) + ld1(y3 + i));
    }
}

}  // namespace wm
