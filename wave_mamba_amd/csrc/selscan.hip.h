// selscan.hip.h - chunked selective-scan forward for gfx950 (MI355X).
//
// Replaces mamba_ssm's selective_scan_fn at its call site in SS2D.forward_core
// (/root/reference/basicsr/archs/wavemamba_arch.py:465-471).
//
//     dt_t = softplus(delta_t + bias)           a_t[n] = exp(dt_t * A[n])
//     h_t[n] = a_t[n] * h_{t-1}[n] + dt_t * B_t[n] * u_t         y_t = sum_n C_t[n] h_t[n] + D u_t
//
// Design (MI355X-first; NOT the upstream one-block-per-(batch,channel) layout, which would put
// 256 blocks on 256 CUs walking ~1000 chunks serially at UHD):
//
//  * lane = channel.  A wave owns 64 consecutive channels of ONE B/C group and walks time
//    sequentially.  The N states of a channel live in the lane's registers (N independent FMA
//    chains = ILP, no cross-lane traffic in the recurrence at all).  B_t / C_t are wave-uniform:
//    they are read from LDS as broadcast ds_read_b128, never per lane from memory.
//  * L-split.  The sequence is cut into chunks; chunk c of every wave-row is an independent
//    workgroup, so even batch 1 (4 wave-rows at d_inner 64) fills the chip.  Three phases:
//      1. chunk-reduce : local scan from h = 0 -> (P = prod a, H = end state) per (chunk, d, n)
//      2. carry        : H_in[c+1] = P[c] * H_in[c] + H[c]  over chunks (tiny)
//      3. chunk-scan   : local scan from h = H_in[c], emits y
//    a in (0,1] (A < 0 in the model, dt > 0), so re-association is numerically benign.
//  * HBM access.  u / delta tiles [64 channels][16 steps] and B / C tiles [N][16 steps] are fetched
//    with 16-byte lane accesses (each 64-B row segment of a channel is one DRAM burst), staged in
//    LDS, and consumed transposed (lane = channel row).  y goes back through the same LDS tile and
//    is written with 16-byte stores.  The next tile's loads are in flight while the current tile
//    is being scanned (register double buffering).
//  * exp(dt*A) is v_exp_f32 on a pre-scaled A*log2(e): one transcendental per (t, d, n).  That is
//    the real floor of this kernel: KD*N exp per position per pass (SURVEY.md 7 "transcendental
//    ceiling"); the HBM side is 3584 B per position at N = 16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wm {

#ifndef WM_KTILE
#define WM_KTILE 16
#endif
constexpr int kTile = WM_KTILE;   // time steps per LDS tile (16 -> 64-B, 32 -> 128-B row segments)
constexpr int kLpr = kTile / 4;   // lanes per tile row (one float4 each)
constexpr int kRpi = 64 / kLpr;   // tile rows covered by one wave-wide float4 access
constexpr int kNld = 64 / kRpi;   // such accesses per [64][kTile] tile
#ifndef WM_KROW
#define WM_KROW (WM_KTILE + 4)
#endif
#ifndef WM_LB_WAVES
#define WM_LB_WAVES 1
#endif
constexpr int kRow = WM_KROW;  // LDS row stride (floats) of the [64][kTile] tiles: 80 B keeps the
                               // per-lane ds_read_b128 of 16 consecutive rows on disjoint banks

struct ScanArgs {
    const float* u; const float* delta; const float* A; const float* Bm; const float* Cm;
    const float* D; const float* z; const float* bias;
    float* out; float* last_state;
    float* wsP; float* wsH;            // [nchunks][batch*dim][NP]
    int batch, dim, L, N, G;
    int dpg;                           // channels per group = dim / G
    int wpg;                           // waves per group = ceil(dpg / 64)
    int chunk_len, nchunks;            // chunk_len % kTile == 0
    int softplus;
};

typedef float v2f __attribute__((ext_vector_type(2)));   // operands of v_pk_{mul,add,fma}_f32

// gfx950 issues a plain fp32 VALU op over a wave64 in 4 cycles and a PACKED one (two fp32 per lane)
// in the same 4 cycles; v_exp_f32 / v_log_f32 / v_rcp_f32 take 8 (tools/microbench.hip).  So all
// non-transcendental math below is written on float pairs.
__device__ __forceinline__ v2f splat(float x) { return (v2f){x, x}; }
__device__ __forceinline__ v2f exp2_2(v2f x) {
    return (v2f){__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
}

// F.softplus(beta=1, threshold=20) = log1p(exp(x)) below the threshold, on two values at once.
// Hardware v_exp_f32 / v_log_f32 (base 2, ~1 ulp).  log1p(e) from w = fl(1 + e): the rounding error of that addition,
// d = (w - 1) - e, is exact in fp32, and log1p(e) = log(w - d) = log(w) - d / w + O(d^2).  Below w = 2 the division is
// dropped (d (1 - 1 / w) <= 2^-24 (w - 1): 1e-7 of the result, which is ~ w - 1 there); from w = 2 on the whole correction
// is (d / w <= 2^-24 against log(w) >= 0.69).  w == 1 (e below half an ulp of 1) gives 0 - (0 - e) = e with no special
// case.  (The first version divided: log(w) * e / (w - 1), two v_rcp_f32 and two packed multiplies more per pair.)
__device__ __forceinline__ v2f softplus2(v2f x) {
    const v2f e = exp2_2(x * 1.4426950408889634f);
    const v2f w = e + 1.0f;
    v2f d = (w - 1.0f) - e;
    d.x = w.x < 2.0f ? d.x : 0.0f;
    d.y = w.y < 2.0f ? d.y : 0.0f;
    const v2f lp = (v2f){__builtin_amdgcn_logf(w.x), __builtin_amdgcn_logf(w.y)} * 0.6931471805599453f - d;
    v2f r;
    r.x = x.x > 20.0f ? x.x : lp.x;
    r.y = x.y > 20.0f ? x.y : lp.y;
    return r;
}

// PHASE 1: reduce (no C, no y; writes P/H).  PHASE 3: scan (reads H_in, writes y).
// NP = N padded to 16 or 32 (padded states have A = 0, B = C = 0 -> stay exactly 0).
// VEC: L % 4 == 0 and 16-byte aligned bases -> float4 global access; else scalar access.
template <int NP, int PHASE, bool VEC>
__global__ __launch_bounds__(64, WM_LB_WAVES) void selscan_chunk_kernel(ScanArgs p) {
    __shared__ __attribute__((aligned(16))) float s_u[64 * kRow];
    __shared__ __attribute__((aligned(16))) float s_d[64 * kRow];
    __shared__ __attribute__((aligned(16))) float s_B[kTile * NP];
    __shared__ __attribute__((aligned(16))) float s_C[(PHASE == 3) ? kTile * NP : 4];

    const int lane = threadIdx.x;
    const int chunk = blockIdx.x;
    int wr = blockIdx.y;                             // wave-row: (b, g, sub)
    const int sub = wr % p.wpg; wr /= p.wpg;
    const int g = wr % p.G;
    const int b = wr / p.G;
    const int nch = min(64, p.dpg - sub * 64);       // live channels in this wave
    const int ch0 = g * p.dpg + sub * 64;            // first channel of the wave
    const bool live = lane < nch;
    const int d = ch0 + (live ? lane : 0);
    const long long L = p.L;
    const int t_begin = chunk * p.chunk_len;
    const int t_end = min(p.L, t_begin + p.chunk_len);

    // per-lane constants
    v2f A2[NP / 2];                                   // A * log2(e), as (n, n+1) pairs
    {
        float raw[NP];                                // (all loads first, from clamped indices: see fetch below)
#pragma unroll
        for (int n = 0; n < NP; ++n) raw[n] = p.A[(long long)d * p.N + min(n, p.N - 1)];
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            const float a = (n < p.N) ? raw[n] * 1.4426950408889634f : 0.0f;
            if (n & 1) A2[n / 2].y = a; else A2[n / 2].x = a;
        }
    }
    const float bias = p.bias ? p.bias[d] : 0.0f;
    const float Dd = (PHASE == 3 && p.D) ? p.D[d] : 0.0f;

    v2f h[NP / 2];                                    // the N states of this lane's channel
    const long long wsrow = ((long long)chunk * p.batch * p.dim + (long long)b * p.dim + d) * NP;
    if (PHASE == 3 && chunk > 0) {
#pragma unroll
        for (int q = 0; q < NP / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(p.wsH + wsrow + 4 * q);
            h[2 * q] = (v2f){v.x, v.y}; h[2 * q + 1] = (v2f){v.z, v.w};
        }
    } else {
#pragma unroll
        for (int n = 0; n < NP / 2; ++n) h[n] = splat(0.0f);
    }
    float sum_dt = 0.0f;

    const float* ub = p.u + ((long long)b * p.dim + ch0) * L;
    const float* db = p.delta + ((long long)b * p.dim + ch0) * L;
    const float* Bb = p.Bm + ((long long)b * p.G + g) * p.N * L;
    const float* Cb = p.Cm + ((long long)b * p.G + g) * p.N * L;
    float* ob = (PHASE == 3) ? p.out + ((long long)b * p.dim + ch0) * L : nullptr;
    const float* zb = (PHASE == 3 && p.z) ? p.z + ((long long)b * p.dim + ch0) * L : nullptr;

    // ---- register staging of one tile (prefetched one tile ahead) ---------------------------
    constexpr int NBQ = (NP + kRpi - 1) / kRpi;         // float4 per lane for a [NP][kTile] tile
    float4 ru[kNld], rd[kNld], rB[NBQ], rC[NBQ];
    const int trow = lane / kLpr, tq = lane % kLpr;      // tile row within a slab, quad column

    auto fetch = [&](int t0) {
        if constexpr (VEC) {
            // UNCONDITIONAL loads from clamped offsets, masked when the tile is staged: an `ok ? load : 0` is a branch
            // around the load with a full wait behind it - the ten loads of a tile went out one at a time, each waiting
            // out its own latency (tools/isa_load_waits.py)
            const int t = t0 + 4 * tq;
            const bool tin = t < t_end;                  // L % 4 == 0: a quad is all-in or all-out
#pragma unroll
            for (int i = 0; i < kNld; ++i) {
                const int r = kRpi * i + trow;
                const long long off = (tin && r < nch) ? (long long)r * L + t : 0LL;
                ru[i] = *reinterpret_cast<const float4*>(ub + off);
                rd[i] = *reinterpret_cast<const float4*>(db + off);
            }
#pragma unroll
            for (int i = 0; i < NBQ; ++i) {
                const int n = kRpi * i + trow;
                const long long off = (tin && n < p.N) ? (long long)n * L + t : 0LL;
                rB[i] = *reinterpret_cast<const float4*>(Bb + off);
                if (PHASE == 3) rC[i] = *reinterpret_cast<const float4*>(Cb + off);
            }
        } else {
            // scalar path: same element->register mapping, element-wise guards
            auto ld4 = [&](const float* base, int r, int rmax, int t) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < rmax) {
                    const float* q = base + (long long)r * L;
                    if (t + 0 < t_end) v.x = q[t + 0];
                    if (t + 1 < t_end) v.y = q[t + 1];
                    if (t + 2 < t_end) v.z = q[t + 2];
                    if (t + 3 < t_end) v.w = q[t + 3];
                }
                return v;
            };
            const int t = t0 + 4 * tq;
#pragma unroll
            for (int i = 0; i < kNld; ++i) {
                ru[i] = ld4(ub, kRpi * i + trow, nch, t);
                rd[i] = ld4(db, kRpi * i + trow, nch, t);
            }
#pragma unroll
            for (int i = 0; i < NBQ; ++i) {
                rB[i] = ld4(Bb, kRpi * i + trow, p.N, t);
                if (PHASE == 3) rC[i] = ld4(Cb, kRpi * i + trow, p.N, t);
            }
        }
    };

    auto stage = [&](int t0) {    // registers -> LDS (u/delta row-major padded, B/C transposed); t0: the staged tile's first step
        const bool tin = !VEC || t0 + 4 * tq < t_end;
        // (component-wise selects: a ternary on the float4 STRUCT goes through memory and drags the staging registers
        // into scratch)
        auto sel4 = [](bool ok, const float4& v) { return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f); };
#pragma unroll
        for (int i = 0; i < kNld; ++i) {
            const int r = kRpi * i + trow;
            const bool ok = !VEC || (tin && r < nch);
            *reinterpret_cast<float4*>(&s_u[r * kRow + 4 * tq]) = sel4(ok, ru[i]);
            *reinterpret_cast<float4*>(&s_d[r * kRow + 4 * tq]) = sel4(ok, rd[i]);
        }
#pragma unroll
        for (int i = 0; i < NBQ; ++i) {
            const int n = kRpi * i + trow;
            if (n < NP) {
                const bool ok = !VEC || (tin && n < p.N);
                const float4 bq = sel4(ok, rB[i]);
                s_B[(4 * tq + 0) * NP + n] = bq.x; s_B[(4 * tq + 1) * NP + n] = bq.y;
                s_B[(4 * tq + 2) * NP + n] = bq.z; s_B[(4 * tq + 3) * NP + n] = bq.w;
                if (PHASE == 3) {
                    const float4 cq = sel4(ok, rC[i]);
                    s_C[(4 * tq + 0) * NP + n] = cq.x; s_C[(4 * tq + 1) * NP + n] = cq.y;
                    s_C[(4 * tq + 2) * NP + n] = cq.z; s_C[(4 * tq + 3) * NP + n] = cq.w;
                }
            }
        }
    };

    fetch(t_begin);
    for (int t0 = t_begin; t0 < t_end; t0 += kTile) {
        stage(t0);
        __syncthreads();
        if (t0 + kTile < t_end) fetch(t0 + kTile);      // in flight during the scan below
        const int tl = min(kTile, t_end - t0);           // wave-uniform

#pragma unroll
        for (int q = 0; q < kTile / 4; ++q) {
            if (4 * q < tl) {
                const float4 u4 = *reinterpret_cast<const float4*>(&s_u[lane * kRow + 4 * q]);
                const float4 d4 = *reinterpret_cast<const float4*>(&s_d[lane * kRow + 4 * q]);
                const float uu[4] = {u4.x, u4.y, u4.z, u4.w};
                v2f dta = (v2f){d4.x, d4.y} + bias, dtb = (v2f){d4.z, d4.w} + bias;
                if (p.softplus) { dta = softplus2(dta); dtb = softplus2(dtb); }
                const float dts[4] = {dta.x, dta.y, dtb.x, dtb.y};
                float yy[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int tt = 4 * q + j;
                    if (tt < tl) {
                        const float dt = dts[j];
                        const float ut = uu[j];
                        const v2f dt2 = splat(dt), du2 = splat(dt * ut);
                        if (PHASE == 1) sum_dt += dt;
                        v2f y2 = splat(0.0f);
#pragma unroll
                        for (int r = 0; r < NP / 4; ++r) {
                            const float4 bv = *reinterpret_cast<const float4*>(&s_B[tt * NP + 4 * r]);
                            const v2f a0 = exp2_2(dt2 * A2[2 * r]);
                            const v2f a1 = exp2_2(dt2 * A2[2 * r + 1]);
                            h[2 * r] = a0 * h[2 * r] + du2 * (v2f){bv.x, bv.y};
                            h[2 * r + 1] = a1 * h[2 * r + 1] + du2 * (v2f){bv.z, bv.w};
                            if (PHASE == 3) {
                                const float4 cv = *reinterpret_cast<const float4*>(&s_C[tt * NP + 4 * r]);
                                y2 = (v2f){cv.x, cv.y} * h[2 * r] + y2;
                                y2 = (v2f){cv.z, cv.w} * h[2 * r + 1] + y2;
                            }
                        }
                        if (PHASE == 3) yy[j] = fmaf(Dd, ut, y2.x + y2.y);
                    }
                }
                if (PHASE == 3)   // y overwrites the consumed u tile (same row, same lane)
                    *reinterpret_cast<float4*>(&s_u[lane * kRow + 4 * q]) =
                        make_float4(yy[0], yy[1], yy[2], yy[3]);
            }
        }
        __syncthreads();
        if (PHASE == 3) {         // LDS y tile -> global, 16 B per lane, z-gate fused
            const int t = t0 + 4 * tq;
#pragma unroll
            for (int i = 0; i < kNld; ++i) {
                const int r = kRpi * i + trow;
                if (r < nch && t < t_end) {
                    float4 v = *reinterpret_cast<const float4*>(&s_u[r * kRow + 4 * tq]);
                    float* o = ob + (long long)r * L + t;
                    if constexpr (VEC) {
                        if (zb) {
                            const float4 zz = *reinterpret_cast<const float4*>(zb + (long long)r * L + t);
                            v.x *= zz.x / (1.0f + expf(-zz.x)); v.y *= zz.y / (1.0f + expf(-zz.y));
                            v.z *= zz.z / (1.0f + expf(-zz.z)); v.w *= zz.w / (1.0f + expf(-zz.w));
                        }
                        *reinterpret_cast<float4*>(o) = v;
                    } else {
                        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (t + j < t_end) {
                                float val = vv[j];
                                if (zb) { const float zz = zb[(long long)r * L + t + j]; val *= zz / (1.0f + expf(-zz)); }
                                o[j] = val;
                            }
                    }
                }
            }
            __syncthreads();      // the y tile is read before the next stage() overwrites it
        }
    }

    if (PHASE == 1) {
        if (live) {
#pragma unroll
            for (int q = 0; q < NP / 4; ++q) {
                *reinterpret_cast<float4*>(p.wsH + wsrow + 4 * q) =
                    make_float4(h[2 * q].x, h[2 * q].y, h[2 * q + 1].x, h[2 * q + 1].y);
                const v2f p0 = exp2_2(splat(sum_dt) * A2[2 * q]), p1 = exp2_2(splat(sum_dt) * A2[2 * q + 1]);
                *reinterpret_cast<float4*>(p.wsP + wsrow + 4 * q) = make_float4(p0.x, p0.y, p1.x, p1.y);
            }
        }
    } else if (p.last_state && chunk == p.nchunks - 1 && live) {
        for (int n = 0; n < p.N; ++n)
            p.last_state[((long long)b * p.dim + d) * p.N + n] = (n & 1) ? h[n / 2].y : h[n / 2].x;
    }
}

// Phase 2.  chain = (b*dim + d)*NP + n; ws layout [chunk][chain] so that consecutive lanes touch
// consecutive words.  A 1024-thread block owns 16 chains x 64 segments of the chunk range (lane =
// 16 chains x 4 segments, 16 waves): each thread folds its segment (loads batched 8 deep so the
// dependent FMA chain never waits on memory), the 64 segment aggregates of a chain are combined
// through LDS, then each thread re-walks its segment replacing the end state H[c] by the carry-in.
constexpr int kCarrySeg = 64;
// Up to eight independent carry problems over the same chains in one launch (blockIdx.z): the fused SS2D core runs
// its four directions' carries together (forward: four; backward: four forward + four adjoint ones) - they are
// latency-bound launches of a few microseconds each.
struct CarryDir { float* wsP; float* wsH; float* segP; float* segH; int nchunks; int nsegs; };
struct CarryBatch { CarryDir d[8]; };

template <bool SEGS>      // SEGS: scan the per-segment aggregates (segP, segH) instead of the chunk summaries
__global__ __launch_bounds__(1024) void selscan_carry_kernel(CarryBatch cb, long long nchains) {
    const CarryDir cd = cb.d[blockIdx.z];
    const float* __restrict__ wsP = SEGS ? cd.segP : cd.wsP;
    float* __restrict__ wsH = SEGS ? cd.segH : cd.wsH;
    const int nchunks = SEGS ? cd.nsegs : cd.nchunks;
    __shared__ float sP[kCarrySeg][17];
    __shared__ float sH[kCarrySeg][17];
    const int cl = threadIdx.x & 15;                       // chain within the block
    const int seg = threadIdx.x >> 4;                      // 0..63
    const long long chain = (long long)blockIdx.x * 16 + cl;
    const bool ok = chain < nchains;
    const int per = (nchunks + kCarrySeg - 1) / kCarrySeg;
    const int c0 = min(nchunks, seg * per), c1 = min(nchunks, c0 + per);
    float P = 1.0f, H = 0.0f;
    // per <= 16 (every sequence of <= 1024 summaries: the case this kernel is launched for): the thread's summaries stay in
    // registers between the fold and the re-walk - read once instead of twice (the op is its traffic: 32 MB of summaries per
    // call of the fused core at UHD level 1, 35 us with the second read)
    const bool keep = per <= 16;                           // uniform
    float pk[16], hk[16];
    if (ok && keep) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const bool in = c0 + j < c1;
            pk[j] = in ? wsP[(long long)(c0 + j) * nchains + chain] : 1.0f;
            hk[j] = in ? wsH[(long long)(c0 + j) * nchains + chain] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) { H = fmaf(pk[j], H, hk[j]); P *= pk[j]; }
    } else if (ok) {
        for (int c = c0; c < c1; c += 8) {
            float pp[8], hh[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool in = c + j < c1;
                pp[j] = in ? wsP[(long long)(c + j) * nchains + chain] : 1.0f;
                hh[j] = in ? wsH[(long long)(c + j) * nchains + chain] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { H = fmaf(pp[j], H, hh[j]); P *= pp[j]; }
        }
    }
    sP[seg][cl] = P; sH[seg][cl] = H;
    __syncthreads();
    float carry = 0.0f;
    for (int s = 0; s < seg; ++s) carry = fmaf(sP[s][cl], carry, sH[s][cl]);
    if (ok && keep) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (c0 + j < c1) wsH[(long long)(c0 + j) * nchains + chain] = carry;
            carry = fmaf(pk[j], carry, hk[j]);
        }
    } else if (ok) {
        for (int c = c0; c < c1; c += 8) {
            float pp[8], hh[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool in = c + j < c1;
                pp[j] = in ? wsP[(long long)(c + j) * nchains + chain] : 1.0f;
                hh[j] = in ? wsH[(long long)(c + j) * nchains + chain] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (c + j < c1) wsH[(long long)(c + j) * nchains + chain] = carry;
                carry = fmaf(pp[j], carry, hh[j]);
            }
        }
    }
}

// Two-level carry for long chunk sequences (column scans have one chunk sequence entry per image
// column and segment: tens of thousands).  Thread = (chain, segment of kCarrySegLen chunks), fully
// independent, 256-B coalesced rows:
//   fold  : segment aggregate (P, H) -> segP/segH [segment][chain]
//   (selscan_carry_kernel on the aggregates turns segH into the per-segment carry-in)
//   apply : re-walk the segment from its carry-in, replacing H[c] by H_in[c]
constexpr int kCarrySegLen = 32;
template <bool APPLY>
__global__ __launch_bounds__(256) void selscan_carry_seg_kernel(CarryBatch cb, long long nchains) {
    const CarryDir cd = cb.d[blockIdx.z];
    const float* __restrict__ wsP = cd.wsP;
    float* __restrict__ wsH = cd.wsH;
    float* __restrict__ segP = cd.segP;
    float* __restrict__ segH = cd.segH;
    const int nchunks = cd.nchunks, nsegs = cd.nsegs;
    const long long chain = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    const int seg = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (chain >= nchains || seg >= nsegs) return;
    const int c0 = seg * kCarrySegLen, c1 = min(nchunks, c0 + kCarrySegLen);
    float P = 1.0f, H = APPLY ? segH[(long long)seg * nchains + chain] : 0.0f;
    for (int c = c0; c < c1; c += 8) {
        float pp[8], hh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool in = c + j < c1;
            pp[j] = in ? wsP[(long long)(c + j) * nchains + chain] : 1.0f;
            hh[j] = in ? wsH[(long long)(c + j) * nchains + chain] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (APPLY && c + j < c1) wsH[(long long)(c + j) * nchains + chain] = H;
            H = fmaf(pp[j], H, hh[j]);
            P *= pp[j];
        }
    }
    if (!APPLY) {
        segP[(long long)seg * nchains + chain] = P;
        segH[(long long)seg * nchains + chain] = H;
    }
}

}  // namespace wm
