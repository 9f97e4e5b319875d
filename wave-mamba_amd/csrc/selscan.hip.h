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

constexpr int kTile = 16;   // time steps per LDS tile
constexpr int kRow = 20;    // LDS row stride (floats) of the [64][kTile] tiles: 80 B keeps the
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

__device__ __forceinline__ float softplus_f(float x) {
    // F.softplus(beta=1, threshold=20): log1p(exp(x)) below the threshold.  Hardware v_exp_f32 /
    // v_log_f32 (both base 2, ~1 ulp) with the w = 1 + e compensation for log1p:
    //   log1p(e) = log(w) * e / (w - 1), exact-cancelling the rounding of 1 + e.
    const float e = __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
    const float w = 1.0f + e;
    const float lp = (w == 1.0f) ? e
                                 : (__builtin_amdgcn_logf(w) * 0.6931471805599453f) * (e / (w - 1.0f));
    return x > 20.0f ? x : lp;
}

// PHASE 1: reduce (no C, no y; writes P/H).  PHASE 3: scan (reads H_in, writes y).
// NP = N padded to 16 or 32 (padded states have A = 0, B = C = 0 -> stay exactly 0).
// VEC: L % 4 == 0 and 16-byte aligned bases -> float4 global access; else scalar access.
template <int NP, int PHASE, bool VEC>
__global__ __launch_bounds__(64) void selscan_chunk_kernel(ScanArgs p) {
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
    float A2[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n)
        A2[n] = (n < p.N) ? p.A[(long long)d * p.N + n] * 1.4426950408889634f : 0.0f;
    const float bias = p.bias ? p.bias[d] : 0.0f;
    const float Dd = (PHASE == 3 && p.D) ? p.D[d] : 0.0f;

    float h[NP];
    const long long wsrow = ((long long)chunk * p.batch * p.dim + (long long)b * p.dim + d) * NP;
    if (PHASE == 3 && chunk > 0) {
#pragma unroll
        for (int q = 0; q < NP / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(p.wsH + wsrow + 4 * q);
            h[4 * q] = v.x; h[4 * q + 1] = v.y; h[4 * q + 2] = v.z; h[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int n = 0; n < NP; ++n) h[n] = 0.0f;
    }
    float sum_dt = 0.0f;

    const float* ub = p.u + ((long long)b * p.dim + ch0) * L;
    const float* db = p.delta + ((long long)b * p.dim + ch0) * L;
    const float* Bb = p.Bm + ((long long)b * p.G + g) * p.N * L;
    const float* Cb = p.Cm + ((long long)b * p.G + g) * p.N * L;
    float* ob = (PHASE == 3) ? p.out + ((long long)b * p.dim + ch0) * L : nullptr;
    const float* zb = (PHASE == 3 && p.z) ? p.z + ((long long)b * p.dim + ch0) * L : nullptr;

    // ---- register staging of one tile (prefetched one tile ahead) ---------------------------
    constexpr int NBQ = (NP * kTile / 4 + 63) / 64;     // float4 per lane for a [NP][16] tile
    float4 ru[4], rd[4], rB[NBQ], rC[NBQ];
    const int trow = lane >> 2, tq = lane & 3;           // tile row within a 16-row slab, quad col

    auto fetch = [&](int t0) {
        if constexpr (VEC) {
            const int t = t0 + 4 * tq;
            const bool tin = t < t_end;                  // L % 4 == 0: a quad is all-in or all-out
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * i + trow;
                const bool ok = tin && r < nch;
                ru[i] = ok ? *reinterpret_cast<const float4*>(ub + (long long)r * L + t)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
                rd[i] = ok ? *reinterpret_cast<const float4*>(db + (long long)r * L + t)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < NBQ; ++i) {
                const int n = 16 * i + trow;
                const bool ok = tin && n < p.N;
                rB[i] = ok ? *reinterpret_cast<const float4*>(Bb + (long long)n * L + t)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
                if (PHASE == 3)
                    rC[i] = ok ? *reinterpret_cast<const float4*>(Cb + (long long)n * L + t)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
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
            for (int i = 0; i < 4; ++i) {
                ru[i] = ld4(ub, 16 * i + trow, nch, t);
                rd[i] = ld4(db, 16 * i + trow, nch, t);
            }
#pragma unroll
            for (int i = 0; i < NBQ; ++i) {
                rB[i] = ld4(Bb, 16 * i + trow, p.N, t);
                if (PHASE == 3) rC[i] = ld4(Cb, 16 * i + trow, p.N, t);
            }
        }
    };

    auto stage = [&]() {          // registers -> LDS (u/delta row-major padded, B/C transposed)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 16 * i + trow;
            *reinterpret_cast<float4*>(&s_u[r * kRow + 4 * tq]) = ru[i];
            *reinterpret_cast<float4*>(&s_d[r * kRow + 4 * tq]) = rd[i];
        }
#pragma unroll
        for (int i = 0; i < NBQ; ++i) {
            const int n = 16 * i + trow;
            if (n < NP) {
                s_B[(4 * tq + 0) * NP + n] = rB[i].x; s_B[(4 * tq + 1) * NP + n] = rB[i].y;
                s_B[(4 * tq + 2) * NP + n] = rB[i].z; s_B[(4 * tq + 3) * NP + n] = rB[i].w;
                if (PHASE == 3) {
                    s_C[(4 * tq + 0) * NP + n] = rC[i].x; s_C[(4 * tq + 1) * NP + n] = rC[i].y;
                    s_C[(4 * tq + 2) * NP + n] = rC[i].z; s_C[(4 * tq + 3) * NP + n] = rC[i].w;
                }
            }
        }
    };

    fetch(t_begin);
    for (int t0 = t_begin; t0 < t_end; t0 += kTile) {
        stage();
        __syncthreads();
        if (t0 + kTile < t_end) fetch(t0 + kTile);      // in flight during the scan below
        const int tl = min(kTile, t_end - t0);           // wave-uniform

#pragma unroll
        for (int q = 0; q < kTile / 4; ++q) {
            if (4 * q < tl) {
                const float4 u4 = *reinterpret_cast<const float4*>(&s_u[lane * kRow + 4 * q]);
                const float4 d4 = *reinterpret_cast<const float4*>(&s_d[lane * kRow + 4 * q]);
                const float uu[4] = {u4.x, u4.y, u4.z, u4.w};
                const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
                float yy[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int tt = 4 * q + j;
                    if (tt < tl) {
                        float dt = dd[j] + bias;
                        if (p.softplus) dt = softplus_f(dt);
                        const float ut = uu[j];
                        const float du = dt * ut;
                        if (PHASE == 1) sum_dt += dt;
                        float Bv[NP], Cv[NP];
#pragma unroll
                        for (int r = 0; r < NP / 4; ++r) {
                            const float4 v = *reinterpret_cast<const float4*>(&s_B[tt * NP + 4 * r]);
                            Bv[4 * r] = v.x; Bv[4 * r + 1] = v.y; Bv[4 * r + 2] = v.z; Bv[4 * r + 3] = v.w;
                            if (PHASE == 3) {
                                const float4 c = *reinterpret_cast<const float4*>(&s_C[tt * NP + 4 * r]);
                                Cv[4 * r] = c.x; Cv[4 * r + 1] = c.y; Cv[4 * r + 2] = c.z; Cv[4 * r + 3] = c.w;
                            }
                        }
                        float y = 0.0f;
#pragma unroll
                        for (int n = 0; n < NP; ++n) {
                            const float a = __builtin_amdgcn_exp2f(dt * A2[n]);
                            h[n] = fmaf(a, h[n], du * Bv[n]);
                            if (PHASE == 3) y = fmaf(Cv[n], h[n], y);
                        }
                        if (PHASE == 3) yy[j] = fmaf(Dd, ut, y);
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
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * i + trow;
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
                    make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
                *reinterpret_cast<float4*>(p.wsP + wsrow + 4 * q) =
                    make_float4(__builtin_amdgcn_exp2f(sum_dt * A2[4 * q]),
                                __builtin_amdgcn_exp2f(sum_dt * A2[4 * q + 1]),
                                __builtin_amdgcn_exp2f(sum_dt * A2[4 * q + 2]),
                                __builtin_amdgcn_exp2f(sum_dt * A2[4 * q + 3]));
            }
        }
    } else if (p.last_state && chunk == p.nchunks - 1 && live) {
        for (int n = 0; n < p.N; ++n) p.last_state[((long long)b * p.dim + d) * p.N + n] = h[n];
    }
}

// Phase 2.  chain = (b*dim + d)*NP + n; ws layout [chunk][chain] so that consecutive lanes touch
// consecutive words.  A 1024-thread block owns 64 chains; its 16 waves split the chunk range:
// each wave folds its segment, the segment aggregates are combined through LDS, then each wave
// re-walks its segment replacing the end state H[c] by the carry-in H_in[c].
__global__ __launch_bounds__(1024) void selscan_carry_kernel(const float* __restrict__ wsP,
                                                             float* __restrict__ wsH,
                                                             long long nchains, int nchunks) {
    __shared__ float sP[16][64];
    __shared__ float sH[16][64];
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const long long chain = (long long)blockIdx.x * 64 + lane;
    const bool ok = chain < nchains;
    const int per = (nchunks + 15) / 16;
    const int c0 = min(nchunks, seg * per), c1 = min(nchunks, c0 + per);
    float P = 1.0f, H = 0.0f;
    if (ok) {
#pragma unroll 4
        for (int c = c0; c < c1; ++c) {
            const float pp = wsP[(long long)c * nchains + chain];
            const float hh = wsH[(long long)c * nchains + chain];
            H = fmaf(pp, H, hh);
            P *= pp;
        }
    }
    sP[seg][lane] = P; sH[seg][lane] = H;
    __syncthreads();
    float carry = 0.0f;
    for (int s = 0; s < seg; ++s) carry = fmaf(sP[s][lane], carry, sH[s][lane]);
    if (ok) {
#pragma unroll 4
        for (int c = c0; c < c1; ++c) {
            const float pp = wsP[(long long)c * nchains + chain];
            const float hh = wsH[(long long)c * nchains + chain];
            wsH[(long long)c * nchains + chain] = carry;
            carry = fmaf(pp, carry, hh);
        }
    }
}

}  // namespace wm
