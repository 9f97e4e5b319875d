/*
 * wavemamba_oracle.c - CPU restatement of the Wave-Mamba hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP kernels in wave-mamba_amd/csrc/.  Only tests/,
 * __graft_entry__.smoke() and the `cpu_baseline` leg of bench.py may load it; the product path
 * never does (it fails loudly when the HIP library is missing).
 *
 * Every function restates, in plain C (fp32 arithmetic, OpenMP over independent channels), one
 * function of the reference (all citations are into /root/reference/):
 *
 *   oracle_dwt2d_fwd      basicsr/archs/wavemamba_arch.py:97-110   dwt_init
 *   oracle_idwt2d_fwd     basicsr/archs/wavemamba_arch.py:113-130  iwt_init
 *   oracle_selscan_fwd    call site wavemamba_arch.py:465-471 (selective_scan_fn of mamba_ssm)
 *   oracle_selscan_bwd    autograd of the above (training, basicsr/models/femasr_model.py:181)
 *   oracle_ss2d_core_fwd  basicsr/archs/wavemamba_arch.py:446-478  SS2D.forward_core
 *
 * PINNING.  dwt/iwt/ss2d_core are pinned against outputs of the reference's own code run in the
 * build container (tests/golden/wavelet.npz, scan.npz; generator tests/golden/make_golden.py).
 * The selective scan itself lives in the third-party package `mamba_ssm` (reference
 * requirements.txt:16: un-vendored, NO pinned version), whose source is not under /root/reference
 * and for which the reference holds no test vectors: for that one function PARITY IS UNPINNED by
 * the reference.  The recurrence below restates the package's published `selective_scan_ref`
 * semantics
 *        delta' = softplus(delta + delta_bias)            (threshold 20, like F.softplus)
 *        h_t    = exp(delta'_t * A) * h_{t-1} + delta'_t * B_t * u_t ,  h_0 = 0
 *        y_t    = <C_t, h_t> + D * u_t ;  out = y * silu(z) when z is given
 * and is checked against a sequential PyTorch statement of the same semantics driven through the
 * reference's real SS2D.forward_core (the `stub_selective_scan` of make_golden.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* F.softplus(x) with beta = 1, threshold = 20 (torch semantics) */
static inline float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
static inline float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

/* ------------------------------------------------------------------------------------------ */
/* Haar analysis: wavemamba_arch.py:97-110.  x (B,C,H,W) -> four (B,C,H/2,W/2) sub-bands.      */
/* a = x[2i,2j]/2, b = x[2i+1,2j]/2, c = x[2i,2j+1]/2, d = x[2i+1,2j+1]/2 (x1..x4 of the ref)   */
/* ------------------------------------------------------------------------------------------ */
void oracle_dwt2d_fwd(const float* x, float* ll, float* hl, float* lh, float* hh,
                      int B, int C, int H, int W) {
    const int h = H / 2, w = W / 2;
    const long planes = (long)B * C;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < planes; ++p) {
        const float* xp = x + p * (long)H * W;
        float* o0 = ll + p * (long)h * w; float* o1 = hl + p * (long)h * w;
        float* o2 = lh + p * (long)h * w; float* o3 = hh + p * (long)h * w;
        for (int i = 0; i < h; ++i) {
            const float* r0 = xp + (long)(2 * i) * W;
            const float* r1 = r0 + W;
            for (int j = 0; j < w; ++j) {
                const float x1 = r0[2 * j] / 2, x2 = r1[2 * j] / 2;
                const float x3 = r0[2 * j + 1] / 2, x4 = r1[2 * j + 1] / 2;
                o0[(long)i * w + j] = x1 + x2 + x3 + x4;
                o1[(long)i * w + j] = -x1 - x2 + x3 + x4;
                o2[(long)i * w + j] = -x1 + x2 - x3 + x4;
                o3[(long)i * w + j] = x1 - x2 - x3 + x4;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Haar synthesis: wavemamba_arch.py:113-130.  x (B,4C,h,w) = [x1|x2|x3|x4] channel blocks      */
/* -> out (B,C,2h,2w), always fp32.                                                            */
/* ------------------------------------------------------------------------------------------ */
void oracle_idwt2d_fwd(const float* x, float* out, int B, int C, int h, int w) {
    const long hw = (long)h * w;
#pragma omp parallel for schedule(static) collapse(2)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const float* p1 = x + ((long)b * 4 * C + c) * hw;
            const float* p2 = p1 + (long)C * hw;
            const float* p3 = p2 + (long)C * hw;
            const float* p4 = p3 + (long)C * hw;
            float* o = out + ((long)b * C + c) * 4 * hw;
            for (int i = 0; i < h; ++i)
                for (int j = 0; j < w; ++j) {
                    const float x1 = p1[i * (long)w + j] / 2, x2 = p2[i * (long)w + j] / 2;
                    const float x3 = p3[i * (long)w + j] / 2, x4 = p4[i * (long)w + j] / 2;
                    float* q0 = o + (long)(2 * i) * (2 * w) + 2 * j;
                    float* q1 = q0 + 2 * w;
                    q0[0] = x1 - x2 - x3 + x4;
                    q1[0] = x1 - x2 + x3 - x4;
                    q0[1] = x1 + x2 - x3 - x4;
                    q1[1] = x1 + x2 + x3 + x4;
                }
        }
}

/* ------------------------------------------------------------------------------------------ */
/* Selective scan forward (operator boundary of wavemamba_arch.py:465-471).                     */
/* u, delta, z, out: (batch, dim, L);  A: (dim, N);  Bm, Cm: (batch, G, N, L), dim % G == 0,    */
/* channel d uses group d / (dim/G);  D, delta_bias: (dim) or NULL;  z NULL = no gate.          */
/* last_state (batch, dim, N) or NULL.                                                         */
/* ------------------------------------------------------------------------------------------ */
void oracle_selscan_fwd(const float* u, const float* delta, const float* A, const float* Bm,
                        const float* Cm, const float* D, const float* z, const float* delta_bias,
                        float* out, float* last_state,
                        int batch, int dim, int L, int N, int G, int delta_softplus) {
    const int dpg = dim / G;
#pragma omp parallel for schedule(static) collapse(2)
    for (int b = 0; b < batch; ++b)
        for (int d = 0; d < dim; ++d) {
            const long row = ((long)b * dim + d) * L;
            const float* Bg = Bm + ((long)b * G + d / dpg) * N * (long)L;
            const float* Cg = Cm + ((long)b * G + d / dpg) * N * (long)L;
            const float bias = delta_bias ? delta_bias[d] : 0.0f;
            float* h = (float*)calloc((size_t)N, sizeof(float));
            for (int t = 0; t < L; ++t) {
                float dt = delta[row + t] + bias;
                if (delta_softplus) dt = softplus_f(dt);
                const float ut = u[row + t];
                float y = 0.0f;
                for (int n = 0; n < N; ++n) {
                    h[n] = expf(dt * A[(long)d * N + n]) * h[n] + dt * Bg[(long)n * L + t] * ut;
                    y += h[n] * Cg[(long)n * L + t];
                }
                if (D) y += ut * D[d];
                if (z) { const float zt = z[row + t]; y *= zt * sigmoid_f(zt); }
                out[row + t] = y;
            }
            if (last_state) memcpy(last_state + ((long)b * dim + d) * N, h, (size_t)N * sizeof(float));
            free(h);
        }
}

/* ------------------------------------------------------------------------------------------ */
/* Selective scan backward (derived from the forward definition).       */
/* With a_t = exp(dt_t A), g_t = C_t dy_t + a_{t+1} g_{t+1} (adjoint state, g_{L} = 0):          */
/*   dC_t[n] += dy_t h_t[n]                    (summed over the channels of the group)          */
/*   dB_t[n] += g_t[n] dt_t u_t                (summed over the channels of the group)          */
/*   du_t     = dt_t sum_n g_t[n] B_t[n] + D dy_t                                               */
/*   ddt_t    = sum_n g_t[n] (A[n] a_t[n] h_{t-1}[n] + B_t[n] u_t)                              */
/*   dA[n]   += g_t[n] a_t[n] h_{t-1}[n] dt_t ;  dD += dy_t u_t                                 */
/*   ddelta_t = ddt_t * sigmoid(delta_t + bias) (softplus) ; dbias = sum_t ddelta_t             */
/* z-gating is not differentiated here (the reference never passes z, :467).                    */
/* dB, dC, dA, dD, dbias must be zero-initialised by the caller; they are accumulated.          */
/* ------------------------------------------------------------------------------------------ */
void oracle_selscan_bwd(const float* u, const float* delta, const float* A, const float* Bm,
                        const float* Cm, const float* D, const float* delta_bias, const float* dy,
                        float* du, float* ddelta, float* dA, float* dB, float* dC, float* dD,
                        float* dbias,
                        int batch, int dim, int L, int N, int G, int delta_softplus) {
    const int dpg = dim / G;
    /* groups are independent: parallelise over (b, g) so dB/dC need no atomics; dA/dD/dbias are
       reduced over the batch serially through per-(b) partial buffers */
    double* pA = (double*)calloc((size_t)batch * dim * N, sizeof(double));
    double* pD = (double*)calloc((size_t)batch * dim, sizeof(double));
    double* pb = (double*)calloc((size_t)batch * dim, sizeof(double));
#pragma omp parallel for schedule(dynamic) collapse(2)
    for (int b = 0; b < batch; ++b)
        for (int g = 0; g < G; ++g) {
            const float* Bg = Bm + ((long)b * G + g) * N * (long)L;
            const float* Cg = Cm + ((long)b * G + g) * N * (long)L;
            float* dBg = dB + ((long)b * G + g) * N * (long)L;
            float* dCg = dC + ((long)b * G + g) * N * (long)L;
            double* accB = (double*)calloc((size_t)N * L, sizeof(double));
            double* accC = (double*)calloc((size_t)N * L, sizeof(double));
            float* hs = (float*)malloc((size_t)(L + 1) * N * sizeof(float));
            float* dts = (float*)malloc((size_t)L * sizeof(float));
            float* gst = (float*)malloc((size_t)N * sizeof(float));
            for (int d = g * dpg; d < (g + 1) * dpg; ++d) {
                const long row = ((long)b * dim + d) * L;
                const float bias = delta_bias ? delta_bias[d] : 0.0f;
                const float* Ad = A + (long)d * N;
                /* forward recompute, keeping every state */
                for (int n = 0; n < N; ++n) hs[n] = 0.0f;
                for (int t = 0; t < L; ++t) {
                    float dt = delta[row + t] + bias;
                    if (delta_softplus) dt = softplus_f(dt);
                    dts[t] = dt;
                    for (int n = 0; n < N; ++n)
                        hs[(long)(t + 1) * N + n] = expf(dt * Ad[n]) * hs[(long)t * N + n]
                                                    + dt * Bg[(long)n * L + t] * u[row + t];
                }
                for (int n = 0; n < N; ++n) gst[n] = 0.0f;
                double accD = 0.0, accb = 0.0;
                for (int t = L - 1; t >= 0; --t) {
                    const float dt = dts[t], ut = u[row + t], dyt = dy[row + t];
                    float s_du = 0.0f, s_dt = 0.0f;
                    for (int n = 0; n < N; ++n) {
                        const float a = expf(dt * Ad[n]);
                        const float hprev = hs[(long)t * N + n], hcur = hs[(long)(t + 1) * N + n];
                        /* gst currently holds a_{t+1} (.) g_{t+1} */
                        const float gt = Cg[(long)n * L + t] * dyt + gst[n];
                        accC[(long)n * L + t] += (double)dyt * hcur;
                        accB[(long)n * L + t] += (double)gt * dt * ut;
                        s_du += gt * Bg[(long)n * L + t];
                        s_dt += gt * (Ad[n] * a * hprev + Bg[(long)n * L + t] * ut);
                        pA[((long)b * dim + d) * N + n] += (double)gt * a * hprev * dt;
                        gst[n] = a * gt;
                    }
                    du[row + t] = dt * s_du + (D ? D[d] * dyt : 0.0f);
                    accD += (double)dyt * ut;
                    const float dd = delta_softplus
                        ? ((delta[row + t] + bias) > 20.0f ? s_dt : s_dt * sigmoid_f(delta[row + t] + bias))
                        : s_dt;
                    ddelta[row + t] = dd;
                    accb += dd;
                }
                pD[(long)b * dim + d] = accD;
                pb[(long)b * dim + d] = accb;
            }
            for (long i = 0; i < (long)N * L; ++i) { dBg[i] += (float)accB[i]; dCg[i] += (float)accC[i]; }
            free(accB); free(accC); free(hs); free(dts); free(gst);
        }
    for (int d = 0; d < dim; ++d) {
        double sD = 0.0, sb = 0.0;
        for (int b = 0; b < batch; ++b) { sD += pD[(long)b * dim + d]; sb += pb[(long)b * dim + d]; }
        if (dD) dD[d] += (float)sD;
        if (dbias) dbias[d] += (float)sb;
        for (int n = 0; n < N; ++n) {
            double sA = 0.0;
            for (int b = 0; b < batch; ++b) sA += pA[((long)b * dim + d) * N + n];
            dA[(long)d * N + n] += (float)sA;
        }
    }
    free(pA); free(pD); free(pb);
}

/* ------------------------------------------------------------------------------------------ */
/* SS2D.forward_core: wavemamba_arch.py:446-478.                                                */
/* x (B,D,H,W) -> y (4,B,D,L) = the four returned tensors (row-major fwd, row-major reversed,     */
/* column-major fwd, column-major reversed), all indexed in row-major l = h*W + w.               */
/* Direction k: 0 row-major, 1 column-major (l = w*H + h), 2 = flip(0), 3 = flip(1)  (:451-452)  */
/* x_dbl[k,c,l] = sum_d xs[k,d,l] Wx[k,c,d], c split [R | N | N] -> dts_r, Bs, Cs   (:453-454)   */
/* dts[k,d,l]   = sum_r dts_r[k,r,l] Wdt[k,d,r]   (bias passed separately, :455,:468)            */
/* As = -exp(A_logs) (:462); scan with softplus (:465-471); merge back to row-major (:474-478).  */
/* ------------------------------------------------------------------------------------------ */
static inline long dir_pos(int k, long l, int H, int W) {
    const long L = (long)H * W;
    if (k >= 2) l = L - 1 - l;
    if (k & 1) { const long wq = l / H, hq = l % H; return hq * W + wq; }   /* column-major */
    return l;
}

void oracle_ss2d_core_fwd(const float* x, const float* Wx, const float* Wdt, const float* dt_bias,
                          const float* A_logs, const float* Ds, float* y,
                          int B, int D, int H, int W, int N, int R) {
    const long L = (long)H * W;
    const int Cc = R + 2 * N;
#pragma omp parallel for schedule(dynamic) collapse(2)
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < 4; ++k) {
            const float* xb = x + (long)b * D * L;
            float* xdbl = (float*)malloc((size_t)Cc * L * sizeof(float));   /* [c][l] in scan order */
            float* us = (float*)malloc((size_t)D * L * sizeof(float));      /* xs[k] in scan order */
            for (long l = 0; l < L; ++l) {
                const long p = dir_pos(k, l, H, W);
                for (int d = 0; d < D; ++d) us[(long)d * L + l] = xb[(long)d * L + p];
            }
            for (int c = 0; c < Cc; ++c)
                for (long l = 0; l < L; ++l) {
                    float s = 0.0f;
                    for (int d = 0; d < D; ++d) s += us[(long)d * L + l] * Wx[((long)k * Cc + c) * D + d];
                    xdbl[(long)c * L + l] = s;
                }
            float* h = (float*)malloc((size_t)N * sizeof(float));
            for (int d = 0; d < D; ++d) {
                const int kd = k * D + d;
                for (int n = 0; n < N; ++n) h[n] = 0.0f;
                /* return order of :478 is (k0, flip k2, k1, flip k3): slot 1 <-> direction 2 */
                const int slot = (k == 1) ? 2 : (k == 2) ? 1 : k;
                float* yo = y + (((long)slot * B + b) * D + d) * L;
                for (long l = 0; l < L; ++l) {
                    float dt = 0.0f;
                    for (int r = 0; r < R; ++r) dt += xdbl[(long)r * L + l] * Wdt[((long)k * D + d) * R + r];
                    dt = softplus_f(dt + dt_bias[kd]);
                    const float ut = us[(long)d * L + l];
                    float acc = 0.0f;
                    for (int n = 0; n < N; ++n) {
                        const float An = -expf(A_logs[(long)kd * N + n]);
                        h[n] = expf(dt * An) * h[n] + dt * xdbl[(long)(R + n) * L + l] * ut;
                        acc += h[n] * xdbl[(long)(R + N + n) * L + l];
                    }
                    yo[dir_pos(k, l, H, W)] = acc + ut * Ds[kd];
                }
            }
            free(h); free(us); free(xdbl);
        }
}
