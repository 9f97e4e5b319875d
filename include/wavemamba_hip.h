/*
 * wavemamba_hip.h - C ABI of libwavemamba_hip.so, the MI355X (gfx950) implementation of the
 * Wave-Mamba hot path: Haar DWT/IWT over NCHW feature maps, the selective-scan recurrence, and the
 * SS2D four-direction scan core.
 *
 * The reference (AlexZou14/Wave-Mamba, citations into /root/reference/) is pure Python; its only
 * FFI on this path is the pybind module `selective_scan_cuda.{fwd,bwd}` of the third-party package
 * `mamba_ssm`, reached through `selective_scan_fn` (basicsr/archs/wavemamba_arch.py:6, :383,
 * :465-471).  Each entry point below names the reference interface it replaces.  The Python host
 * side (wave_mamba_amd/ops.py) binds these symbols with ctypes and mirrors the reference's
 * operator surface (same names, argument meaning, error behaviour).
 *
 * Conventions
 *   - Plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise.
 *   - Ownership: the caller allocates every input, output and workspace buffer.  The library never
 *     allocates or frees device memory and keeps no state between calls (profiling hooks aside).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls only enqueue work;
 *     they never synchronise the host.
 *   - Return value: WM_OK (0) on success; a negative WM_E* code for argument errors; a positive
 *     value is a hipError_t from the launch.  wm_strerror() renders either.
 *   - Tensors are dense, row-major ("contiguous") in the stated shape unless a stride is given.
 *   - dtype codes: WM_F32 = 0 (float), WM_BF16 = 1 (bfloat16 storage, fp32 arithmetic unless noted).
 */
#ifndef WAVEMAMBA_HIP_H
#define WAVEMAMBA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WM_OK 0
#define WM_EINVAL (-1)      /* bad shape / size argument */
#define WM_ENULL (-2)       /* required pointer is NULL */
#define WM_EALIGN (-3)      /* pointer not aligned to the element size */
#define WM_EWORKSPACE (-4)  /* workspace too small */
#define WM_EUNSUPPORTED (-5)/* argument combination not implemented */
#define WM_EHIP (-6)        /* a HIP runtime call made on behalf of the launch failed (device query, function attribute) */

#define WM_F32 0
#define WM_BF16 1

/* ABI version of this header (bumped on any signature change). */
int wm_abi_version(void);
/* Identity of the sources this binary was compiled from: the first 16 hex digits of a sha256 over csrc/ and this header
 * (wave_mamba_amd/build.py passes it as -DWM_BUILD_ID; "unknown" for a bare hipcc build).  Measurements that describe
 * one binary (profiles/pmc_traffic.json) carry it, and bench.py quotes them only for the binary that is loaded. */
const char* wm_build_id(void);
const char* wm_strerror(int code);

/* --------------------------------------------------------------------------------------------
 * Haar DWT.  Replaces dwt_init(x) -> (x_LL, x_HL, x_LH, x_HH), wavemamba_arch.py:97-110
 * (DWT.forward :138-139, called from DownFRG.forward :973).
 *   x  (B, C, H, W)  H, W even   ->  ll, hl, lh, hh  each (B, C, H/2, W/2), same dtype as x.
 * bf16: every add rounds to bf16 like the reference's eager bf16 tensor arithmetic.
 * wm_dwt2d_bwd: gradient of the above; (dll, dhl, dlh, dhh) -> dx (B, C, H, W).
 * -------------------------------------------------------------------------------------------- */
int wm_dwt2d_fwd(const void* x, void* ll, void* hl, void* lh, void* hh,
                 int B, int C, int H, int W, int dtype, void* stream);
int wm_dwt2d_bwd(const void* dll, const void* dhl, const void* dlh, const void* dhh, void* dx,
                 int B, int C, int H, int W, int dtype, void* stream);

/* --------------------------------------------------------------------------------------------
 * Haar IWT.  Replaces iwt_init(x) -> h, wavemamba_arch.py:113-130 (IWT.forward :147-148, called
 * from upFRG.forward :1006 on cat([x_l, h_out_conv(x_h)], dim=1)).
 * The reference takes one (B, 4C, h, w) tensor whose channel blocks are [x1 | x2 | x3 | x4].  Here
 * each block is its own base pointer with its own batch stride (in elements), so the caller can
 * pass the concatenated tensor (x_k = x + k*C*h*w, stride 4*C*h*w) or the un-concatenated pair
 * (x1 = x_l, stride C*h*w;  x2..x4 = x_h + {0,1,2}*C*h*w, stride 3*C*h*w) without a copy.
 *   out (B, C, 2h, 2w) is ALWAYS fp32 (reference :122-123), whatever `dtype` the inputs have.
 * wm_idwt2d_bwd: dout (B, C, 2h, 2w) fp32 -> d1..d4 (same addressing, element type `dtype`).
 * -------------------------------------------------------------------------------------------- */
int wm_idwt2d_fwd(const void* x1, const void* x2, const void* x3, const void* x4,
                  int64_t bs1, int64_t bs2, int64_t bs3, int64_t bs4,
                  float* out, int B, int C, int h, int w, int dtype, void* stream);
int wm_idwt2d_bwd(const float* dout, void* d1, void* d2, void* d3, void* d4,
                  int64_t bs1, int64_t bs2, int64_t bs3, int64_t bs4,
                  int B, int C, int h, int w, int dtype, void* stream);

/* --------------------------------------------------------------------------------------------
 * Selective scan.  Replaces mamba_ssm's
 *   selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
 *                     return_last_state=False)
 * as called at wavemamba_arch.py:465-471 (fp32 in / out / state, asserted at :472).
 *   u, delta, z, out : (batch, dim, L) fp32          A : (dim, N) fp32
 *   Bm, Cm           : (batch, G, N, L) fp32, dim % G == 0, channel d reads group d / (dim / G)
 *   D, delta_bias    : (dim) fp32 or NULL            z : NULL = no gate (out = y), else y*silu(z)
 *   last_state       : (batch, dim, N) fp32 or NULL
 *   1 <= N <= 32.
 *   delta' = softplus(delta + delta_bias) if delta_softplus else delta + delta_bias
 *   h_t = exp(delta'_t A) h_{t-1} + delta'_t B_t u_t ;  y_t = <C_t, h_t> + D u_t
 * The sequence is split into chunks scanned in parallel (local reduce -> carry scan -> local scan);
 * `workspace` holds the per-chunk (decay product, end state) pairs; query its size first.
 * -------------------------------------------------------------------------------------------- */
size_t wm_selscan_fwd_workspace_bytes(int batch, int dim, int L, int N, int G);
int wm_selscan_fwd(const float* u, const float* delta, const float* A, const float* Bm,
                   const float* Cm, const float* D, const float* z, const float* delta_bias,
                   float* out, float* last_state, void* workspace, size_t workspace_bytes,
                   int batch, int dim, int L, int N, int G, int delta_softplus, void* stream);

/* Backward of the above (mamba_ssm's selective_scan_cuda.bwd, reached by autograd in training,
 * basicsr/models/femasr_model.py:181).  z is not supported (the reference passes z=None, :467).
 *   dy (batch, dim, L) -> du, ddelta (batch, dim, L); dB, dC (batch, G, N, L);
 *   dA (dim, N), dD (dim) or NULL, dbias (dim) or NULL.  All outputs are OVERWRITTEN.
 */
size_t wm_selscan_bwd_workspace_bytes(int batch, int dim, int L, int N, int G);
int wm_selscan_bwd(const float* u, const float* delta, const float* A, const float* Bm,
                   const float* Cm, const float* D, const float* delta_bias, const float* dy,
                   float* du, float* ddelta, float* dA, float* dB, float* dC, float* dD,
                   float* dbias, void* workspace, size_t workspace_bytes,
                   int batch, int dim, int L, int N, int G, int delta_softplus, void* stream);

/* --------------------------------------------------------------------------------------------
 * Fused SS2D four-direction scan core.  Replaces SS2D.forward_core(x), wavemamba_arch.py:446-478:
 * direction flatten (:451-452), x_proj / dt_proj einsums (:453-455), the selective scan (:465-471)
 * and the flips / transposes back (:474-478), without materialising any of the intermediates.
 *   x (B, D, H, W) fp32;  x_proj_weight (4, R + 2N, D);  dt_projs_weight (4, D, R);
 *   dt_projs_bias (4, D);  A_logs (4 D, N)  [A = -exp(A_logs)];  Ds (4 D)
 *   outputs, each (B, D, H*W) in row-major l, in the reference's return order (:478):
 *     y_row_fwd = out_y[:,0], y_row_rev = flip(out_y[:,2]), y_col_fwd, y_col_rev (column scans,
 *     transposed back).  merged != 0: only y_row_fwd is written and holds the SUM of the four
 *     (what SS2D.forward computes next, :490, added in the reference's order); the other three
 *     pointers may be NULL.  wm_lfss_mid_fwd can add the four un-merged outputs while it reads them.
 *     merged must be 0 or 1 (2 was round 4's two-plane mode - measured slower, profiles/r04/core_ab_paired_planes.txt - deleted).
 *   Supported: N <= 32, R <= 4, D <= 64, D * H * W < 2^31 (else WM_EUNSUPPORTED: use wm_selscan_fwd); any H, W.
 *   W % 4 == 0 with 16-byte aligned x / y buffers: 16-byte tile accesses; otherwise (odd widths, fp32 planes only)
 *   the same kernels with element-wise tile accesses, slower per position.
 *   The workspace size depends on `merged` (three temporary y buffers).
 */
size_t wm_ss2d_core_fwd_workspace_bytes(int B, int D, int H, int W, int N, int R, int merged);
/*   Host-only: the work split wm_ss2d_core_fwd uses for this shape (no GPU needed; tests and tools).
 *   out[0..9] = waves per workgroup, rows per column segment, segments per column, column tiles, column workgroup slots
 *   per direction, steps per row chunk, row chunks, row workgroups per direction, workgroups per launch, estimated
 *   launch length in row-tile times x 100 under the dispatch model (wavemamba_hip.hip: core_makespan).
 */
int wm_ss2d_core_plan(int B, int D, int H, int W, int N, int R, int* out10);
/*   plane_dtype: storage type of x and of the y buffers, WM_F32 or WM_BF16 (bf16-storage mode: bf16 planes between
 *   the kernels, fp32 tiles / projection / state inside; needs W % 4 == 0 and 16-byte aligned planes: WM_EUNSUPPORTED /
 *   WM_EALIGN otherwise).
 */
/*   prepared: NULL, or the buffer wm_ss2d_core_prep filled from THESE parameters (wm_ss2d_core_prep_bytes(N) bytes,
 *   16-byte aligned): the bf16-split x_proj weight fragments, A * log2(e) and the per-channel constants the kernels
 *   read.  With NULL the call prepares them itself into the workspace (one more small launch); inference callers whose
 *   parameters do not change keep one prepared buffer per SS2D module.
 */
int wm_ss2d_core_fwd(const void* x, const float* x_proj_weight, const float* dt_projs_weight,
                     const float* dt_projs_bias, const float* A_logs, const float* Ds,
                     void* y_row_fwd, void* y_row_rev, void* y_col_fwd, void* y_col_rev,
                     int merged, void* workspace, size_t workspace_bytes, const void* prepared,
                     int B, int D, int H, int W, int N, int R, int plane_dtype, void* stream);
size_t wm_ss2d_core_prep_bytes(int N);
int wm_ss2d_core_prep(const float* x_proj_weight, const float* dt_projs_weight, const float* dt_projs_bias,
                      const float* A_logs, const float* Ds, void* prepared, int D, int N, int R, void* stream);

/* --------------------------------------------------------------------------------------------
 * Depth-wise 3x3 convolution, stride 1, zero padding 1, + bias, + optional SiLU.
 * Replaces nn.Conv2d(groups=channels) + nn.SiLU of SS2D (wavemamba_arch.py:346-355, :487) and the
 * ffn's conv2 (:220, :226) in LFSSBlock.  x, y (B, C, H, W) fp32; weight (C, 1, 3, 3); bias (C) or
 * NULL; act: 0 = none, 1 = SiLU, 2 = GELU (exact erf form), + 4: the nine taps of every channel rotated by 180 degrees
 * (weight.flip(2, 3) without the flipped copy: the input gradient of the same convolution is act = 4 on gy).
 * plane_dtype: storage type of x and y (WM_F32 / WM_BF16, fp32 arithmetic).  Forward only.
 * -------------------------------------------------------------------------------------------- */
int wm_dwconv3x3_fwd(const void* x, const float* weight, const float* bias, void* y,
                     int B, int C, int H, int W, int act, int plane_dtype, void* stream);

/* --------------------------------------------------------------------------------------------
 * LFSSBlock / SS2D per-position glue around the scan core, fused (forward only; C in {8,16,32},
 * inner width D = 2C, ffn hidden = 2C).  Token tensors are (B, L, C) rows unless the *_nchw flag
 * says (B, C, L); plane tensors are (B, D, L).  Reference lines: wavemamba_arch.py
 *   wm_lfss_in_fwd  : ln_1 (:524) + in_proj (:483) + chunk (:484) + NHWC->NCHW (:486)
 *                     tok -> x (conv input), z (gate)
 *   wm_lfss_mid_fwd : transpose (:491) + out_norm (:492) + *silu(z) (:493) + out_proj (:494)
 *                     + skip_scale residual (:525) + ln_2 + ffn.conv1 (:526, :226)
 *                     ysum, z, tok -> tok1, f (input of ffn.conv2).  ny = 1: `ysum` is the merged core output;
 *                     ny = 4: `ysum` points at four (B, D, L) buffers `ystride` ELEMENTS of the plane dtype apart in the order
 *                     [y_row_fwd, y_row_rev, y_col_fwd, y_col_rev] and the kernel adds them (:490) as it loads
 *   wm_lfss_out_fwd : gelu gate (:227-228) + ffn.conv3 (:230) + skip_scale2 residual (:526)
 *                     fc, tok1 -> out
 * -------------------------------------------------------------------------------------------- */
/* plane_dtype: storage type of the plane tensors x, z, ysum, f, fc (WM_F32; WM_BF16 at C = 32 - the bf16-storage
 * mode); token tensors are always fp32. */
int wm_lfss_in_fwd(const float* tok, int tok_nchw, const float* ln_w, const float* ln_b, float ln_eps,
                   const float* in_proj_weight, void* x, void* z, int B, int64_t L, int C, int plane_dtype, void* stream);
int wm_lfss_mid_fwd(const void* ysum, int ny, int64_t ystride, const void* z, const float* tok, int tok_nchw,
                    const float* out_norm_w, const float* out_norm_b, float out_norm_eps,
                    const float* out_proj_weight, const float* skip_scale,
                    const float* ln2_w, const float* ln2_b, float ln2_eps,
                    const float* conv1_weight, const float* conv1_bias,
                    float* tok1, void* f, int B, int64_t L, int C, int plane_dtype, void* stream);
/* wm_lfss_mid_fwd with the gate RECOMPUTED from `tok` instead of read: z = in_proj(ln_1(tok))[..., D:2D] (:485-486) on the
 * matrix cores, in the matrix-instruction order of wm_lfss_in_fwd (fp32 planes: bit-identical results).  With it
 * wm_lfss_in_fwd is called with z = NULL (the gate half is never written): 512 fewer bytes per position of the block's 3456.
 * C == 32 only (WM_EUNSUPPORTED otherwise). */
int wm_lfss_mid_rz_fwd(const void* ysum, int ny, int64_t ystride, const float* tok, int tok_nchw, const float* ln1_w,
                       const float* ln1_b, float ln1_eps, const float* in_proj_weight, const float* out_norm_w,
                       const float* out_norm_b, float out_norm_eps, const float* out_proj_weight,
                       const float* skip_scale, const float* ln2_w, const float* ln2_b, float ln2_eps,
                       const float* conv1_weight, const float* conv1_bias, float* tok1, void* f, int B, int64_t L,
                       int C, int plane_dtype, void* stream);
int wm_lfss_out_fwd(const void* fc, const float* tok1, const float* conv3_weight, const float* conv3_bias,
                    const float* skip_scale2, float* out, int out_nchw, int B, int64_t L, int C, int plane_dtype,
                    void* stream);
/* wm_lfss_out_fwd with the gated ffn's depth-wise 3x3 (conv2, wavemamba_arch.py:220, :226) folded in: takes conv1's
 * output planes f (B, 2C, H, W) instead of conv2's, so `fc` never exists in HBM (-512 B per position, one launch less
 * per LFSSBlock).  Bit-identical to wm_dwconv3x3_fwd(act = none) + wm_lfss_out_fwd on fp32 planes.
 * C == 32 and W % 32 == 0 only (WM_EUNSUPPORTED otherwise: use the two calls).  conv2_bias may be NULL. */
int wm_lfss_out_conv_fwd(const void* f, const float* conv2_weight, const float* conv2_bias, const float* tok1,
                         const float* conv3_weight, const float* conv3_bias, const float* skip_scale2, float* out,
                         int out_nchw, int B, int H, int W, int C, int plane_dtype, void* stream);

/* --------------------------------------------------------------------------------------------
 * LayerNorm2d of the HFE branch (first "next" row): per-pixel LayerNorm over the C
 * channels of an NCHW map, eps inside the square root, biased variance (wavemamba_arch.py:532-569).
 * x, y (B, C, L) fp32; C in {8, 16, 32, 64}.  Forward only.
 * -------------------------------------------------------------------------------------------- */
int wm_layernorm2d_fwd(const float* x, const float* weight, const float* bias, float eps, float* y,
                       int B, int64_t L, int C, void* stream);

/* --------------------------------------------------------------------------------------------
 * Channel Gram matrix over the pixel axis (HFE branch): G[b][i][j] = sum_l X[b][i][l] Y[b][j][l],
 * nx[b][i] = sum_l X^2, ny likewise.  X, Y (B, C, L) fp32, C <= 32; G (B, C, C), nx, ny (B, C) are
 * overwritten.  Replaces torch.cdist(x, perception) (wavemamba_arch.py:664) and
 * normalize(q) @ normalize(k)^T (:787-790), both contractions over H*W.  `workspace`
 * (wm_gram_workspace_bytes, 16-byte aligned) holds the per-block partial sums, which are added in a
 * fixed order: the result is bit-reproducible run to run, as the reference's matmul is.
 * -------------------------------------------------------------------------------------------- */
size_t wm_gram_workspace_bytes(int B, int C, int64_t L);
int wm_gram_fwd(const float* X, const float* Y, float* G, float* nx, float* ny, void* workspace,
                size_t workspace_bytes, int B, int C, int64_t L, void* stream);

/* Training-side gradients of the two streaming helpers above.
 *   wm_dwconv3x3_wgrad: dW (C,1,3,3) and db (C, may be NULL) of the depth-wise conv from x and gy (B,C,H,W);
 *     the input gradient is wm_dwconv3x3_fwd(gy, weight, NULL, act = 4).
 *   wm_layernorm2d_bwd: reference LayerNormFunction.backward (wavemamba_arch.py:545-557): gx, dweight, dbias. */
int wm_dwconv3x3_wgrad(const float* x, const float* gy, float* dW, float* db, int B, int C, int H, int W,
                       void* stream);
int wm_layernorm2d_bwd(const float* x, const float* weight, const float* gy, float eps, float* gx,
                       float* dweight, float* dbias, int B, int64_t L, int C, void* stream);


/* Dense 3x3 (stride 1, zero padding 1) and 1x1 convolutions, NCHW fp32, with the element-wise neighbours they
 * have in the HFE branch fused (wavemamba_arch.py: PAConv k2/k3/k4 :690-697 on
 * cat([x, gather(candidates, idx)]) :666/:713, CMTAttention.qkv/.project_out :768-797, FeedForward 1x1s :733-742,
 * DownFRG.l_conv on cat([x_LL, x_d]) :966/:975, upFRG.h_out_conv :993/:1006, UNet.conv_01/.last/ps_down*
 * :1015-1037).  Forward only.  Runs on the bf16 matrix cores with a two-term split of both operands (three
 * products, fp32 accumulation): 3-4e-6 relative to an fp64 convolution, see csrc/conv2d.hip.h.
 *   wm_conv2d_prep    weight (Cout, Cin, ks, ks), ks in {1, 3} -> `wfrag` (wm_conv2d_wfrag_bytes(Cout, Cin, ks) bytes,
 *                     16-byte aligned, caller-owned): split, zero-padded to 16 | Cin and 32 | Cout, MFMA fragment
 *                     order.  Redo whenever the weight changes.
 *   wm_conv2d_fwd     x = cat([xa (B, Ca, H, W), xb' (B, Cb, H, W)], 1) where xb' = xb (Cb_src == Cb, xb_index NULL)
 *                     or xb'[b, c] = xb[b, xb_index[b, c]] for xb (B, Cb_src, H, W), xb_index (B, Cb) int32
 *                     (torch.gather over channels, never materialised); Cb may be 0 (xb, xb_index ignored);
 *                     Ca % 8 == 0 when Cb > 0.
 *                     y (B, Cout, H, W) = conv(x) + bias;  y *= sigmoid(gate) if gate;  y += residual if residual
 *                     (gate, residual: (B, Cout, H, W), may be NULL; bias may be NULL).  `wfrag` prepared for
 *                     Cin = Ca + Cb and the same ks.  H * W < 2^31.
 *   wm_conv2d_gated_fwd  y = conv3x3(x; wfrag3) * sigmoid(conv1x1(x; wfrag1) + bias1) over the same (concatenated /
 *                     gathered) input: PAConv's k3(x) * sigmoid(k2(x)) (:694-697) in one kernel - the 1x1 shares the
 *                     3x3's centre-tap operand fragments.  wfrag3 / wfrag1 prepared with ks = 3 / 1 for (Cout, Ca + Cb).
 *   wm_conv2d_select  which 3x3 kernel serves wm_conv2d_fwd / wm_conv2d_gated_fwd: 0 (default) by problem size - the
 *                     persistent wave-specialised kernel (one workgroup per compute unit, producer waves fetch / split
 *                     / stage, consumer waves multiply: conv2d_ws.hip.h) when every compute unit gets a few tiles, the
 *                     first-generation kernel otherwise; 1 always the first generation; 2 the wave-specialised one
 *                     wherever its limits allow (a batch element of each tensor < 4 GiB, <= 128 gathered channels, not
 *                     gate and residual together).  Both accumulate in the same order: results are bit-identical.
 *                     Process-wide; WM_CONV_WS=0 in the environment = select(1).  WM_EINVAL for another mode. */
int wm_conv2d_select(int mode);
size_t wm_conv2d_wfrag_bytes(int Cout, int Cin, int ks);
int wm_conv2d_prep(const float* weight, void* wfrag, int Cout, int Cin, int ks, void* stream);
/* y = conv1x1(LayerNorm2d(x; ln_weight, ln_bias, ln_eps)) + bias (+ residual): HFEBlock's norm1 -> attn.qkv and norm2 ->
 * ffn.project_in[0] (wavemamba_arch.py:843-851 over LayerNorm2d :532-569) as ONE kernel - the normalisation happens in the registers
 * the 1x1 kernel stages its pixels through (same arithmetic as wm_layernorm2d_fwd + wm_conv2d_fwd: bit-identical results), the
 * LayerNorm launch and its 256 B per position are gone.  Cin == 32 only (WM_EUNSUPPORTED otherwise); `wfrag` from
 * wm_conv2d_prep(weight (Cout, 32, 1, 1)); bias, residual may be NULL. */
int wm_conv2d_ln_fwd(const float* x, const float* ln_weight, const float* ln_bias, float ln_eps, const void* wfrag, const float* bias,
                     const float* residual, float* y, int B, int Cin, int Cout, int H, int W, void* stream);
/* The training form of the same convolutions (the F.conv2d calls of wavemamba_arch.py under autograd, femasr_model.py:170-181:
 * forward, and - on the transposed, flipped weight - the input gradient): fp16 matrix cores, two-term split of both operands
 * (22 significant bits, fp32-class: ~1e-7 relative to an fp64 convolution instead of the bf16 form's 3-4e-6, which the
 * parameter gradients of the deepest blocks do not tolerate), each tensor scaled by a power of two taken from its largest
 * finite magnitude so that fp16's exponent range is never the limit.  y = conv(x, weight) + bias (bias may be NULL), ks in {1, 3}.
 * `amax`: TWO device floats, written by the call ({max |x|, max |weight|}; in a registered zero arena - wm_zero_arena_register -
 * the call issues no memset node); `wfrag`: wm_conv2d_wfrag_bytes(Cout, Cin, ks) bytes, 16-byte aligned; both caller-owned
 * scratch.  dgrad != 0: `weight` (Cin, Cout, ks, ks) is the FORWARD convolution's weight and the call computes that convolution's
 * input gradient from x = gy: the convolution with w'[co][ci][ky][kx] = weight[ci][co][ks-1-ky][ks-1-kx] (what autograd evaluates
 * as conv(gy, weight.transpose(0, 1).flip(2, 3))), its fragments read straight from `weight` - no flipped copy exists; Cin / Cout
 * are those of the gradient convolution (= the forward's Cout / Cin).  Two launches: magnitudes + weight preparation, convolution.
 * Same kernels, tilings and limits as wm_conv2d_fwd; no concatenated / gathered second input, no gate; nothing synchronises the
 * host.  (Rounds 4-5 exported the steps one by one - wm_conv2d_amax / _prep_f16 / _fwd_f16 / wm_conv2d_f16: folded into this call.) */
int wm_conv2d_f16_steps(const float* x, const float* weight, const float* bias, float* y, float* amax, void* wfrag, int B, int Cin,
                        int Cout, int H, int W, int ks, int dgrad, void* stream);
int wm_conv2d_fwd(const float* xa, const float* xb, const int* xb_index, const void* wfrag, const float* bias,
                  const float* gate, const float* residual, float* y, int B, int Ca, int Cb, int Cb_src, int Cout,
                  int H, int W, int ks, void* stream);
int wm_conv2d_gated_fwd(const float* xa, const float* xb, const int* xb_index, const void* wfrag3, const void* wfrag1,
                        const float* bias1, float* y, int B, int Ca, int Cb, int Cb_src, int Cout, int H, int W,
                        void* stream);

/* Backward of wm_ss2d_core_fwd: autograd of SS2D.forward_core (wavemamba_arch.py:446-478) - the four directional
 * flattenings, the x_proj / dt_proj einsums and the selective scan - without materialising xs / dts / Bs / Cs.
 *   dy_*            gradients of the four outputs in wm_ss2d_core_fwd's order, each (B, D, L) row-major positions; for
 *                   the merged forward pass the same pointer four times.
 *   dx (B, D, H, W), dx_proj_weight (4, R + 2N, D), ddt_projs_weight (4, D, R), ddt_projs_bias (4, D),
 *   dA_logs (4 D, N), dDs (4 D): overwritten.
 * Limits: N <= 32, R <= 4, D <= 64; workspace wm_ss2d_core_bwd_workspace_bytes(...), 16-byte aligned. */
size_t wm_ss2d_core_bwd_workspace_bytes(int B, int D, int H, int W, int N, int R);
int wm_ss2d_core_bwd(const float* x, const float* x_proj_weight, const float* dt_projs_weight,
                     const float* dt_projs_bias, const float* A_logs, const float* Ds, const float* dy_row_fwd,
                     const float* dy_row_rev, const float* dy_col_fwd, const float* dy_col_rev, float* dx,
                     float* dx_proj_weight, float* ddt_projs_weight, float* ddt_projs_bias, float* dA_logs, float* dDs,
                     void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int N, int R, void* stream);

/* nn.LayerNorm(C) over the last axis of contiguous token tensors (T, C): LFSSBlock.ln_1 / ln_2, SS2D.out_norm
 * (wavemamba_arch.py:385, :493, :507-526), forward and backward (training).  C in {8, 16, 32, 64}; biased variance,
 * eps inside the square root; dweight / dbias (C) are overwritten.  16-byte aligned pointers. */
int wm_layernorm_tok_fwd(const float* x, const float* weight, const float* bias, float eps, float* y, int64_t T, int C,
                         void* stream);
int wm_layernorm_tok_bwd(const float* x, const float* weight, const float* gy, float eps, float* gx, float* dweight,
                         float* dbias, int64_t T, int C, void* stream);

/* The caller's I/O step (inference_wavemamba.py:99-113, basicsr/utils/img_util.py:9-98).
 *   wm_image_pre_u8   image (h, w, 3) uint8 on the device -> out (3, Hp, Wp) fp32 = channel-major, / 255, reflect-padded
 *                     bottom / right (Hp >= h, Wp >= w, pads smaller than the image); swap_rb: BGR -> RGB.
 *   wm_image_post_u8  in (3, Hp, Wp) fp32 -> image (h, w, 3) uint8 = round_half_even(clamp(in[:, :h, :w], 0, 1) * 255);
 *                     swap_rb: RGB -> BGR. */
int wm_image_pre_u8(const uint8_t* image, float* out, int h, int w, int Hp, int Wp, int swap_rb, void* stream);
int wm_image_post_u8(const float* in, uint8_t* image, int h, int w, int Hp, int Wp, int swap_rb, void* stream);

/* dW (O, I) = gy^T x for token-major gy (T, O), x (T, I): weight gradient of nn.Linear (SS2D.in_proj / out_proj,
 * wavemamba_arch.py:345 / :386) in training.  (O, I) in {(128,32), (32,64), (64,16), (16,32), (32,16), (16,16), (64,32),
 * (32,32), (16,64)}: the in_proj / out_proj shapes of hidden_dim 32, 16, 8. */
int wm_linear_wgrad(const float* gy, const float* x, float* dW, int64_t T, int O, int I, void* stream);
/* Weight gradient of a dense convolution (stride 1, 'same' zero padding, ks = 1 or 3):
 *   dW[co][ci][ky][kx] = sum_{b,h,w} gy[b][co][h][w] * x[b][ci][h + ky - ks/2][w + kx - ks/2]
 * - what autograd evaluates for the weight of every nn.Conv2d reached from WaveMamba.forward in training (ATen:
 * convolution_backward, output_mask[1]; on ROCm MIOpen's NHWC implicit-GEMM kernels between layout transposes).
 * gy (B, Cout, H, W), x (B, Cin, H, W) fp32 NCHW, 16-byte aligned; dW (Cout, Cin, ks, ks) is overwritten.
 * bf16 matrix cores with split operands (three products, ~4e-6 per product, fp32 accumulation).
 * Supported: W % 32 == 0, Cout <= 16, 32, 64 or 96 (tiles of 16: 1, 2, 4, 6); else WM_EUNSUPPORTED (workspace_bytes 0) and
 * the caller keeps ATen's gradient.
 * db (Cout) or NULL: the BIAS gradient db[co] = sum_{b,h,w} gy[b][co][h][w] (output_mask[2]) from the same pass over gy - fp32
 * sums in a fixed order (bit-reproducible), overwritten; wm_plane_sums is the stand-alone form.
 */
size_t wm_conv2d_wgrad_workspace_bytes(int B, int Cin, int Cout, int H, int W, int ks);
int wm_conv2d_wgrad(const float* gy, const float* x, float* dW, float* db, void* workspace, size_t workspace_bytes, int B, int Cin,
                    int Cout, int H, int W, int ks, void* stream);

/* The two loss terms of FeMaSRModel.optimize_parameters (basicsr/models/femasr_model.py:171-179): nn.L1Loss() on the prediction and
 * FFTLoss (basicsr/losses/losses.py:306-313: an L1 mean over the stacked real / imaginary parts of rfft2 = the same mean over the
 * interleaved floats of the complex tensor).  out[0] = mean_i |a_i - b_i| (zeroed here; one atomic per workgroup: ~1e-7 run to run);
 * ga_i = gout[0] * sign(a_i - b_i) / n (sign(0) = 0, like ATen).  a, b, ga: n fp32 elements, dense; gout: one device float.
 * (ATen's own reduction zeroes its semaphores with hipMemsetAsync, which must not be captured into a HIP graph on this runtime:
 * see the note on zero-initialised outputs below.) */
int wm_l1_mean_fwd(const float* a, const float* b, float* out, int64_t n, void* stream);
int wm_l1_mean_bwd(const float* a, const float* b, const float* gout, float* ga, int64_t n, void* stream);

/* sums (C) = sum over batch and plane of x (B, C, H, W): the bias gradient of a convolution (training). */
int wm_plane_sums(const float* x, float* sums, int B, int C, int H, int W, void* stream);

/* --------------------------------------------------------------------------------------------
 * Element-wise gates and scaled skips of the LFSSBlock TRAINING path, forward and backward:
 *   wm_gate_fwd:      out = act(a) * b          act 1 = SiLU  (SS2D: y * F.silu(z),           wavemamba_arch.py:493)
 *                                               act 2 = GELU  (ffn:  F.gelu(x1) * x2, erf form, :228-229)
 *                                               act 3 = sigmoid (PAConv: k3(x) * sigmoid(k2(x)), :697-699)
 *   wm_gate_bwd:      ga = g * b * act'(a),  gb = g * act(a)
 *   wm_scale_add_fwd: out = x * scale[c] + o    (LFSSBlock: input * skip_scale + ..., x * skip_scale2 + ..., :525-526)
 *   wm_scale_add_bwd: gx = g * scale[c],  gscale[c] = sum_{b, p} g * x   (zeroed here, accumulated with one atomic per block)
 * Gate operands are (B, per_b) fp32 with batch strides in elements, so the two halves of a channel chunk are passed as
 * views; scale_add operands are dense (B, C, L).  16-byte accesses when sizes / strides / pointers allow, scalar otherwise.
 */
int wm_gate_fwd(const float* a, const float* b, float* out, int act, int B, int64_t per_b, int64_t stride_a, int64_t stride_b,
                int64_t stride_out, void* stream);
int wm_gate_bwd(const float* a, const float* b, const float* g, float* ga, float* gb, int act, int B, int64_t per_b,
                int64_t stride_a, int64_t stride_b, int64_t stride_g, int64_t stride_ga, int64_t stride_gb, void* stream);
int wm_scale_add_fwd(const float* x, const float* scale, const float* o, float* out, int B, int C, int64_t L, void* stream);
int wm_scale_add_bwd(const float* g, const float* x, const float* scale, float* gx, float* gscale, int B, int C, int64_t L,
                     void* stream);

/* Small-tensor steps of the HFE branch as single kernels (csrc/hfe.hip.h).  Forward only.
 *   wm_match_index   channel matching with every channel kept (wavemamba_arch.py:659-666, match_factor = 1):
 *                    index[b, c] = argmin_j (nx[b, c] + ny[b, j] - 2 G[b, c, j]) from wm_gram_fwd's outputs; (B, C) int32.
 *   wm_attn_fold     transposed attention folded into its output projection (:787-797): Wout[b] (C, C) =
 *                    Wpo (C, C) @ blockdiag_h softmax_j(G[b, h] / (max(|q_i|, 1e-12) max(|k_j|, 1e-12)) * temperature[h]);
 *                    G (B * heads, C / heads, C / heads), nq / nk (B * heads, C / heads) squared norms.  C <= 64.
 *                    project_out(attn @ v) is then the 1x1 convolution of v with Wout[b] (+ project_out's bias).
 *   wm_skff_fwd      SKFF of three sub-bands (:937-959): out = sum_i softmax_i(Wfc[i] prelu(Wdu mean(x0 + x1 + x2))) x_i.
 *                    x_i, out (B, C, H, W); Wdu (d, C); prelu (1); Wfc (3, C, d); C <= 64, d <= 16; workspace
 *                    wm_skff_workspace_bytes(B, C) bytes. */
int wm_match_index(const float* G, const float* nx, const float* ny, int* index, int B, int C, void* stream);
int wm_attn_fold(const float* G, const float* nq, const float* nk, const float* temperature, const float* Wpo,
                 float* Wout, int B, int C, int heads, void* stream);
size_t wm_skff_workspace_bytes(int B, int C);
int wm_skff_fwd(const float* x0, const float* x1, const float* x2, const float* Wdu, const float* prelu,
                const float* Wfc, float* out, void* workspace, size_t workspace_bytes, int B, int C, int d, int H, int W,
                void* stream);

/* The UNet's pixel-unshuffled image inputs, ps_down{1,2,3} = nn.Sequential(nn.PixelUnshuffle(r), nn.Conv2d(r*r*Cin, Cout, 1))
 * (wavemamba_arch.py:1014-1025, applied to the input image at :1043-1045), as one r x r / stride-r convolution read straight from
 * the image: y[b, o, yo, xo] = bias[o] + sum_{c,i,j} weight[o, (c r + i) r + j] * img[b, c, r yo + i, r xo + j].
 *   img (B, Cin, H, W) fp32, 16-byte aligned, H and W multiples of r; weight (Cout, Cin r r) = the Conv2d's (Cout, Cin r r, 1, 1);
 *   bias (Cout) or NULL; y (B, Cout, H / r, W / r) fp32.  r in {2, 4, 8}, Cout in {16, 32, 48, 64}, Cin r r Cout <= 16384
 *   (else WM_EUNSUPPORTED: callers keep the two modules).  fp32 FMAs in PixelUnshuffle's (c, i, j) channel order. */
int wm_patchify_conv_fwd(const float* img, const float* weight, const float* bias, float* y, int B, int Cin, int Cout, int H, int W,
                         int r, void* stream);

/* --------------------------------------------------------------------------------------------
 * Profiling hooks used by bench.py (HIP events recorded on the launch stream around each kernel
 * class).  Disabled by default; when disabled the library records nothing.
 *   kernel ids: 0 haar analysis (dwt fwd / iwt bwd), 1 haar synthesis (iwt fwd / dwt bwd),
 *               2 scan chunk-reduce, 3 scan carry, 4 scan chunk-scan (drop-in op), 5 wm_lfss_in_fwd,
 *               6 ss2d projection records (core backward), 7 depth-wise conv + SiLU (SS2D.conv2d), 8 ss2d core chunk-scan
 *               (+ the merged-output sum), 9 wm_lfss_mid_fwd, 10 ss2d core chunk-reduce (+ the parameter prep),
 *               11 wm_lfss_out_fwd / wm_lfss_out_conv_fwd, 12 selective-scan backward (all phases),
 *               13 dense 3x3 convolution, 14 1x1 convolution, 15 SKFF (all three kernels), 16 LayerNorm2d,
 *               17 depth-wise conv without SiLU (HFE branch, ffn), 18 wm_patchify_conv_fwd, 19 unused
 * wm_prof_enable(mask): bit k of `mask` switches recording for kernel id k (0 = off, ~0u = every class);
 * a non-zero mask also clears what was recorded before.  Two hipEventRecord calls cost ~10 us of stream
 * time per launch, so a caller timing a whole step enables only the classes it needs.
 * wm_prof_collect synchronises the recorded events (host-blocking) and returns, per kernel id,
 * the number of launches and their summed duration in milliseconds since the last wm_prof_enable(mask != 0).
 * -------------------------------------------------------------------------------------------- */
#define WM_PROF_NKERNELS 20
void wm_prof_enable(unsigned mask);
int wm_prof_collect(int* launches /*[WM_PROF_NKERNELS]*/, double* total_ms /*[WM_PROF_NKERNELS]*/);

/* Zero-initialised accumulator outputs.  The entry points whose kernels ADD into an output with atomics - wm_dwconv3x3_wgrad
 * (dW, db), wm_layernorm2d_bwd and wm_layernorm_tok_bwd (dweight, dbias), wm_linear_wgrad (dW), wm_plane_sums (sums),
 * wm_scale_add_bwd (gscale), wm_conv2d_f16_steps (amax), wm_l1_mean_fwd (out) - zero that output first, on the stream, with a small
 * KERNEL (round 6; never hipMemsetAsync: captured into a HIP graph that is a memset node, and ROCm 7.0.2 fills with a pattern it
 * re-reads from recycled memory when the graph is launched again after eager work - profiles/r06/graph_memset_node.md).  A caller that
 * hands these buffers out of memory it has ALREADY zeroed (one memset for many buffers: the host side's bump arena,
 * ops._zeros_small) registers the range; a buffer lying entirely inside a registered range is then taken as zero and the launch is
 * skipped (a BASELINE config-3 training step had 323 of them, 1.4 ms of stream time).  The caller's contract: every buffer it
 * passes from a registered range is zero at that point of the stream (ONE-SHOT: a buffer that has been accumulated into is not zero
 * any more - passing it again, to this or another accumulate-into entry point, adds onto its old contents without any error; hand
 * every buffer out of the range once), and the range is unregistered before its memory is freed.
 * register: WM_EINVAL for an empty range or one that overlaps a registered range; unregister: WM_EINVAL for an unknown base.
 * Process-wide, thread-safe.  No reference counterpart (the reference's gradients come from ATen). */
int wm_zero_arena_register(void* base, size_t bytes);
int wm_zero_arena_unregister(void* base);

/* Host-blocking wait for `event` (a hipEvent_t recorded OUTSIDE any capture) that is legal while a stream of this
 * thread is being captured: hipEventSynchronize under a thread-local relaxed capture mode
 * (hipThreadExchangeStreamCaptureMode), restored before returning.  The host side uses it for a prepared-weight buffer
 * whose preparation kernel ran on another stream before the capture began: an uncaptured event cannot be waited for by a
 * capturing stream, and a plain hipEventSynchronize invalidates a capture in the global mode (torch.cuda.graph's
 * default).  No reference counterpart (the reference has no graph capture). */
int wm_event_synchronize_relaxed(void* event);

#ifdef __cplusplus
}
#endif
#endif /* WAVEMAMBA_HIP_H */
