/*
 * fcd_hip.h -- C ABI of libfcd_hip.so: the MI355X (gfx950) kernels behind the
 * FCD Fixed-Point GAN hot path.
 *
 * The reference (denix56/fcd) has no FFI for this path: its boundary is the
 * Python nn.Module / solver API (models/fixed_point_gan.py, fcd_solver.py) and
 * every op below is a stock torch.nn / torch op there.  Each entry point cites
 * the reference line whose arithmetic it replaces.  The Python side
 * (fcd_amd/) binds these with ctypes and keeps the reference's module and
 * solver signatures.
 *
 * Conventions
 *  - all pointers are DEVICE pointers unless named host_*; the library borrows
 *    them for the duration of the call and allocates nothing;
 *  - activations inside the library are NHWC fp32 with the channel count padded
 *    to a multiple of 4 ("physical" channels, pad lanes hold zeros); the
 *    module boundary is NCHW fp32 like the reference;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *  - return value: 0 = ok, otherwise a negative FCD_E* code or a positive
 *    hipError_t; fcd_last_error() gives a message for the calling thread;
 *  - scratch comes from the caller: query *_workspace_bytes first.
 */
#ifndef FCD_HIP_H
#define FCD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: FcdConvDesc grew a_amax / w_amax / out_amax and FcdWgradDesc g_amax / x_amax (FCD_PREC_F16X2): a caller built
 * against version 1 would hand over descriptors that are too short.  Check fcd_abi_version() == FCD_ABI_VERSION and
 * fcd_conv_desc_bytes() / fcd_wgrad_desc_bytes() == sizeof of your own structs before the first call. */
#define FCD_ABI_VERSION 2

#define FCD_OK 0
#define FCD_EINVAL (-1)   /* bad descriptor / shape / alignment            */
#define FCD_EWORKSPACE (-2) /* workspace too small                          */
#define FCD_ENODEVICE (-3)  /* no HIP device / wrong architecture           */

/* precision of the MFMA contraction */
#define FCD_PREC_F32 0   /* v_mfma_f32_32x32x2_f32: exact fp32 fma chain     */
#define FCD_PREC_BF16 1  /* operands rounded to bf16 at LDS staging, fp32 accumulate */
#define FCD_PREC_BF16X3 2 /* each fp32 operand split into three bf16 pieces in registers, six bf16 MFMAs
                             per product group, fp32 accumulate: fp32-faithful at the bf16 matrix rate */
#define FCD_AMAX_SLOTS 1024 /* an "amax" scalar of FCD_PREC_F16X2 is an array of 1024 floats (one per producing block) whose maximum is the value */
#define FCD_PREC_F16X2 3  /* convolutions that take pre-split weights (fcd_conv_weight_format 2): each fp32 operand scaled by
                             a power of two and split into two fp16 pieces, three fp16 MFMAs (22+ operand bits; fp32
                             accumulate); every other shape runs as FCD_PREC_BF16X3.  Needs FcdConvDesc.a_amax / w_amax. */

#define FCD_PREC_F16 4    /* the reduced-precision variant ("16-bit MFMA", BASELINE.json configs[1]): FCD_PREC_F16X2's kernels, operand
                             formats (weight format 2 / 3, the same maxima, the same overflow guard) and fall-backs, but only the
                             high fp16 piece of every operand is used: ONE fp16 MFMA per product group, 11 significant bits per
                             operand (bf16 has 8), fp32 accumulate.  Per-kernel rel-err <= 1e-3 (north_star's bf16 bar: 1e-2). */

/* activation codes accepted where an InstanceNorm entry point takes `act_slope` (values outside [0,1]) */
#define FCD_ACT_SILU 2.0f
#define FCD_ACT_MISH 3.0f

const char* fcd_last_error(void);
int fcd_abi_version(void);
/* sizeof(FcdConvDesc) / sizeof(FcdWgradDesc) as this build of the library sees them (a binding compares them with
 * its own struct sizes: a mismatch means mixed versions of header and library).  No reference counterpart. */
int fcd_conv_desc_bytes(void);
int fcd_wgrad_desc_bytes(void);
/* Checks that `device` is a gfx950 part and makes it current. */
int fcd_init(int device);
/* Overflow guard of FCD_PREC_F16X2 (no reference counterpart: the reference computes in plain fp32).  The scaled
 * fp16 pieces rely on max |.| values that OTHER kernels recorded (a_amax, w_amax, g_amax, x_amax); a stale or
 * under-estimated maximum would overflow fp16 (65504) and yield Inf / NaN silently.  Register one device word per
 * GPU (4-byte aligned, zeroed by the caller, owned by the caller; NULL unregisters): every FCD_PREC_F16X2 kernel and
 * weight pack stores 1 to it (plain store, sticky) when a scaled operand exceeds 2^15 -- four times what a correct
 * maximum allows (a NaN operand is not an overflow: it propagates into the result as in fp32).  The caller polls the
 * word whenever it likes (it costs a device-to-host copy) and zeroes it again.  Guarantee with a correct maximum: operands are represented to 2^-22 relative for every element
 * within 2^-15 of the tensor's maximum (two 11-bit pieces); smaller elements keep an ABSOLUTE error below
 * 2^-37 x the tensor's maximum (the low piece becomes subnormal), i.e. they lose relative, not absolute accuracy. */
int fcd_set_overflow_flag(int device, uint32_t* flag);

/* ------------------------------------------------------------------------- *
 * Implicit-GEMM convolution, "gather" form.
 *   out[n, a*ostride+ooff_h[ph], b*ostride+ooff_w[ph], co] =
 *     epilogue( sum_{th,tw,ci} in[n, a*istride - pad_h[ph] + th,
 *                                    b*istride - pad_w[ph] + tw, ci]
 *                              * w[ph][co][th][tw][ci] )
 * for (a, b) in GH x GW and every phase ph < nphase.  One phase with
 * ostride 1 is an ordinary nn.Conv2d (fixed_point_gan.py:27,30,45,52,68,111,
 * 134,135); four 2x2-tap phases with ostride 2 are nn.ConvTranspose2d(k4,s2,p1)
 * (fixed_point_gan.py:63) and equally the data gradient of Conv2d(k4,s2,p1);
 * one phase with flipped taps is the data gradient of a stride-1 conv.
 * epilogue(v): v += bias[co]; v = v > 0 ? v : v*act_slope (act_slope 1 = none);
 *              v *= (mask_src[o] > 0 ? 1 : mask_slope) if mask_src;
 *              v += residual[o] if residual.
 * ------------------------------------------------------------------------- */
typedef struct FcdConvDesc {
  int32_t N, IH, IW, Cin;        /* input NHWC, Cin physical (multiple of 4)     */
  int32_t GH, GW;                /* output pixel grid per phase                  */
  int32_t OH, OW, Cout, CoutP;   /* output NHWC, CoutP physical channel stride   */
  int32_t KH, KW;                /* taps per phase                               */
  int32_t istride, ostride;
  int32_t nphase;
  int32_t pad_h[4], pad_w[4];
  int32_t ooff_h[4], ooff_w[4];
  int32_t splitk;                /* 0 = chosen by the library; >1: partial sums go
                                    through the workspace                        */
  int32_t precision;             /* FCD_PREC_*                                   */
  float act_slope;               /* LeakyReLU slope of the epilogue, 1 = identity */
  float mask_slope;              /* slope used with mask_src                     */
  const float* a_amax;           /* FCD_PREC_F16X2 only: device scalars max |in| and max |weight| (the powers of two  */
  const float* w_amax;           /* that scale the operands before the fp16 split are derived from them); else NULL */
  float* out_amax;               /* optional, fcd_conv_gather only, any precision: max |out| of this launch into a zeroed  */
                                 /* array of FCD_AMAX_SLOTS floats (the next layer's a_amax); from the kernel's epilogue   */
                                 /* where it writes final values, else by an fcd_absmax pass inside the call; else NULL   */
} FcdConvDesc;

size_t fcd_conv_workspace_bytes(const FcdConvDesc* d);
/* Symbol (template instance as rocprofv3 prints it, without namespace and arguments) of the last convolution
 * or weight-gradient kernel the calling thread launched -- kernel selection is by shape inside the library;
 * a benchmark uses this to find the launch that really ran in a rocprofv3 counter file.  No reference
 * counterpart (the reference dispatches to cuDNN). */
const char* fcd_last_conv_kernel(void);
int fcd_conv_gather(const FcdConvDesc* d, const float* in, const float* w_packed,
                    const float* bias, const float* mask_src, const float* residual,
                    float* out, void* workspace, size_t workspace_bytes, void* stream);

/* The convolution in front of an InstanceNorm (no bias, no activation) together with that norm's
 * statistics: out as fcd_conv_gather, and mean[n][c], rstd[n][c] (and, if not NULL, the unbiased
 * variance for the running-stat update) as fcd_instnorm_stats of out
 * (fixed_point_gan.py:27-28, 45-46, 52-53, 63-64: Conv2d / ConvTranspose2d + InstanceNorm2d).
 * Where the tile shape allows, the per-channel sums are taken from the accumulators in the
 * convolution's epilogue (out is not read again); otherwise this is the two calls in a row. */
size_t fcd_conv_gather_stats_workspace_bytes(const FcdConvDesc* d);
int fcd_conv_gather_stats(const FcdConvDesc* d, const float* in, const float* w_packed, float* out,
                          float* mean, float* rstd, float* var_unbiased, float eps, void* workspace,
                          size_t workspace_bytes, void* stream);

/* The same followed by the norm's own output: y = act((z - mean) * rstd * gamma + beta) (+ residual), and max |y| into
 * y_amax if not NULL (fcd_instnorm_apply_amax) -- Conv2d + InstanceNorm2d + activation of fixed_point_gan.py:27-29, 30-34
 * (the residual block: y = x + IN(conv(...))), 45-47, 52-54, 63-65.  z (the raw convolution output, kept for the backward
 * pass), mean, rstd and var_unbiased are written as by fcd_conv_gather_stats; results are bit-identical to
 * fcd_conv_gather_stats + fcd_instnorm_apply_amax.  For 32 x 32 outputs with fused statistics and a (Leaky)ReLU slope the
 * statistics are finished inside the normalising kernel (one launch after the convolution instead of two; switch
 * FCD_NO_IN1F=1 for A/B runs).  Workspace as fcd_conv_gather_stats_workspace_bytes.  y must not alias z or in. */
int fcd_conv_gather_stats_apply(const FcdConvDesc* d, const float* in, const float* w_packed, float* z,
                                float* mean, float* rstd, float* var_unbiased, float eps, const float* gamma,
                                const float* beta, const float* residual, float* y, float act_slope,
                                float* y_amax, void* workspace, size_t workspace_bytes, void* stream);

/* Weight gradient of the same convolution family:
 *   dw[o][i][kh][kw] (+)= sum_{n,a,b} gsmall[n,a,b,o] *
 *                          xbig[n, a*stride - pad + kh, b*stride - pad + kw, i]
 * (torch's conv2d weight gradient; for ConvTranspose2d the roles of input and
 * output gradient swap).  dw is in the PyTorch parameter layout [O][I][KH][KW]. */
typedef struct FcdWgradDesc {
  int32_t N, GH, GW, O, OP;      /* gsmall NHWC [N][GH][GW][OP], O logical       */
  int32_t BH, BW, I, IP;         /* xbig   NHWC [N][BH][BW][IP], I logical       */
  int32_t KH, KW, stride, pad;
  int32_t splitk;                /* 0 = chosen by the library                    */
  int32_t precision;
  int32_t accumulate;            /* 1: dw += result                              */
  const float* g_amax;           /* FCD_PREC_F16X2 only (3x3 halo kernel; BF16X3 elsewhere): device scalars      */
  const float* x_amax;           /* max |gsmall| and max |xbig|; else NULL                                       */
} FcdWgradDesc;

size_t fcd_wgrad_workspace_bytes(const FcdWgradDesc* d);
int fcd_conv_wgrad(const FcdWgradDesc* d, const float* gsmall, const float* xbig, float* dw,
                   void* workspace, size_t workspace_bytes, void* stream);
/* The same with a SECOND operand pair of the same geometry (N2 images): dw (+)= gsmall^T xbig + gsmall2^T xbig2 in one
 * pass over dw -- the K axis (pixels / images) of the kernel runs through both pairs.  What torch's autograd does with
 * two `AccumulateGrad`s into one parameter (the discriminator's weight gradients of the paired real / fake batch and of
 * the gradient penalty, fcd_solver.py:286-299; the generator's of its two passes per G step, fcd_solver.py:341-365): one
 * read-modify-write of the gradient -- and one slab reduction -- instead of two.  Only where fcd_conv_wgrad2_supported(d)
 * is 1: more than 32 output channels, FCD_PREC_BF16X3 or the FCD_PREC_F16 family, and either the 3 x 3 / 4 x 4 halo form
 * or N GH GW a multiple of 32 (split-once kernel).  g_amax2 / x_amax2: the second pair's maxima, as d->g_amax / d->x_amax
 * are the first's (the scaled-fp16 forms take one scale per operand from the larger of the two; NULL: the bf16 x 3 form,
 * as when d's are NULL).  Workspace: fcd_wgrad_workspace_bytes of the descriptor with N + N2 images.  No reference
 * counterpart as an interface (the reference's two backward passes each call cuDNN's weight gradient). */
int fcd_conv_wgrad2_supported(const FcdWgradDesc* d);
int fcd_conv_wgrad2(const FcdWgradDesc* d, const float* gsmall, const float* xbig, const float* gsmall2,
                    const float* xbig2, int N2, const float* g_amax2, const float* x_amax2, float* dw,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Parameter layout [O][I][KH][KW]  ->  gather-form operand [O][KH][KW][IP]. */
int fcd_pack_weight_fwd(const float* w, float* packed, int O, int I, int IP, int KH, int KW,
                        void* stream);
/* Parameter layout [O][I][KH][KW]  ->  adjoint operand [phase][I][TH][TW][OP]:
 * stride 1: one phase, taps flipped; stride 2 (k4,p1): four phases of 2x2 taps. */
int fcd_pack_weight_adj(const float* w, float* packed, int O, int OP, int I, int KH, int KW,
                        int stride, void* stream);

/* FCD_PREC_BF16X3: where fcd_conv_wants_split_weights(d) is 1 the convolution takes its weights
 * already split into three bf16 planes (each fp32 weight = h + m + l, round to nearest at each
 * level): the same element order as fcd_pack_weight_fwd / _adj, laid out [phase][plane][row][K]
 * bf16 (6 bytes per weight).  Elsewhere BF16X3 takes the fp32 packs and splits in registers. */
int fcd_conv_wants_split_weights(const FcdConvDesc* d);
/* Weight operand the convolution described by d takes: 0 = fp32 packs (fcd_pack_weight_fwd / _adj), 1 = three bf16
 * planes (_x3), 2 = two fp16 planes of the scaled weights (_h2; FCD_PREC_F16X2 wherever BF16X3 would take format 1).
 * 3 = fp32 packs as for 0, and the kernel (the 7x7 halo forms and the register-staged form of the layers with few
 * output rows -- 128- / 64-row tiles of 128 channels, Cin a multiple of 32 -- which split their operands themselves) takes
 * the two-fp16-piece form if a_amax and w_amax are both set (FCD_PREC_F16X2 only; without them it runs as BF16X3).
 * No reference counterpart (cuDNN chooses its own operand formats). */
int fcd_conv_weight_format(const FcdConvDesc* d);
/* FCD_PREC_F16X2 weight operands: [plane h | plane l][row][K] fp16 of s*w, s the power of two derived from *w_amax
 * (device scalar, max |w|), element orders as fcd_pack_weight_fwd / _adj ([phase][plane]... for the adjoint form). */
int fcd_pack_weight_fwd_h2(const float* w, void* packed, const float* w_amax, int O, int I, int IP, int KH, int KW,
                           void* stream);
int fcd_pack_weight_adj_h2(const float* w, void* packed, const float* w_amax, int O, int OP, int I, int KH, int KW,
                           int stride, void* stream);
/* The same two packs for up to FCD_PACK_BATCH_MAX weights of ONE shape in one launch (HOST arrays of `count` device
 * pointers): what a network does after every optimizer step for its equal-shaped layers (the generator's twelve
 * residual-block convolutions).  Same bytes as `count` single calls. */
#define FCD_PACK_BATCH_MAX 16
int fcd_pack_weight_fwd_h2_batch(const float* const* w, void* const* packed, const float* const* w_amax, int count,
                                 int O, int I, int IP, int KH, int KW, void* stream);
int fcd_pack_weight_adj_h2_batch(const float* const* w, void* const* packed, const float* const* w_amax, int count,
                                 int O, int OP, int I, int KH, int KW, int stride, void* stream);
int fcd_pack_weight_fwd_x3(const float* w, void* packed, int O, int I, int IP, int KH, int KW,
                           void* stream);
int fcd_pack_weight_adj_x3(const float* w, void* packed, int O, int OP, int I, int KH, int KW,
                           int stride, void* stream);

/* Small-Cout convolution heads of the Discriminator (conv1/conv2,
 * fixed_point_gan.py:134-135; stride 1): direct dot products, no MFMA.
 * w_packed = fcd_pack_weight_fwd layout [O][KH][KW][IP]; in NHWC [N][IH][IW][IP];
 * out / gout NCHW [N][O][OH][OW] with OH = IH + 2*pad - KH + 1. */
int fcd_head_fwd(const float* in, const float* w_packed, float* out, int N, int IH, int IW, int IP,
                 int O, int KH, int KW, int pad, void* stream);
/* din NHWC = (sum_{o,taps} gout[n,o,oh,ow] * w[o][kh][kw][i]) * lrelu'(mask_src) (+ din). */
int fcd_head_dgrad(const float* gout, const float* w_packed, float* din, const float* mask_src,
                   float mask_slope, int accumulate, int N, int IH, int IW, int IP, int O,
                   int KH, int KW, int pad, void* stream);
/* dw in the parameter layout [O][I][KH][KW] */
int fcd_head_wgrad(const float* gout, const float* in, float* dw, int accumulate, int N, int IH,
                   int IW, int I, int IP, int O, int KH, int KW, int pad, void* stream);

/* ------------------------------------------------------------------------- *
 * InstanceNorm2d(affine=True[, track_running_stats]) + activation, NHWC.
 * (fixed_point_gan.py:28,31,46,53,64 with the ReLU of :29,47,54,65.)
 * ------------------------------------------------------------------------- */
/* mean[n][c], rstd[n][c] (biased variance, eps) over HW.  Workspace:
 * fcd_instnorm_workspace_bytes(N, HW, C). */
size_t fcd_instnorm_workspace_bytes(int N, int HW, int C);
int fcd_instnorm_stats(const float* z, float* mean, float* rstd, float* var_unbiased, int N, int HW,
                       int C, float eps, void* workspace, size_t workspace_bytes, void* stream);
/* fcd_instnorm_stats for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same kernels and
 * the same bits as fcd_instnorm_stats, but the column partials live in a device array built into the
 * library (FCD_NOALLOC_SCRATCH_BYTES per GPU, held until the process ends; its address is resolved
 * once per device, in fcd_init or on the first call) instead of a caller workspace: the call makes no
 * hipMalloc / hipFree and takes no workspace.  Shapes whose partials exceed the array are rejected with
 * FCD_EWORKSPACE (all InstanceNorm shapes of the baseline step fit: <= 4 MiB).  Every call on a GPU
 * shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race; calls on one
 * stream are ordered.  Nothing has to be cleared between calls (every partial read is written first). */
#define FCD_NOALLOC_SCRATCH_BYTES (4u << 20)
int fcd_instnorm_stats_noalloc(const float* z, float* mean, float* rstd, float* var_unbiased, int N, int HW,
                               int C, float eps, void* stream);
/* eval mode of a tracked norm: mean/rstd broadcast from the running buffers. */
int fcd_instnorm_stats_from_running(const float* running_mean, const float* running_var,
                                    float* mean, float* rstd, int N, int C, float eps, void* stream);
/* running = (1-m)*running + m*batch-mean of (mean | unbiased var).  groups > 1: the N samples are
 * `groups` consecutive reference calls of N/groups samples each (paired batches): one momentum
 * update per group, in order, as the separate calls would make them. */
int fcd_instnorm_update_running(float* running_mean, float* running_var, const float* mean,
                                const float* var_unbiased, int N, int C, int groups, float momentum,
                                void* stream);
/* The same update for up to FCD_RUNNING_BATCH_MAX tracked norms of one generator call in ONE launch (the
 * reference makes them one by one inside nn.InstanceNorm2d.forward, models/fixed_point_gan.py:28,31; they
 * are independent, so the order between norms is free).  The four arrays are HOST arrays of `count` device
 * pointers; all norms share N, C, groups, momentum. */
#define FCD_RUNNING_BATCH_MAX 16
int fcd_instnorm_update_running_batch(float* const* running_mean, float* const* running_var,
                                      const float* const* mean, const float* const* var_unbiased, int count,
                                      int N, int C, int groups, float momentum, void* stream);
/* y = act((z-mean)*rstd*gamma + beta) (+ residual); relu: act_slope 0, lrelu: 0.01, none: 1;
 * FCD_ACT_SILU / FCD_ACT_MISH in place of a slope select nn.SiLU / nn.Mish
 * (fixed_point_gan.py:13-16; instnorm_apply / instnorm_bwd only: the generator's activations). */
int fcd_instnorm_apply(const float* z, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, const float* residual, float* y, int N, int HW, int C,
                       float act_slope, void* stream);
/* Backward.  g1 = gy * act'(y).  Pass 1 reduces s1[n][c] = sum g1, s2 = sum g1*xhat;
 * pass 2 writes dz = gamma*rstd*(g1 - s1/HW - xhat*s2/HW) (per-instance statistics) or
 * dz = gamma*rstd*g1 (use_running: eval-mode tracked norm), and
 * dgamma[c] (+)= sum_n s2, dbeta[c] (+)= sum_n s1.
 * The activation mask (y > 0) is taken from y, or - beta given, y may be NULL - recomputed
 * from z with the forward's own expression (one tensor read less); act_slope 1 needs neither. */
int fcd_instnorm_bwd(const float* gy, const float* y, const float* z, const float* mean,
                     const float* rstd, const float* gamma, const float* beta, float* dz,
                     float* dgamma, float* dbeta, int accumulate, int N, int HW, int C,
                     float act_slope, int use_running, void* workspace, size_t workspace_bytes,
                     void* stream);
/* FCD_PREC_F16X2 support (no reference counterpart: the reference runs cuDNN fp32 / AMP).  The _amax variants are the
 * calls above that also leave max |y| (max |dz|) in a device "scalar" of FCD_AMAX_SLOTS floats (one slot per
 * producing block, no atomics; the value is the maximum of the slots): slot = max(slot, ...), so the caller
 * zeroes the array first; fcd_absmax does the same for any tensor (n floats, a multiple of 4, 16-byte aligned). */
int fcd_absmax(const float* x, size_t n, float* amax, void* stream);
/* max |.| of up to FCD_ABSMAX_BATCH_MAX tensors in one launch (what a network's weights need once per optimizer step):
 * x and n are HOST arrays of `count` device pointers / lengths in floats; row i of amax_rows (count x FCD_AMAX_SLOTS
 * floats) is OVERWRITTEN with tensor i's partial maxima -- no zeroing beforehand, unlike fcd_absmax. */
#define FCD_ABSMAX_BATCH_MAX 32
int fcd_absmax_batch(const float* const* x, const size_t* n, int count, float* amax_rows, void* stream);
int fcd_instnorm_apply_amax(const float* z, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, const float* residual, float* y, int N, int HW, int C,
                            float act_slope, float* y_amax, void* stream);
int fcd_instnorm_bwd_amax(const float* gy, const float* y, const float* z, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, float* dz, float* dgamma, float* dbeta,
                          int accumulate, int N, int HW, int C, float act_slope, int use_running,
                          float* dz_amax, void* workspace, size_t workspace_bytes, void* stream);
/* The backward with the affine gradients deferred: dz (and max |dz|) as fcd_instnorm_bwd_amax, the per-image sums
 * s12[n][c] = (sum g1, sum g1*xhat) into the caller's buffer (N*C*2 floats) and no dgamma / dbeta.
 * fcd_instnorm_affine_batch then finishes dgamma[c] (+)= sum_n s12[n][c][1], dbeta[c] (+)= sum_n s12[n][c][0] for up to
 * FCD_AFFINE_BATCH_MAX norms of equal N and C in ONE launch (HOST arrays of `count` device pointers; either gradient
 * pointer of a norm may be NULL) -- the same sums in the same order as fcd_instnorm_bwd computes them, so the results
 * are bit-identical; what goes is one small launch per norm from the backward chain's critical path
 * (autograd of nn.InstanceNorm2d's weight / bias, fixed_point_gan.py:28,31,46,53,64). */
#define FCD_AFFINE_BATCH_MAX 32
int fcd_instnorm_bwd_s12(const float* gy, const float* y, const float* z, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, float* dz, float* s12, int N, int HW, int C,
                         float act_slope, int use_running, float* dz_amax, void* workspace,
                         size_t workspace_bytes, void* stream);
int fcd_instnorm_affine_batch(const float* const* s12, float* const* dgamma, float* const* dbeta, int count,
                              int N, int C, int accumulate, void* stream);

/* ------------------------------------------------------------------------- *
 * Generator prologue / epilogue (fixed_point_gan.py:73-75 and :78-82).
 * ------------------------------------------------------------------------- */
/* x NCHW [N][C][H][W], c [N][CD] -> NHWC [N][H][W][CP] = cat(x, tiled c), zero pad. */
int fcd_g_prologue(const float* x, const float* c, float* out, int N, int C, int CD, int CP, int H,
                   int W, void* stream);
/* x_fake = tanh(delta + x) with invalid (x == -1) passthrough; delta NHWC [.][CP], out NCHW. */
int fcd_g_epilogue_fwd(const float* delta, const float* x, float* x_fake, int N, int C, int CP,
                       int H, int W, void* stream);
/* gpre = gout * (1 - x_fake^2) * (x != -1): written NHWC [.][CP] (d delta, pad 0) and,
 * if gx != NULL, NCHW into gx (accumulate_gx: +=): d x through the tanh(delta + x) skip. */
int fcd_g_epilogue_bwd(const float* gout, const float* x, const float* x_fake, float* gdelta,
                       float* gx, int accumulate_gx, int N, int C, int CP, int H, int W,
                       void* stream);
/* layout changes at the module boundary */
int fcd_nchw_to_nhwc(const float* x, float* out, int N, int C, int CP, int H, int W, void* stream);
/* out (+)= first C channels of the NHWC tensor */
int fcd_nhwc_to_nchw(const float* x, float* out, int accumulate, int N, int C, int CP, int H, int W,
                     void* stream);

/* ------------------------------------------------------------------------- *
 * Elementwise / reductions of the loss graph (fcd_solver.py).
 * All scalar results are written to device memory (no host sync).
 * ------------------------------------------------------------------------- */
/* g * (y > 0 ? 1 : slope): LeakyReLU backward from the (in-place) output sign,
 * fixed_point_gan.py:115. */
int fcd_lrelu_bwd(const float* g, const float* y, float* out, size_t n, float slope, void* stream);
/* Smooth discriminator activations (act = FCD_ACT_SILU / FCD_ACT_MISH; a slope in [0,1] also works),
 * fixed_point_gan.py:13-16, 114-115, evaluated at the stored pre-activation z:
 *   fcd_act_fwd:  a = act(z)
 *   fcd_act_bwd:  out (+)= g * act'(z)                       (first backward)
 *   fcd_act_bwd2: u_out = ubar * act'(z), q_out = ubar * ga * act''(z)
 *                 (backward of the first backward, as the gradient penalty of
 *                  fcd_solver.py:227-239 needs it: ubar is the cotangent of g * act'(z), ga the g of
 *                  the first backward; q_out is the cotangent that flows on into z). */
int fcd_act_fwd(const float* z, float* a, size_t n, float act, void* stream);
int fcd_act_bwd(const float* g, const float* z, float* out, size_t n, float act, int accumulate, void* stream);
int fcd_act_bwd2(const float* ubar, const float* ga, const float* z, float* u_out, float* q_out, size_t n,
                 float act, void* stream);
/* out[c] (+)= sum over rows of x[rows][C]: bias gradients. */
int fcd_colsum(const float* x, float* out, int accumulate, size_t rows, int C, void* workspace,
               size_t workspace_bytes, void* stream);
size_t fcd_colsum_workspace_bytes(size_t rows, int C);
/* The three scalar reductions below are two-stage (fixed order, double accumulation);
 * their workspace holds fcd_reduce_workspace_bytes() bytes. */
size_t fcd_reduce_workspace_bytes(void);
/* out[0] = scale * sum(x) (torch.mean with scale = +-1/n; fcd_solver.py:287,293,318,324) */
int fcd_sum_scaled(const float* x, float* out, size_t n, float scale, void* workspace,
                   size_t workspace_bytes, void* stream);
/* out[i] = scale * gscale[0]: the gradient of scale*sum(x) */
int fcd_fill_scaled(float* out, size_t n, float scale, const float* gscale, void* stream);
/* out[0] = mean |a - b| (fcd_solver.py:326,354,359) */
int fcd_l1_mean(const float* a, const float* b, float* out, size_t n, void* workspace,
                size_t workspace_bytes, void* stream);
/* ga (+)= gscale[0]*sign(a-b)/n ; gb likewise with the opposite sign (either may be NULL) */
int fcd_l1_mean_bwd(const float* a, const float* b, const float* gscale, float* ga, float* gb,
                    int accumulate, size_t n, void* stream);
/* F.binary_cross_entropy_with_logits mean (fcd_solver.py:263):
 * max(x,0) - x*t + log1p(exp(-|x|)) */
int fcd_bce_logits_mean(const float* logit, const float* target, float* out, size_t n,
                        void* workspace, size_t workspace_bytes, void* stream);
int fcd_bce_logits_mean_bwd(const float* logit, const float* target, const float* gscale,
                            float* glogit, size_t n, void* stream);
/* x_hat = alpha[n]*a + (1-alpha[n])*b (fcd_solver.py:297) */
int fcd_lerp_rows(const float* a, const float* b, const float* alpha, float* out, int N,
                  size_t per_row, void* stream);
/* WGAN-GP (fcd_solver.py:237-239): norm[n] = sqrt(sum_row g^2); out[0] = mean (norm-1)^2 */
int fcd_gp_fwd(const float* g, float* norm, float* out, int N, size_t per_row, void* stream);
/* d out / d g = gscale[0] * 2*(norm-1)/(N*norm) * g */
int fcd_gp_bwd(const float* g, const float* norm, const float* gscale, float* gg, int N,
               size_t per_row, void* stream);

/* torch.optim.Adam step (fcd_solver.py:183,186), one flat tensor. */
int fcd_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                  float beta2, float eps, int step, void* stream);

/* ------------------------------------------------------------------------- *
 * Mask generation (fcd_solver.py:802-807, 762, 771-772).
 * diff[n][h][w] = mean_c |x_fake - x| / 2 (NCHW inputs);
 * mask = diff > thr compared in fp32 (uint8 0/1); either output may be NULL.
 * ------------------------------------------------------------------------- */
int fcd_diffmap_threshold(const float* x_fake, const float* x, float* diff, uint8_t* mask,
                          float thr, int N, int C, int H, int W, void* stream);
/* make_psuedo_masks' per-batch tail (fcd_solver.py:762-772) in one pass: pseudo = diff > thr (fp32
 * compare); cm[truth][pred] += counts over valid pixels (target 1 = clear, 2 = cloudy; cm is 4 device
 * uint64, may be NULL); out = pseudo + 1, invalid pixels (target == 0) -> 0.  target NULL: all valid. */
int fcd_pseudo_mask(const float* diff, const uint8_t* target, size_t n, float thr, uint8_t* out,
                    unsigned long long* cm, void* stream);

/* ------------------------------------------------------------------------- *
 * Scene path on the device.
 *
 * fcd_scene_tiles: PatchDataset.__getitem__ + Normalize for a batch of patches
 *   (data_loader.py:74-103 with crop_size 0, evaluate.py:90-93): scene[H][W][C] uint16,
 *   origins_rc[nb][2] = (row, col) patch origins (device memory), numpy 'reflect' padding
 *   past the bottom/right edge, x = (v - 32767.5) * fp32(1/32767.5) written NCHW fp32.
 *   *any_valid (device int, zeroed by the caller) is set to 1 if any sample is non-zero:
 *   the caller skips all-nodata batches like evaluate.py:104.
 * fcd_scene_stitch: difference[row:row+h, col:col+w] += delta[b, :h, :w] in float64
 *   (evaluate.py:117-122); delta[nb][patch][patch] fp32, raster[H][W] double.
 * fcd_scene_binarize: difference[~valid] = 0 with valid = any band != 0, then
 *   mask = difference > threshold compared in float64 (evaluate.py:49-51,
 *   fcd_solver.py:566-567); raster is updated in place, mask[H][W] uint8.
 * ------------------------------------------------------------------------- */
int fcd_scene_tiles(const uint16_t* scene, int H, int W, int C, int patch, const int* origins_rc, int nb,
                    float* x_nchw, int* any_valid, void* stream);
int fcd_scene_stitch(const float* delta, const int* origins_rc, int nb, int patch, double* raster, int H,
                     int W, void* stream);
/* valid[b] = 1 if tile b (origin origins_rc[b], reflect padding as fcd_scene_tiles) holds any non-zero
 * sample: the all-nodata test of evaluate.py:104 for every tile of a scene in ONE launch, so that the
 * caller decides which batches to skip with one device-to-host copy per scene instead of one per batch. */
int fcd_scene_tile_valid(const uint16_t* scene, int H, int W, int C, int patch, const int* origins_rc, int nb,
                         uint8_t* valid, void* stream);
/* write_tifs' byte rasters (evaluate.py:384-390): diff_u8 = (difference * 255).astype(uint8);
 * mask_u8 = 128 where prediction == 0, 255 where prediction == 1, 0 where not valid.  valid[H][W]
 * (uint8, e.g. target > 0) or, when NULL, "any band of scene != 0".  Either output may be NULL. */
int fcd_scene_tif_bytes(const double* raster, const uint8_t* prediction, const uint8_t* valid,
                        const uint16_t* scene, int H, int W, int C, uint8_t* diff_u8, uint8_t* mask_u8,
                        void* stream);
int fcd_scene_binarize(double* raster, const uint16_t* scene, int H, int W, int C, double threshold,
                       uint8_t* mask, void* stream);

/* ------------------------------------------------------------------------- *
 * Training / validation input pipeline.  Replaces, for a patch pool
 * resident in device memory, L8BiomeHDFDataset.__getitem__ + the transform chain of get_dataset
 * (data_loader.py:287-313, 370-377) + the collate of DataLoader:
 *   images (n_images,H,W,C) uint16 and optional masks (n_images,H,W) uint8 in the layout of
 *   prepare_landsat8_biome.py:136-138; index[nb] (device int64) picks the samples; aug (device
 *   int32 [nb][3] or NULL) = {flags, dy, dx}: bit 0 HorizontalFlip, bit 1 VerticalFlip
 *   (:372-373), then an integer translation by (dy, dx) whose uncovered pixels are 0 (nodata);
 *   Normalize(mean .5, std .5, max 65535) (:375) in fp32: (v - 32767.5) * fp32(1/32767.5);
 *   x_nchw (nb,C,H,W) fp32 (ToTensorV2, :376); mask_out (nb,H,W) uint8 gets the same geometry.
 * The host draws the random parameters; sub-pixel translation (albumentations Affine's
 * interpolation) is not reproduced.
 * ------------------------------------------------------------------------- */
int fcd_patch_batch(const uint16_t* images, long n_images, int H, int W, int C, const long* index,
                    const int* aug, int nb, float* x_nchw, const uint8_t* masks, uint8_t* mask_out,
                    void* stream);

/* ------------------------------------------------------------------------- *
 * Threshold search (fcd_solver.py:680-725, metrics.py:4-15).
 * target: 0 = invalid, 1 = clear, 2 = cloudy (class = target - 1).
 * fcd_valid_minmax folds min / max of diff over valid pixels into minmax[2]
 *   (device floats, initialised by the caller to +inf and 0; diff >= 0).
 * fcd_threshold_hist adds to hist[2][T+1] (device uint64, zeroed by the caller):
 *   hist[cls][j] counts valid pixels of class cls whose diff exceeds exactly the first j of
 *   the ascending fp32 thresholds, i.e. that are predicted 1 for thresholds 0..j-1; the
 *   confusion matrix of threshold t follows by suffix sums.  Integer counts: exact.
 * ------------------------------------------------------------------------- */
int fcd_valid_minmax(const float* diff, const uint8_t* target, size_t n, float* minmax, void* stream);
int fcd_threshold_hist(const float* diff, const uint8_t* target, size_t n, const float* thresholds, int T,
                       unsigned long long* hist, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FCD_HIP_H */
