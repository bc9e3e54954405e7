/*
 * wgs.h — C ABI of libwgs_hip.so, the MI355X (gfx950) native kernels behind the WarpedGANSpace
 * training inner loop (warp -> G -> R -> loss -> backward -> Adam).
 *
 * Conventions (all entry points):
 *   - plain `extern "C"`, pointers + sizes only, no torch types;
 *   - every pointer is a DEVICE pointer on the current HIP device, contiguous, fp32 unless noted;
 *   - the CALLER owns every buffer (outputs, workspaces); the library never allocates or frees;
 *   - asynchronous on the `stream` argument (a hipStream_t passed as void*), no hidden syncs,
 *     no global mutable state => callable concurrently from different threads / streams;
 *   - returns 0 on success, a negative errno-style code on failure (never throws);
 *     wgs_last_error() returns the thread-local message of the last failure.
 *
 * Each group cites the reference interface (file:line under the reference tree) it replaces.
 */
#ifndef WGS_H
#define WGS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* wgs_stream_t; /* hipStream_t */

const char* wgs_last_error(void);
int wgs_abi_version(void);
/* Development A/B switches (WGS_DMA_ALWAYS, WGS_NO_PATCH, ...: DESIGN.md) are read from the environment ONCE, at the first
 * launch that consults them, and never change afterwards — except through this test hook, which re-reads them (call it
 * with no launch in flight on another thread).  All default to off. */
void wgs_dev_reload_flags(void);
/* Launch accounting for bench.py (statistics; no launch path reads them): the number of kernels this library has launched in
 * the process so far, and — after wgs_dev_trace_kernels(1) — the symbol of the last implicit-GEMM kernel launched on the calling
 * thread, spelled as rocprofv3 prints it (e.g. "igemm_patch_kernel<1, 128, 128, 2, 2, 1, 0>"; "" before the first one). */
int64_t wgs_dev_launch_count(void);
void wgs_dev_trace_kernels(int on);
const char* wgs_dev_last_kernel(void);

/* ------------------------------------------------------------------------------------------------
 * RBF warping field  —  replaces SupportSets.forward + its autograd backward
 * (lib/support_sets.py:81-101).  For sample b with k = idx[b]:
 *     g   = -2 * sum_{i<n2} alpha[k,i] * gamma_k * exp(-gamma_k * |z_b - s_{k,i}|^2) * (z_b - s_{k,i})
 *     out = g / |g|_2                      (times scale[b] when `scale` != NULL)
 * gamma_k = exp(loggamma[k]) when `loggamma` != NULL (learn_gammas), else the constant `gamma`.
 *
 *   table    [K, n2*d]   SUPPORT_SETS (row = n2 = 2N support vectors of dim d)
 *   alphas   [K, n2]     ALPHAS
 *   loggamma [K] or NULL LOGGAMMA (the [K,1] parameter viewed flat)
 *   idx      [B] int64   selected warping function per sample (the one-hot mask's argmax)
 *   z        [B, d]
 *   scale    [B] or NULL optional per-sample factor (the trainer's shift magnitude)
 *   out      [B, d]
 *   ws       workspace of wgs_rbf_ws_floats(B,n2,d) floats; wgs_rbf_fwd fills it, wgs_rbf_bwd reads it
 *            (layout: g_raw[B,d] | gnorm[B] | r2[B,n2] | partial[B,S,d])
 * Requirements: d % 4 == 0, d <= 2048, idx[b] in [0,K).
 */
int64_t wgs_rbf_ws_floats(int B, int n2, int d);
int wgs_rbf_fwd(const float* table, const float* alphas, const float* loggamma, float gamma,
                const int64_t* idx, const float* z, const float* scale, float* out, float* ws,
                int B, int K, int n2, int d, wgs_stream_t stream);

/* Backward of the above. `gout` [B,d] is dL/dout.  Gradients are ACCUMULATED (atomicAdd, because
 * a batch may select the same k twice) into buffers the caller has zeroed:
 *   dtable    [K, n2*d]            (required)
 *   dloggamma [K]   or NULL
 *   dalphas   [K,n2] or NULL
 *   dz        [B,d]  or NULL       (overwritten, not accumulated; the reference never needs it)
 */
int wgs_rbf_bwd(const float* table, const float* alphas, const float* loggamma, float gamma,
                const int64_t* idx, const float* z, const float* scale, const float* gout,
                const float* ws, float* dtable, float* dloggamma, float* dalphas, float* dz,
                int B, int K, int n2, int d, wgs_stream_t stream);

/* All-paths latent traversal — replaces the walk loops of traverse_latent_space.py:361-438.
 * For every start code c (n_codes) and every path k (K) integrates T steps in each direction:
 *     z <- z + sign * eps * S_k(z)          (sign = +1 for the positive walk, -1 for the negative)
 * and stores the visited codes.  One workgroup per (code, path, direction); the support set stays
 * LDS-resident across the T sequential steps.
 *   codes [n_codes, d]   start codes (z, or w when walking in W space)
 *   path  [n_codes, K, 2*T+1, d]   out: index T = start code, T+t = t-th positive step, T-t = negative
 *   shift [n_codes, K, 2*T+1, d]   out: the shift applied to reach each stored code (0 at index T)
 */
int wgs_rbf_traverse(const float* table, const float* alphas, const float* loggamma, float gamma,
                     const float* codes, float eps, int T, float* path, float* shift,
                     int n_codes, int K, int n2, int d, wgs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * fused bias + activation — replaces the pybind seam fused.fused_bias_act(input,bias,refer,act,
 * grad,alpha,scale) (models/StyleGAN2/op/fused_bias_act.cpp:11-21, kernel
 * fused_bias_act_kernel.cu:18-49).  y = act(x + b[(i/step_b) % size_b]) * scale, with the
 * reference's act*10+grad switch: act 1 linear, act 3 leaky-relu; grad 0 forward, grad 1 backward
 * gated on `ref` > 0, grad 2 -> 0.  bias / ref may be NULL (the reference passes empty tensors).
 * Extension: act 9 = tanh (grad 0) and its backward x*(1-ref^2) (grad 1), for the SNGAN / BigGAN output layer.
 */
int wgs_bias_act(const float* x, const float* bias, const float* ref, float* y, int act, int grad,
                 float alpha, float scale, int64_t size_x, int step_b, int size_b,
                 wgs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * upfirdn2d — replaces the pybind seam upfirdn2d_op.upfirdn2d(input[major,in_h,in_w,minor],
 * kernel[kh,kw], up_x,up_y,down_x,down_y,pad_x0,pad_x1,pad_y0,pad_y1) (models/StyleGAN2/op/
 * upfirdn2d.cpp:12-23, kernel upfirdn2d_kernel.cu:52-137): zero-insert upsample, pad/crop,
 * correlate with the FLIPPED kernel, decimate.  Output [major,out_h,out_w,minor] with
 * out_h = (in_h*up_y + pad_y0 + pad_y1 - kh)/down_y + 1 (same for w) — caller allocates it.
 */
int wgs_upfirdn2d(const float* x, const float* kernel, float* y, int major, int in_h, int in_w,
                  int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                  int pad_x1, int pad_y0, int pad_y1, wgs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the matrix cores (exact fp32 MFMA) — the dense contractions behind
 * F.conv2d / F.conv_transpose2d in ModulatedConv2d.forward (models/StyleGAN2/model.py:187-228),
 * the generator blocks of models/ProgGAN/model.py:35-62 and models/SNGAN/sn_gen_resnet.py:24-54, and
 * the Reconstructor's torchvision ResNet-18 / LeNet convs (lib/reconstructor.py:21-33,54-60) with
 * their autograd backward.  Activations are NHWC.
 *
 * One launch computes, for every GEMM pixel m = (b, gy, gx) of a Hg x Wg grid and every n < Co:
 *   acc = sum_{t < ntaps} sum_{k < Ci} A(b, gy*isy + dy[t], gx*isx + dx[t], k) * w[wt[t]*w_tap_stride + n*w_row_stride + k]
 *   A(b,iy,ix,k) = x[b,iy,ix,k] * (a_scale ? a_scale[b*a_ld + k] : 1)   (0 outside the image)
 *   v   = alpha * acc * (col_scale ? col_scale[b*col_ld + n] : 1) + (noise ? noise_w[0]*noise[oy*Wo+ox] : 0) + (bias ? bias[n] : 0)
 *         + (addend ? addend[b, oy>>add_ups, ox>>add_ups, n] : 0)
 *   y[b, oy, ox, n] = act ? tanh(v) : (v > 0 ? v : v*act_slope) * gain,   oy = gy*osy + oy0, ox = gx*osx + ox0
 * Plain conv: Hg=Ho, isy=stride, osy=1.  Stride-2 transposed conv / dgrad of strided conv: one launch
 * per output parity phase (osy=2, oy0=phase) with that phase's tap subset.
 * Requirements: Ci % 8 == 0, 16-B aligned rows, ntaps <= 64.
 */
typedef struct wgs_conv_desc {
    const float* x;          /* [B,Hi,Wi,Ci] */
    const float* w;          /* packed weights, see formula */
    float* y;                /* [B,Ho,Wo,Co] */
    const float* a_scale;    /* [B,Ci] or NULL  (StyleGAN2 style modulation / dgrad demod) */
    const float* col_scale;  /* [B,Co] or NULL  (StyleGAN2 demodulation) */
    const float* bias;       /* [Co] or NULL */
    const float* noise;      /* [Ho*Wo] or NULL (StyleGAN2 NoiseInjection buffer) */
    const float* noise_w;    /* device scalar (NoiseInjection.weight) or NULL */
    int32_t B, Hi, Wi, Ci, Hg, Wg, isy, isx, Ho, Wo, Co, osy, osx, oy0, ox0, ntaps;
    int32_t a_ld, col_ld;    /* row strides of a_scale / col_scale (0 = Ci / Co) */
    int32_t ups;             /* nearest-neighbour upsampling of the INPUT by 2^ups, folded into the gather: tap
                                coordinates live on the (Hi<<ups)x(Wi<<ups) grid (ProgGAN / SNGAN / BigGAN blocks) */
    int32_t add_ups, act;    /* addend is [B, Ho>>add_ups, Wo>>add_ups, Co]; act: 0 leaky (act_slope, gain), 1 tanh */
    int32_t precision;       /* 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32).  1: split-bf16 — every fp32 operand is split
                                into bf16 hi + lo while it is staged and each product block is 3 x v_mfma_f32_32x32x16_bf16
                                (hi*hi + hi*lo + lo*hi, fp32 accumulate): ~2^-16 relative per product, 5.3x the fp32 MFMA rate.
                                2: fp16 — operands rounded to fp16 (RNE) while they are staged, ONE v_mfma_f32_32x32x16_f16 per
                                product block, fp32 accumulate / demodulation / epilogue (the half dispatch of the reference's
                                ops, fused_bias_act_kernel.cu:79, upfirdn2d_kernel.cu:225): 2^-11 per operand, 16x the fp32 MFMA
                                rate; the activation operand is scaled by a power of two taken from a_amax (below) first.
                                3: fp16 x2 — as 2, but the (frozen) weights as fp16 hi + lo and 2 MFMAs per product block.
                                Shapes the 16-bit kernels do not cover (Ci % 32 != 0) silently use the exact kernel. */
    float alpha;             /* accumulator scale (0 = 1): ProgGAN WScale, BigGAN 1/sigma */
    const float* addend;     /* optional tensor added before the activation (residual / bypass), or NULL */
    int64_t w_tap_stride, w_row_stride;
    float act_slope, gain;   /* identity: 1,1;  relu: 0,1;  fused lrelu: 0.2,sqrt(2) */
    int8_t dy[64], dx[64];
    int16_t wt[64];
    const uint16_t* w_hi;    /* optional pre-split weights: 16-bit planes hi = rn(w), lo = rn(w - hi) in the layout of w: bf16 for
                                precision 1 (wgs_split_bf16), fp16 for precision 2 (hi only) / 3 (wgs_split_f16) */
    const uint16_t* w_lo;    /* (wgs_split_bf16).  With them and a workspace of 4 bytes per INPUT element, precision = 1
                                launches that take the 8-wave tiles use the LDS-DMA kernel: activations are style-modulated and
                                split into the workspace by a pre-pass, and both operands are copied global -> LDS without
                                staging registers (conv_igemm_dma.hip).  Results are bit-identical to the register-staged form. */
    float* ws;               /* optional split-K workspace (caller-owned, like every buffer) or NULL.  Launches whose */
    int64_t ws_bytes;        /* 128x128 tile count cannot fill the 256 CUs (4x4..16x16 generator layers: M = B*Hg*Wg of
                                512..2048) split the K = taps*Ci contraction over up to 16 workgroups per tile; the
                                partial tiles go to ws[split][M][Co] (plain stores, deterministic) and a second kernel
                                reduces them and applies the epilogue.  Needs 4*ksplit*M*Co bytes; too small => fewer splits. */
    const float* a_amax;     /* precision 2 / 3: optional DEVICE scalar m >= max|x| (any over-estimate).  The kernels scale the
                                activation operand x * a_scale by 2^k, k chosen so that m * a_bound * 2^k lies in [2^11, 2^12),
                                before rounding it to fp16 and multiply the accumulators by 2^-k: gradients (dgrad launches) of
                                any magnitude keep 11 significant bits and cannot overflow.  NULL: no scaling (k = 0). */
    float a_bound;           /* bound of |a_scale| (and of any linear map the caller folded into x after measuring m); 0 = 1 */
    const float* a_amax2;    /* optional second device scalar multiplied into the bound (forward launches: max |a_scale|) */
    float* y_amax;           /* every precision: optional device scalar (caller-zeroed) raised (atomic max) to max |y| of this launch
                                — chained into the next layer's a_amax, so that a forward pass in the fp16 modes cannot overflow
                                whatever the magnitude of a checkpoint's activations */
    const uint16_t* x_f16;   /* optional: the activation operand ALREADY as its fp16 plane, hi = f16_rn(x * 2^k) in the layout of x
                                with k from (a_amax, a_bound, a_amax2) as above — written by the producing kernel
                                (wgs_sg2_blur_bwd_f16) instead of the fp32 tensor.  Then x may be NULL; precision must be 2,
                                a_scale NULL, ups 0, w_hi given, Ci % 32 == 0, Co % 128 == 0: the launch runs the LDS-DMA
                                kernel without its pre-pass (same bits as the fp32 route) or fails with WGS_EINVAL. */
    /* ToRGB in the epilogue (models/StyleGAN2/model.py:270-282: the 1 x 1 modulated conv to 3 channels that reads the layer's output
       again): rgb_out[b*Ho*Wo + p][o] = rgb_scale * sum_n y[b,p,n] * rgb_s[b*rgb_ld + n] * rgb_w[o*Co + n], o = 0..2 (slot 3 = 0), written as
       one 16-byte pixel; wgs_sg2_torgb_up_fwd(x = rgb_out, C = 4, unit style, identity weight) then adds bias and the up-sampled skip.
       With rgb_out given y may be NULL (a pass that keeps nothing: the 1 GB output of StyleGAN2-256's last layer is neither written nor
       read back).  Supported for the launches the generators use it for — x_f16 operand, precision 2, stride-1 3 x 3, and either Co == 128 with
       Ci <= 128 (igemm_patch_kernel's 128 x 128 tile) or Co == 256 with Hg * Wg % 256 == 0 (igemm_dma16_kernel's 256 x 256 tile): one tile then
       holds all output channels of its pixels — or, without x_f16, a 16-bit precision with Co == 32 or 64 where wgs_conv_rgb_supported() says so
       (the few-channel kernel: StyleGAN2-1024's layers at 512^2 / 1024^2) — otherwise WGS_EINVAL. */
    float* rgb_out; const float* rgb_s; const float* rgb_w; float rgb_scale; int32_t rgb_ld;
    /* > 0: the activation operand is PixelNorm(x) (models/ProgGAN/model.py:12-18: x * rsqrt(mean_c x^2 + eps), eps = this field), applied
       while the few-channel kernel stages its input patch — the normalised tensor is never written or read (ProgGAN's 16- / 32-channel
       layers at 512^2 / 1024^2: 1 - 2 GB each).  Same bits as wgs_pixelnorm_fwd followed by the conv.  Supported where
       wgs_conv_pixelnorm_supported() says so (precision >= 1, no a_scale / x_f16, Ci in {16, 32}, a launch conv_halo16.hip takes); else WGS_EINVAL. */
    float a_pixelnorm_eps;
    double* col_stats;       /* (ABI 8) optional: the launch also accumulates the per-channel sums of its OUTPUT, sum y and sum y^2 over every
                                output pixel, into this BatchNorm scratch (WGS_BN_WS_DOUBLES(Co) doubles in wgs_bn_fwd's replica layout, zero
                                on entry): the statistics pass of the train-mode BatchNorm behind the conv comes out of the conv's epilogue
                                (wgs_bn_fwd_sums finishes it) instead of re-reading the tensor.  fp32 partial sums over <= 64 rows per lane,
                                fp64 atomics from there.  Precisions 0 / 1 / 2 / 3 through the tiled kernels, and wgs_conv_wino; a launch that would fall to a
                                kernel without the epilogue fails with WGS_EINVAL.  Not with rgb_out / x_f16. */
} wgs_conv_desc;
int wgs_conv_igemm(const wgs_conv_desc* desc, wgs_stream_t stream);
/* 1 when a launch of `desc` with a_pixelnorm_eps > 0 is covered (the operand normalised inside the few-channel kernel), else 0. */
int wgs_conv_pixelnorm_supported(const wgs_conv_desc* desc);
/* 1 when a launch of `desc` WITHOUT an x_f16 operand may carry rgb_out (ToRGB in the few-channel kernel's epilogue: a 16-bit precision,
 * Co == 32 or 64, a launch conv_halo16.hip takes), else 0. */
int wgs_conv_rgb_supported(const wgs_conv_desc* desc);
/* n launches that share every operand and differ only in (Hg, Wg, oy0, ox0, taps) — the 4 sub-pixel phases of a
 * stride-2 transposed conv (models/StyleGAN2/model.py:201-212).  Same results as n wgs_conv_igemm calls; when the
 * split-bf16 8-wave kernel covers the shape they run as ONE launch (short-K phases fill the chip together). */
int wgs_conv_igemm_multi(const wgs_conv_desc* descs, int n, wgs_stream_t stream);
/* 1 when wgs_conv_igemm_multi would run these launches as ONE merged kernel, 0 when it would issue them one by one (nothing is
 * launched): lets a profiler time the phases separately in the second case. */
int wgs_conv_igemm_multi_merges(const wgs_conv_desc* descs, int n);

/* 3x3 stride-1 'same' convolutions in fp32 through Winograd F(2x2, 3x3) on the fp32 matrix cores (conv_wino_f32.hip): the same
 * contract and epilogue as wgs_conv_igemm with precision 0, 2.25x fewer multiplies, results equal to the direct form up to fp32
 * rounding of the transforms (~1e-6 relative; 3e-6 against fp64 convolutions, no wider than the direct fp32 kernel).  Covered: all nine taps dy, dx in {-1, 0, 1} each exactly once (any weight-slab order: forward and
 * input-gradient launches alike), isy = osy = 1, ups 0, Hi = Ho % 16 == 0, Wi = Wo % 16 == 0, Ci % 16 == 0, Co % 64 == 0, act 0,
 * act_slope in [0, 1], no addend; a sample's tensors < 2 GiB (any batch).  wgs_conv_wino_supported() tells (1 / 0).
 * wgs_conv_wino_weight: U (16 * Ci * Co floats, caller-owned) = the launch's weights G g G^T in the kernel's staging order.
 * That order depends on the workgroup shape the launch takes, which depends on B, H, W, Co: wgs_conv_wino_layout() returns the
 * layout id (> 0) of a launch, and a U may be reused by any launch of the same weights, taps AND layout id (not "once per weight
 * tensor": a frozen layer called at two batch sizes may need two).  wgs_conv_wino runs the launch with it (desc->w is not read). */
int wgs_conv_wino_supported(const wgs_conv_desc* desc);
int wgs_conv_wino_layout(const wgs_conv_desc* desc);
int wgs_conv_wino_weight(const wgs_conv_desc* desc, float* U, wgs_stream_t stream);
int wgs_conv_wino(const wgs_conv_desc* desc, const float* U, wgs_stream_t stream);

/* 3x3 stride-1 'same' convolutions in split-bf16 (the arithmetic of precision 1: fp32-class, ~2^-16 per product) with the horizontal
 * taps in the Winograd form F(2, 3) and the vertical taps direct (conv_wino_bf16.hip): 12 instead of 18 products per pair of output
 * pixels and input channel, i.e. two bf16 MFMAs per direct product instead of three; V = B^T (x * style) and U = G g are formed in fp32
 * and split into bf16 hi + lo afterwards.  Same contract and epilogue as wgs_conv_igemm with precision 1 (a_scale, col_scale, noise,
 * bias, leaky-relu, alpha, y_amax); results equal the direct split-bf16 kernels' up to that arithmetic's own rounding (1e-5 against
 * fp64 convolutions).  Covered: all nine taps dy, dx in {-1, 0, 1} each exactly once (forward and input-gradient launches alike),
 * isy = osy = 1, ups 0, Hi = Ho % 8 == 0, Wi = Wo % 32 == 0, Ci % 32 == 0, Co % 128 == 0, act 0, act_slope in [0, 1], no addend /
 * x_f16 / col_stats / a_pixelnorm_eps; rgb_out (ToRGB in the epilogue, y then optional) with Co <= 512: a tile holds 128 channels of its
 * pixels, so rgb_out is [B, Ho * Wo, 4 * Co / 128] — the 16-byte slot of channel block j at floats 4 j .. 4 j + 3 holds that block's partial
 * sums (the whole sum at Co == 128); wgs_sg2_torgb_up_fwd(x = rgb_out, C = 4 * Co / 128, unit style, weight [3, C] with ones at [o, 4 j + o])
 * adds the slots, the bias and the up-sampled skip; row strides of a_scale / col_scale % 4 == 0, a sample's tensors < 2 GiB, and at
 * least 200 workgroups (B * Hi / 8 * Wi / 32 * Co / 128): wgs_conv_wino16_supported() tells (1 / 0).
 * wgs_conv_wino16_weight: U (24 * Ci * Co uint16, caller-owned) = the launch's weights G g as bf16 hi / lo planes in the kernel's
 * B-fragment order (one layout: reusable by every covered launch of the same weights and taps).  wgs_conv_wino16 runs the launch with
 * it (desc->w is not read). */
int wgs_conv_wino16_supported(const wgs_conv_desc* desc);
int wgs_conv_wino16_weight(const wgs_conv_desc* desc, uint16_t* U, wgs_stream_t stream);
int wgs_conv_wino16(const wgs_conv_desc* desc, const uint16_t* U, wgs_stream_t stream);

/* Weight gradient of a (strided) conv:  dw[co*w_row_stride + wt[t]*w_tap_stride + ci] +=
 *   sum_{b,oy,ox} dy[b,oy,ox,co] * x[b, oy*isy + dy[t], ox*isx + dx[t], ci]
 * ACCUMULATED (atomicAdd across the K splits) into the caller-zeroed `dw`.
 * ksplit <= 0 lets the library choose.  Ci % 4 == 0, Co % 4 == 0.
 */
typedef struct wgs_wgrad_desc {
    const float* x;   /* [B,Hi,Wi,Ci] */
    const float* dy;  /* [B,Ho,Wo,Co] */
    float* dw;
    int32_t B, Hi, Wi, Ci, Ho, Wo, Co, isy, isx, ntaps, ksplit;
    int64_t w_tap_stride, w_row_stride;
    int8_t dy_t[64], dx_t[64];
    int16_t wt[64];
    int32_t precision;   /* 0: exact fp32 MFMA.  1: split-bf16 x3 (operands split into bf16 hi + lo while they are transposed into the
                            [channel][pixel] LDS image, 3 bf16 MFMAs per product, fp32 accumulate: ~2^-16 per product) for
                            Ci % 64 == 0 and Co % 64 == 0, and for few input channels (Ci <= 32, ntaps * Ci >= 64, Co % 64 == 0, Wo % 8 == 0:
                            the (tap, channel) pairs flattened into the GEMM columns); other shapes use the exact kernel. */
    int32_t x_s2d;       /* != 0: x is stored space-to-depth, [B, Hi/2, Wi/2, 4*Ci] with channel (py*2 + px)*Ci + c (wgs_pack_pair_s2d);
                            Ci == 8: the ResNet stem's weight gradient without a second copy of its input */
    void* ws;            /* (ABI 8) optional scratch, ws_bytes long, 16-byte aligned, private to the stream.  With it, stride-1 3 x 3 'same'
                            launches with Co % 64 == 0, Ci % 64 == 0, Wo % 8 == 0, pixels % 16 == 0 take the direct-fragment kernel
                            (conv_wgrad_direct.hip): operand fragments straight from global memory, no LDS staging, partial tiles of the
                            pixel-range splits written here and added in split order — no atomics, bit-reproducible.  One split needs
                            Co * Ci * 9 * 4 bytes; NULL: the staged kernels (split-K atomics). */
    int64_t ws_bytes;
} wgs_wgrad_desc;
int wgs_conv_wgrad(const wgs_wgrad_desc* desc, wgs_stream_t stream);

/* hi[i] = bf16_rn(x[i]), lo[i] = bf16_rn(x[i] - hi[i]) (raw bf16 bit patterns), n % 4 == 0: the split used by precision = 1. */
int wgs_split_bf16(const float* x, uint16_t* hi, uint16_t* lo, int64_t n, wgs_stream_t stream);

/* fp16 planes for precision 2 / 3: hi[i] = f16_rn(x[i]), lo[i] = f16_rn(x[i] - hi[i]) (lo may be NULL), n % 4 == 0. */
int wgs_split_f16(const float* x, uint16_t* hi, uint16_t* lo, int64_t n, wgs_stream_t stream);

/* dst[t][ci][co] = src[co][t][ci]  (pack [Cout,T,Cin] weights for the dgrad contraction). */
int wgs_repack_w_t(const float* src, float* dst, int Co, int T, int Ci, wgs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * StyleGAN2 generator glue (models/StyleGAN2/model.py) — everything that is not a 3x3 contraction.
 */
/* PixelNorm (:9-15) over `rows` rows of length d: y = x * rsqrt(mean(x^2) + eps); and its backward. */
int wgs_pixelnorm_fwd(const float* x, float* y, int rows, int d, float eps, wgs_stream_t stream);
int wgs_pixelnorm_bwd(const float* x, const float* gy, float* gx, int rows, int d, float eps, wgs_stream_t stream);
/* The same with the leaky-relu backward of the layer that PRODUCED x folded into the store: gx *= (x > 0 ? 1 : act_slope).  In ProgGAN
 * (models/ProgGAN/model.py:35-62) the PixelNorm input of a block is the activated output of the previous block, so this replaces that
 * block's separate activation-backward pass over the tensor. */
int wgs_pixelnorm_bwd_act(const float* x, const float* gy, float* gx, int rows, int d, float eps, float act_slope, wgs_stream_t stream);
/* ... and raises the device scalar gx_amax (caller-zeroed, atomic max) to max |gx|: the magnitude bound (wgs_conv_desc.a_amax) that lets the
 * gradient conv consuming gx round it to fp16 under a power-of-two scale (ProgGAN's backward in the f16 modes). */
int wgs_pixelnorm_bwd_act_amax(const float* x, const float* gy, float* gx, float* gx_amax, int rows, int d, float eps, float act_slope, wgs_stream_t stream);

/* The whole mapping network (model.py:288-295: PixelNorm, then L x EqualLinear(d, d, lr_mul, activation='fused_lrelu')) in ONE
 * launch.  w / bias: host arrays of L device pointers ([d,d] and [d] per layer); acts: device [(L+1), B, d] — acts[0] =
 * PixelNorm(z), acts[l+1] = output of layer l (the last is the latent w; all of them are what the backward's gates need).
 * Bit-identical to wgs_pixelnorm_fwd + L x wgs_linear_fwd(..., epilogue 1).  d must be 512 (the reference's style_dim). */
int wgs_mapping_mlp_fwd(const float* z, const float* const* w, const float* const* bias, float* acts, int B, int d, int L,
                        float wscale, float lr_mul, float eps, wgs_stream_t stream);
/* Backward of those L layers in ONE launch (the PixelNorm backward stays wgs_pixelnorm_bwd): gw [B,d] = gradient w.r.t. the latent
 * acts[L]; gx [B,d] = gradient w.r.t. acts[0], i.e. L x wgs_linear_dgrad(g, w[l], gate_y = acts[l+1], slope 0.2, gain sqrt 2) for
 * l = L-1 .. 0 (fp32, different summation order: ~1e-6).  d must be 512. */
int wgs_mapping_mlp_bwd(const float* gw, const float* const* w, const float* acts, float* gx, int B, int d, int L, float wscale,
                        wgs_stream_t stream);

/* EqualLinear (:110-136) and friends, M = batch rows:
 *   y[m*ldy + n] = out_gain * epi( wscale * sum_k f(x[m*ldx + k]) * w[n*K + k] + bscale * bias[n] )
 * f = square when in_square (demodulation sum, :194);  epilogue 0 none, 1 leaky-relu(0.2)*sqrt(2)
 * (fused_leaky_relu, :127-129), 2 rsqrt(v + eps) (demod, :195).  K % 4 == 0, K <= 2048. */
int wgs_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int ldx,
                   int ldy, float wscale, float bscale, int in_square, int epilogue, float eps, float out_gain,
                   wgs_stream_t stream);
/* gx[m*ldx + k] (+)= wscale * sum_n gy[m*ldg + n] * gate(gate_y[m*ldg + n]) * w[n*K + k];
 * gate(v) = v > 0 ? gain : gain*slope when gate_y != NULL (backward of the fused leaky-relu), else 1. */
/* Up to 16 independent bias-free layers y_i = epi(wscale_i * f(x_i) W_i^T) * out_gain_i in one launch (same M, f and epi
 * as wgs_linear_fwd; K_i <= 512): the per-layer demodulation vectors of the StyleGAN2 synthesis network
 * (model.py:196-199) are 13 such GEMVs per pass. */
typedef struct wgs_linear_batch {
    int32_t n, M, in_square, epilogue;
    const float* x[16];
    const float* w[16];
    float* y[16];
    int32_t N[16], K[16], ldx[16], ldy[16];
    float wscale[16], eps[16], out_gain[16];
} wgs_linear_batch;
int wgs_linear_fwd_batch(const wgs_linear_batch* batch, wgs_stream_t stream);

int wgs_linear_dgrad(const float* gy, const float* w, const float* gate_y, float* gx, int M, int N, int K, int ldg,
                     int ldx, float wscale, float gate_slope, float gate_gain, int accumulate, wgs_stream_t stream);
/* dw[n*K+k] = sum_m gy[m*N+n] x[m*K+k];  db[n] = sum_m gy[m*N+n] (db may be NULL). */
int wgs_linear_wgrad(const float* gy, const float* x, float* dw, float* db, int M, int N, int K, wgs_stream_t stream);

/* Blur(4x4, pad (1,1)) after the transposed conv (:165,212) fused with NoiseInjection (:231-241) and
 * FusedLeakyReLU (:264): x [B,Ho+1,Wo+1,C] NHWC -> y [B,Ho,Wo,C]. */
/* y_amax: optional device scalar (caller-zeroed) raised to max |y| — the next layer's wgs_conv_desc.a_amax. */
int wgs_sg2_blur_noise_bias_act(const float* x, const float* kernel4x4, const float* noise, const float* noise_w,
                                const float* bias, float* y, float* y_amax, int B, int Ho, int Wo, int C, wgs_stream_t stream);

/* Backward of that Blur for a fp16 consumer: dt = upfirdn2d(dy, flip(kernel), pad (2,2)) (upfirdn2d.py:110-115 with the g_pad of
 * the forward's pad (1,1)), dy [B,H,W,C] fp32 -> dt_hi [B,H+1,W+1,C] as the fp16 operand plane f16_rn(dt * 2^k) of the stride-2
 * gradient conv that consumes it (wgs_conv_desc.x_f16; k from the device scalar a_amax >= max|dy| and a_bound = sum|kernel|).
 * kernel4x4: the 16 taps as wgs_upfirdn2d takes them for that call. */
int wgs_sg2_blur_bwd_f16(const float* dy, const float* kernel4x4, uint16_t* dt_hi, const float* a_amax, float a_bound,
                         int B, int H, int W, int C, wgs_stream_t stream);
/* (ABI 10) The same with dy ALREADY an fp16 plane, dy_hi = f16_rn(dy * 2^k1), k1 from a_amax alone (a_bound 1: the plane wgs_sg2_act_bwd_f16
 * writes with dy_bound = a_amax): the activation backward of an up-sampling layer (models/StyleGAN2/op/fused_act.py:19-48) then stores 2
 * instead of 4 bytes per element and this pass reads 2 instead of 4.  Fed the same values, dt_hi is bit for bit what wgs_sg2_blur_bwd_f16
 * makes of the fp32 tensor; against the fp32 route the gradient sees one more fp16 rounding in front of a 16-tap average. */
int wgs_sg2_blur_bwd_f16_x16(const uint16_t* dy_hi, const float* kernel4x4, uint16_t* dt_hi, const float* a_amax, float a_bound,
                             int B, int H, int W, int C, wgs_stream_t stream);

/* The whole up-sampling StyledConv in one launch (fp16 operand schemes): modulated stride-2 transposed 3x3 conv
 * (ModulatedConv2d.forward upsample branch, models/StyleGAN2/model.py:201-212: F.conv_transpose2d + Blur(pad (1,1))),
 * demodulation, NoiseInjection (:231-241), FusedLeakyReLU (:264) — the (2H+1)^2 intermediate never reaches HBM.
 *   t[b,u,v,n]  = alpha * col_scale[b,n] * sum_{ky,kx,k} x[b,(u-ky)/2,(v-kx)/2,k] a_scale[b,k] w[n,ky*3+kx,k]     (u-ky, v-kx even)
 *   y[b,oy,ox,n] = lrelu_0.2( sum_{i,j} flip(kernel4x4)[i][j] t[b,oy+i-1,ox+j-1,n] + noise_w[0]*noise[oy*2H+ox] + bias[n] ) * sqrt(2)
 * x [B,H,H,Ci] NHWC fp32, y [B,2H,2H,Co]; w_hi (w_lo) = the fp16 planes of the [Co,9,Ci] weights from wgs_split_f16.
 * precision 2 (fp16) or 3 (fp16 x2: here the ACTIVATION operand is split hi + lo against the single plane w_hi — two MFMAs per
 * product and the error class of wgs_conv_desc's weight split; w_lo is not read); Ci % 32 == 0, Co % 64 == 0;
 * a_amax / a_amax2 / a_bound / y_amax as in wgs_conv_desc.
 * y_f16 (optional): the kernel also writes the fp16 operand plane of the NEXT conv (its wgs_conv_desc.x_f16), with that conv's
 * style vector folded in:  y_f16[b,oy,ox,n] = f16_rn( fl32(y[b,oy,ox,n] * y_f16_scale[b*y_f16_ld + n]) * 2^k ),
 * k chosen so that  bound = (y_f16_mul * a_amax[0] + y_f16_add) * a_amax2[0]  lands in [2^11, 2^12): an A-PRIORI bound of
 * |y * scale| supplied by the caller (|t| <= sqrt(4 Ci) max|x|, blur gain 4, noise, bias, sqrt(2): see stylegan2.py), because the
 * kernel cannot know its own output maximum before it has run.  `bound` is written to *y_f16_bound: the consumer passes it as its
 * a_amax (a_bound = 1, a_amax2 = NULL, a_scale = NULL) and so derives the same k.  The plane holds the same bits the consumer
 * would have produced from the fp32 tensor itself.  With y_f16 given, y may be NULL (a pass that keeps nothing for a backward). */
typedef struct wgs_upconv_desc {
    const float* x; const void* w_hi; const void* w_lo; float* y;
    const float* a_scale; const float* col_scale; const float* bias; const float* noise; const float* noise_w;
    const float* kernel4x4;
    const float* a_amax; const float* a_amax2; float* y_amax;
    float a_bound, alpha;
    int B, H, Ci, Co, a_ld, col_ld, precision;
    uint16_t* y_f16; const float* y_f16_scale; float* y_f16_bound;
    float y_f16_mul, y_f16_add;
    int y_f16_ld;
} wgs_upconv_desc;
int wgs_sg2_upconv_blur_act(const wgs_upconv_desc* desc, wgs_stream_t stream);

/* ToRGB (:270-282): img[b,o,p] = wscale * sum_c x[b,p,c] s[b,c] w[o,c] + bias[o] + (skip ? skip[b,o,p] : 0).
 * x NHWC [B,P,C]; img/skip NCHW [B,3,P]. C power of two. */
int wgs_sg2_torgb_fwd(const float* x, const float* s, const float* w, const float* bias, const float* skip, float* img,
                      int B, int P, int C, float wscale, wgs_stream_t stream);
/* The same with the skip branch's Upsample (model.py:257-262,279-281: upfirdn2d(skip, kernel*4, up=2, pad=(2,1))) evaluated in
 * place: skip_lo [B,3,H/2,W/2] is the previous resolution's image, img [B,3,H,W]; s_ld = row stride of s (0 = C: compact). */
int wgs_sg2_torgb_up_fwd(const float* x, const float* s, int s_ld, const float* w, const float* bias, const float* skip_lo,
                         const float* up_kernel4x4, float* img, int B, int H, int W, int C, float wscale, wgs_stream_t stream);

/* Backward through one StyledConv output `out` [B,P,C] (post-activation, saved by the forward):
 *   dOut = sA*gA + sR*(sum_o drgb[b,o,p]*wR[o,c]*rscale);  dy = dOut * lrelu'(out)*sqrt(2)  -> dy [B,P,C]
 *   num[b,c] += sum_p dy*ypre  (ypre = conv output before noise/bias/activation, recovered from `out`)
 *   dsA[b,c] += sum_p out*gA ;  dsR[b,c] += sum_p out*gR       (caller zeroes num/dsA/dsR)
 * gA = UN-scaled dgrad of the consumer conv (or NULL at the last layer), drgb = image gradient (or NULL).
 * post_scale [B,C] or NULL: the STORED gradient is dy * post_scale[b,c] (this layer's demodulation vector, i.e. the
 *   A-operand factor of its dgrad conv, which then needs no a_scale); num / dsA / dsR use the un-scaled dy.
 * dy_amax: optional device scalar (caller-zeroed), raised to max |stored gradient| (atomic max) — the magnitude bound the
 *   fp16 dgrad launches take as wgs_conv_desc.a_amax.   s_ld: row stride of sA and sR (0 = C: compact [B,C] arrays). */
int wgs_sg2_act_bwd(const float* out, const float* gA, const float* sA, const float* drgb, const float* wR,
                    const float* sR, float rscale, const float* noise, const float* noise_w, const float* bias,
                    float* dy, float* num, float* dsA, float* dsR, const float* post_scale, float* dy_amax, int B, int P,
                    int C, int s_ld, wgs_stream_t stream);
/* The same, storing the gradient ONLY as the fp16 operand plane of the dgrad conv that consumes it (wgs_conv_desc.x_f16 with
 * a_amax = dy_bound, a_bound = 1): dy_hi [B,P,C] = f16_rn(dy * post_scale * 2^k), k from the device scalar dy_bound >= max |dy *
 * post_scale| (any over-estimate up to ~2^7; wgs_sg2_dy_bound computes one from the operands' magnitudes).  The fp32 tensor
 * is not written: 2 bytes instead of 4 per element here, 2 instead of 4 read by the conv, and no conversion in the conv. */
int wgs_sg2_act_bwd_f16(const float* out, const float* gA, const float* sA, const float* drgb, const float* wR,
                        const float* sR, float rscale, const float* noise, const float* noise_w, const float* bias,
                        uint16_t* dy_hi, const float* dy_bound, float* num, float* dsA, float* dsR, const float* post_scale,
                        float* dy_amax, int B, int P, int C, int s_ld, wgs_stream_t stream);
/* bound[0] = sqrt(2) * max_{b,c} post_scale[b,c] * (|sA[b,c]| * gA_amax + |sR[b,c]| * rscale * drgb_amax * drgb_factor *
 * sum_o |wR[o,c]|): an upper bound of |dy * post_scale| of the wgs_sg2_act_bwd launch with these operands, from the device
 * scalars gA_amax >= max|gA| (the producing conv's y_amax) and drgb_amax * drgb_factor >= max|drgb|; either source may be NULL. */
int wgs_sg2_dy_bound(const float* gA_amax, const float* sA, const float* drgb_amax, float drgb_factor, const float* wR,
                     const float* sR, float rscale, const float* post_scale, float* bound, int B, int C, int s_ld,
                     wgs_stream_t stream);
/* ds[b,c] += sum_p x[(x_batched ? b : 0), p, c] * g[b,p,c] */
int wgs_xg_reduce(const float* x, int x_batched, const float* g, float* ds, int B, int P, int C, wgs_stream_t stream);
/* dstyle[b*ld_out + i] = dsdir[b*Ci + i] - s[b*ld_s + i]*scale2 * sum_o num[b,o]*demod[b,o]^2*wsq[o,i]
 * (demod == NULL: no demodulation, dstyle = dsdir). */
int wgs_sg2_style_grad(const float* num, const float* demod, const float* s, const float* dsdir, const float* wsq,
                       float scale2, float* dstyle, int B, int Co, int Ci, int ld_s, int ld_out, wgs_stream_t stream);
/* Up to 24 of them in one launch (every modulated conv and ToRGB of a synthesis backward pass; B, ld_s, ld_out shared). */
typedef struct wgs_style_grad_batch {
    int32_t n, B, ld_s, ld_out;
    const float* num[24]; const float* demod[24]; const float* s[24]; const float* dsdir[24]; const float* wsq[24];
    float* dstyle[24];
    int32_t Co[24], Ci[24];
    float scale2[24];
} wgs_style_grad_batch;
int wgs_sg2_style_grad_batch(const wgs_style_grad_batch* batch, wgs_stream_t stream);
/* wsq[o,i] = sum_t w[o,t,i]^2 for packed [Co,T,Ci] weights. */
int wgs_sg2_wsq(const float* w_packed, float* wsq, int Co, int T, int Ci, wgs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Reconstructor glue (lib/reconstructor.py + torchvision ResNet-18 BasicBlocks / LeNet), NHWC.
 */
/* torch.cat([x1, x2], dim=1) (lib/reconstructor.py:73,77) of two NCHW [B,c,H,W] images into one NHWC
 * [B,H,W,Cp] tensor, channels [x1 | x2 | zero padding] (Cp >= 2c, Cp % 8 == 0 for the conv kernel);
 * and the gradient of that w.r.t. x1 / x2 (either may be NULL). */
int wgs_pack_pair_nhwc(const float* x1, const float* x2, float* y, int B, int c, int HW, int Cp, wgs_stream_t stream);
int wgs_unpack_pair_grad(const float* dy, float* d1, float* d2, int B, int c, int HW, int Cp, wgs_stream_t stream);
/* The same concatenation in SPACE-TO-DEPTH form: y [B, H/2, W/2, 32], channel (py*2 + px)*8 + j = channel j (x1 | x2 | zeros, 2c <= 8) of
 * image pixel (2 oy + py, 2 ox + px); its gradient; and the matching re-indexing of the ResNet stem's weights (torchvision resnet18
 * conv1: 7 x 7, stride 2, pad 3 — lib/reconstructor.py:54-63): w [Co, 49, Ci] -> ws [Co, 16, 32], tap r*4 + s = block offset
 * (r - 2, s - 2), so that conv7x7/2(x) == conv over the 4 x 4 block window of the s2d tensor with ws (zeros where 2r + py - 1 or
 * 2s + px - 1 leaves 0..6); back != 0: the reverse gather ws -> w (a weight gradient computed in the s2d form). */
int wgs_pack_pair_s2d(const float* x1, const float* x2, float* y, int B, int c, int H, int W, wgs_stream_t stream);
int wgs_unpack_pair_s2d_grad(const float* dys, float* d1, float* d2, int B, int c, int H, int W, wgs_stream_t stream);
int wgs_stem_weight_s2d(const float* src, float* dst, int Co, int Ci, int back, wgs_stream_t stream);

/* nn.BatchNorm2d / BatchNorm1d on [N rows, C] (N = B*H*W) fused with the residual add and ReLU of a
 * BasicBlock:  y = relu?( (x - mean)*invstd*gamma + beta (+ residual) ).
 * train != 0: batch statistics (biased variance), running stats updated with `momentum` and the unbiased
 * variance, num_batches_tracked += 1 (all three may be NULL); else running statistics are used.
 * save_mean / save_invstd [C] are written for the backward; ws = WGS_BN_WS_DOUBLES(C) doubles of scratch: 32 replicas of the 2*C
 * partial sums (so that the reduction's fp64 atomics do not all hit the same addresses; min(32, 2048 / C) of them are used).  ws MUST BE ZERO ON ENTRY and is left zero
 * on exit (zero it once when allocating it; one buffer serves any sequence of wgs_bn_fwd / wgs_bn_bwd / wgs_colsum calls on one
 * stream, with any C): the launch that sums the replicas zeroes them again, so no reduction needs a memset.  A buffer that is NOT zero
 * (never zeroed, a call cut short, two streams sharing it) gives wrong statistics without an error; WGS_CHECK_WS=1 in the environment
 * makes every call verify it first (synchronously: debugging only).  C % 4 == 0.  The fused launches (wgs_bn_fwd_fused / wgs_bn_bwd_fused)
 * zero max(4096, 2 * C) doubles of the buffer they leave clean whatever the current C >= 64, so one pair sized for the widest layer serves
 * any sequence of widths in [64, 2048]; calls with C < 64 or C > 2048 must not share a pair with calls of another width. */
#define WGS_BN_WS_DOUBLES(C) (64 * (C))
int wgs_bn_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y, float* save_mean,
               float* save_invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, double* ws,
               int64_t N, int C, float eps, float momentum, int relu, int train, wgs_stream_t stream);
/* Train-mode wgs_bn_fwd whose statistics pass has ALREADY run: ws holds sum x and sum x^2 of x's N rows in the replica layout, written
 * by the epilogue of the conv that produced x (wgs_conv_desc.col_stats = ws).  Two launches (finalise, apply) instead of three, and x is
 * read once instead of twice.  ws is left zero as by wgs_bn_fwd. */
int wgs_bn_fwd_sums(const float* x, const float* gamma, const float* beta, const float* residual, float* y, float* save_mean,
                    float* save_invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, double* ws,
                    int64_t N, int C, float eps, float momentum, int relu, wgs_stream_t stream);
/* The same in ONE launch over a PAIR of scratch buffers (round 5): every workgroup of the apply kernel finishes the statistics from
 * ws_sums (<= 32 KB: the replica count shrinks with C) in its prologue, workgroup 0 writes save_mean / save_invstd / the running
 * statistics, and ws_zero — the other buffer of the pair, the one the NEXT producer will accumulate into — is left zero.  ws_sums itself
 * stays as it is: it is the ws_zero of the next call.  Contract for the caller: alternate the two buffers producer by producer, both zero
 * before the first one (warpedganspace_amd/reconstructor.py: _BNScratch). */
int wgs_bn_fwd_fused(const float* x, const float* gamma, const float* beta, const float* residual, float* y, float* save_mean,
                     float* save_invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, const double* ws_sums,
                     double* ws_zero, int64_t N, int C, float eps, float momentum, int relu, wgs_stream_t stream);
/* Backward: g = (dyA + (dyB ? dyB : 0)) * (out ? out > 0 : 1)   [out = the saved post-ReLU output]
 *   dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)) (train) ;  dgamma = sum g*xhat ; dbeta = sum g ;
 *   dres (optional) = g  (gradient of the residual branch).
 * dgamma and dbeta are REQUIRED in train mode: the reduction launch OVERWRITES them with the two sums (fp64 partials rounded to fp32;
 * they cannot be accumulate-into buffers) and the input-gradient launch reads them back; in eval mode pass both or neither (neither: no
 * reduction is launched).  ws as in wgs_bn_fwd. */
int wgs_bn_bwd(const float* x, const float* dyA, const float* dyB, const float* out, const float* save_mean,
               const float* save_invstd, const float* gamma, float* dx, float* dres, float* dgamma, float* dbeta,
               double* ws, int64_t N, int C, int train, wgs_stream_t stream);
/* Train-mode wgs_bn_bwd in two launches over the scratch pair of wgs_bn_fwd_fused: the reduction accumulates into ws (zero on entry), the
 * input-gradient launch finishes the sums in its prologue (workgroup 0 writes dgamma / dbeta) and leaves ws_zero zero; ws stays dirty. */
int wgs_bn_bwd_fused(const float* x, const float* dyA, const float* dyB, const float* out, const float* save_mean,
                     const float* save_invstd, const float* gamma, float* dx, float* dres, float* dgamma, float* dbeta,
                     double* ws, double* ws_zero, int64_t N, int C, wgs_stream_t stream);

/* nn.MaxPool2d(k, stride s, padding p) on NHWC; idx [B,Ho,Wo,C] bytes = window position of the first
 * maximum (torch tie-breaking); backward gathers through idx. */
int wgs_maxpool_fwd(const float* x, float* y, unsigned char* idx, int B, int Hi, int Wi, int C, int k, int s, int p,
                    wgs_stream_t stream);
int wgs_maxpool_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int Hi, int Wi, int C, int k, int s, int p,
                    wgs_stream_t stream);
/* AdaptiveAvgPool2d(1) / features.mean([-1,-2]) (lib/reconstructor.py:74,78): [B,P,C] -> [B,C]. */
int wgs_avgpool_fwd(const float* x, float* y, int B, int P, int C, wgs_stream_t stream);
int wgs_avgpool_bwd(const float* dy, float* dx, int B, int P, int C, wgs_stream_t stream);
/* Backward of nn.Upsample(scale_factor=2, nearest) on NHWC: dx[b,y,x,:] = sum of dy[b,2y..2y+1,2x..2x+1,:]
 * (models/ProgGAN/model.py:53, models/SNGAN/sn_gen_resnet.py:37,45, BigGAN GBlock). dy [B,2H,2W,C] -> dx [B,H,W,C]. */
int wgs_upsample2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, wgs_stream_t stream);
/* out[c] = sum over rows of x[N,C] (conv bias gradient); ws = WGS_BN_WS_DOUBLES(C) doubles, zero on entry / exit as above. */
int wgs_colsum(const float* x, float* out, double* ws, int64_t N, int C, wgs_stream_t stream);

/* Loss of lib/trainer.py:245-249 and the statistics of :257-261:
 *   ce = CrossEntropy(logits, target) (mean), l1 = mean|mag_pred - mag_target|, total = lambda_cls*ce + lambda_reg*l1
 *   stats[0..3] = (ce, l1, total, accuracy);  argmax[b] = first index of max logits[b,:] (int64)
 *   dlogits = d total / d logits, dmag = d total / d mag_pred.   ws: 2*B floats. */
int wgs_ce_l1_loss(const float* logits, const int64_t* target, const float* mag_pred, const float* mag_target, float lambda_cls,
                   float lambda_reg, float* dlogits, float* dmag, float* stats, int64_t* argmax, float* ws, int B, int K,
                   wgs_stream_t stream);

/* torch.optim.Adam (defaults: betas given, eps, no weight decay / amsgrad) on one flat buffer, update
 * number `step` (1-based); the gradient is multiplied by grad_scale first (1/world after an all-reduce sum). */
int wgs_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, int step, float grad_scale, wgs_stream_t stream);

/* One training step's batch drawn in HBM by ONE launch (replaces the host-side sampling of lib/trainer.py:195-221 and lib/aux.py:39-53;
 * SURVEY section 8b `sample_step`):
 *   z [B, d] ~ N(0, 1), truncated to [-trunc, trunc] when 0 < trunc != 1 (inverse CDF; trunc <= 0 or == 1: plain normal);
 *   idx [B] (int64) ~ U{0 .. K-1};
 *   mag [B]: B entries of the pool [neg_0 .. neg_{B-1}, pos_0 .. pos_{B-1}] (neg_i = (lo - hi) u - lo, pos_i = (lo - hi) u' + hi) drawn
 *            WITHOUT replacement with weights 0 .. 2B-1, in torch.multinomial's order (the reference's arange-weighted draw, :218-221).
 * Counter-based generator (Philox4x32-10): the values are a pure function of (seed, step) — stateless and reproducible; a caller
 * increments `step` per draw and gives every rank its own seed.  B <= 1024. */
int wgs_sample_step(float* z, int64_t* idx, float* mag, int B, int d, int K, float lo, float hi, float trunc, uint64_t seed, uint64_t step,
                    wgs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * BigGAN generator glue (models/BigGAN/layers.py).
 */
/* Eval-mode (class-conditional) BatchNorm + ReLU (ccbn.forward :303-322, bn.forward :358-363, GBlock :393-405):
 *   y[b,p,c] = relu?( x[b,p,c]*scale[b,c] + shift[b,c] ),  x NHWC [B,P,C];
 * backward: g' = g*(y > 0), dx = g'*scale, dscale[b,c] += sum_p g'*x, dshift[b,c] += sum_p g' (caller zeroes both). */
int wgs_affine_relu_fwd(const float* x, const float* scale, const float* shift, float* y, int B, int P, int C, int relu,
                        wgs_stream_t stream);
int wgs_affine_relu_bwd(const float* x, const float* y, const float* g, const float* scale, float* dx, float* dscale,
                        float* dshift, int B, int P, int C, int relu, wgs_stream_t stream);
/* Row softmax and its backward (Attention.forward :163: F.softmax(theta^T phi, -1)). */
int wgs_softmax_rows_fwd(const float* x, float* y, int64_t rows, int n, wgs_stream_t stream);
int wgs_softmax_rows_bwd(const float* y, const float* dy, float* dx, int64_t rows, int n, wgs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Self-attention core of BigGAN's non-local block — replaces the two torch.bmm + F.softmax of Attention.forward
 * (models/BigGAN/layers.py:157-166) and their autograd backward, batched over the samples and fused:
 *     beta[b,q,:] = softmax_k( theta[b,q,:] . phi[b,k,:] ),      o[b,q,:] = sum_k beta[b,q,k] * g[b,k,:]
 * theta [B,Pq,c8], phi [B,Pk,c8], g [B,Pk,c2], o [B,Pq,c2]: NHWC rows (pixel-major, channels contiguous).  The [B,Pq,Pk]
 * score / attention tensors are never written; `lse` [B,Pq] (log-sum-exp of every score row) is the only state the backward
 * needs besides the operands and o.  Exact fp32 (f32-input MFMA).  Supported: Pq % 128 == 0, Pk % 64 == 0 and
 * (c8, c2) in {(24, 96), (48, 192), (96, 384)} (= ch in {192, 384, 768}); wgs_attn_supported() tells.
 * wgs_attn_bwd: d_o = dL/do [B,Pq,c2]; ws = B*Pq floats of scratch; dtheta / dphi / dg are overwritten.
 */
int wgs_attn_supported(int B, int Pq, int Pk, int c8, int c2);
int wgs_attn_fwd(const float* theta, const float* phi, const float* g, float* o, float* lse, int B, int Pq, int Pk, int c8, int c2,
                 wgs_stream_t stream);
int wgs_attn_bwd(const float* theta, const float* phi, const float* g, const float* o, const float* lse, const float* d_o, float* ws,
                 float* dtheta, float* dphi, float* dg, int B, int Pq, int Pk, int c8, int c2, wgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WGS_H */
