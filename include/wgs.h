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

#ifdef __cplusplus
}
#endif
#endif /* WGS_H */
