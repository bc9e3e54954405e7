// Reconstructor-side streaming kernels (everything in lib/reconstructor.py / torchvision ResNet-18 /
// LeNet that is not a dense contraction) plus the loss (lib/trainer.py:245-249,257-258) and Adam
// (lib/trainer.py:153-156,253-254).  NHWC activations viewed as [rows, C]; all HBM-bound: float4
// accesses over C, per-thread fp32 partials folded into fp64 block partials and fp64 atomics for the
// batch statistics (train-mode BatchNorm normalises over B*H*W, up to 5e5 rows per channel).
#include "wgs_common.h"
#include "../../include/wgs.h"

namespace {

// ---- (img, img_shifted) NCHW pair -> one NHWC tensor with Cp >= 2c channels (zero padded) ----------
__global__ __launch_bounds__(256) void pack_pair_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                        float* __restrict__ y, int B, int c, int HW, int Cp) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int b = (int)(e / HW);
        const int p = (int)(e % HW);
        float* o = y + e * Cp;
        for (int j = 0; j < Cp; ++j) {
            float v = 0.f;
            if (j < c) v = x1[((size_t)b * c + j) * HW + p];
            else if (j < 2 * c) v = x2[((size_t)b * c + (j - c)) * HW + p];
            o[j] = v;
        }
    }
}
// ---- the same in SPACE-TO-DEPTH form: y [B, H/2, W/2, 32], channel (py*2 + px)*8 + j = channel j of pixel (2 oy + py, 2 ox + px)
// (j < c: x1, j < 2c: x2, else 0).  A 7 x 7 stride-2 conv over [B,H,W,2c] is a 4 x 4-window stride-1 conv over this tensor
// (wgs_stem_weight_s2d): the ResNet stem then runs through the few-channel halo kernel instead of a gather over 6 channels.
__global__ __launch_bounds__(256) void pack_pair_s2d_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                            float* __restrict__ y, int B, int c, int H, int W) {
    const int Wo = W >> 1, Ho = H >> 1;
    const int64_t total = (int64_t)B * Ho * Wo * 4;          // one thread per (output pixel, sub-pixel): 8 channels = two float4
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int sp = (int)(e & 3);
        const int64_t op = e >> 2;
        const int ox = (int)(op % Wo), oy = (int)((op / Wo) % Ho), b = (int)(op / ((int64_t)Wo * Ho));
        const int iy = 2 * oy + (sp >> 1), ix = 2 * ox + (sp & 1);
        const size_t pix = (size_t)iy * W + ix, HW = (size_t)H * W;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = 0.f;
            if (j < c) v[j] = x1[((size_t)b * c + j) * HW + pix];
            else if (j < 2 * c) v[j] = x2[((size_t)b * c + (j - c)) * HW + pix];
        }
        float4* o = reinterpret_cast<float4*>(y + (size_t)op * 32 + sp * 8);
        o[0] = make_float4(v[0], v[1], v[2], v[3]);
        o[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}
// gradient of the above w.r.t. x1 / x2 (either may be NULL): dys [B, H/2, W/2, 32]
__global__ __launch_bounds__(256) void unpack_pair_s2d_kernel(const float* __restrict__ dys, float* __restrict__ d1,
                                                              float* __restrict__ d2, int B, int c, int H, int W) {
    const int64_t total = (int64_t)B * H * W;                // one thread per image pixel (coalesced NCHW stores)
    const size_t HW = (size_t)H * W;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int ix = (int)(e % W), iy = (int)((e / W) % H), b = (int)(e / HW);
        const float* g = dys + (((size_t)b * (H >> 1) + (iy >> 1)) * (W >> 1) + (ix >> 1)) * 32 + ((iy & 1) * 2 + (ix & 1)) * 8;
        const float4 g0 = reinterpret_cast<const float4*>(g)[0], g1 = reinterpret_cast<const float4*>(g)[1];
        const float v[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const size_t pix = (size_t)iy * W + ix;
        for (int j = 0; j < c; ++j) {
            if (d1) d1[((size_t)b * c + j) * HW + pix] = v[j];
            if (d2) d2[((size_t)b * c + j) * HW + pix] = v[c + j];
        }
    }
}
// stem weights w [Co, 49, Ci] (7 x 7 taps ky*7 + kx, Ci = 2c <= 8 input channels) -> ws [Co, 16, 32]: tap r*4 + s of the 4 x 4 window
// (block offsets r - 2, s - 2), channel (py*2 + px)*8 + j  <-  ky = 2r + py - 1, kx = 2s + px - 1 (zero where that falls outside 0..6
// or j >= Ci).  back != 0: the reverse gather, w[co][ky*7 + kx][j] = ws[co][...] (the weight gradient computed in the s2d form).
__global__ __launch_bounds__(256) void stem_weight_s2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int Co, int Ci, int back) {
    if (!back) {
        const int e = blockIdx.x * 256 + threadIdx.x;
        if (e >= Co * 16 * 32) return;
        const int n = e & 31, t = (e >> 5) & 15, co = e >> 9;
        const int j = n & 7, px = (n >> 3) & 1, py = n >> 4;
        const int ky = 2 * (t >> 2) + py - 1, kx = 2 * (t & 3) + px - 1;
        dst[e] = (j < Ci && (unsigned)ky < 7u && (unsigned)kx < 7u) ? src[((size_t)co * 49 + ky * 7 + kx) * Ci + j] : 0.f;
    } else {
        const int e = blockIdx.x * 256 + threadIdx.x;
        if (e >= Co * 49 * Ci) return;
        const int j = e % Ci, tap = (e / Ci) % 49, co = e / (Ci * 49);
        const int ky = tap / 7, kx = tap % 7;
        const int r = (ky + 1) >> 1, py = (ky + 1) & 1, sx = (kx + 1) >> 1, px = (kx + 1) & 1;
        dst[e] = src[((size_t)co * 16 + r * 4 + sx) * 32 + (py * 2 + px) * 8 + j];
    }
}
// gradient of the above w.r.t. x1 / x2 (either may be NULL)
__global__ __launch_bounds__(256) void unpack_pair_kernel(const float* __restrict__ dy, float* __restrict__ d1,
                                                          float* __restrict__ d2, int B, int c, int HW, int Cp) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int b = (int)(e / HW);
        const int p = (int)(e % HW);
        const float* g = dy + e * Cp;
        for (int j = 0; j < c; ++j) {
            if (d1) d1[((size_t)b * c + j) * HW + p] = g[j];
            if (d2) d2[((size_t)b * c + j) * HW + p] = g[c + j];
        }
    }
}

// ---- per-channel reductions over rows of an [N, C] tensor ---------------------------------------------
// MODE 0: s1 = sum x, s2 = sum x^2                                   (BN forward statistics)
// MODE 1: g = (dyA + dyB) * (out > 0),  s1 = sum g, s2 = sum g*xhat    (BN backward statistics)
// MODE 2: s1 = sum x                                                  (bias gradient)
// ws: double[wgs_bn_nrep(C)][2*C], ZERO ON ENTRY AND LEFT ZERO ON EXIT (the caller zeroes it once, when it allocates it); workgroup b
// accumulates into replica b % nrep with fp64 atomics (hundreds of workgroups adding to the same 2*C addresses serialise in the L2
// atomic units: replicas cut that 32x); the small second launch (bn_finalize_kernel / ws_collapse_kernel) sums the replicas, writes what
// the consumer needs (mean / invstd / running statistics, or dbeta / dgamma) and ZEROES the replicas again — no memset launch per
// reduction (40 per training step over the Reconstructor's 20 BatchNorms).
// (Tried and dropped, round 4: the whole reduction in ONE launch, the last workgroup to finish — ticket counter — finalising.  The
// agent-scope fence every workgroup needs in front of its ticket writes back its XCD's L2 on this 8-XCD part: 112 us instead of 72 us
// for the stem's BatchNorm forward, the training step 27.4 -> 29.5 ms.  Kernel boundaries do that write-back once.)
template <int MODE>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dyA,
                                                          const float* __restrict__ dyB, const float* __restrict__ out,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          double* __restrict__ ws, int64_t N, int C, int rows_per_block) {
    __shared__ double red[2][256][4];
    const int c4n = C >> 2;
    const int tpr = c4n < 256 ? c4n : 256;
    const int rpi = 256 / tpr;
    const int cl = threadIdx.x % tpr, sub = threadIdx.x / tpr;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = (r_begin + rows_per_block < N) ? r_begin + rows_per_block : N;
    const bool active = sub < rpi;
    for (int c = cl * 4; c < C; c += tpr * 4) {
        float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
        float4 mu = a1, is = a1;
        if (MODE == 1) {
            mu = *reinterpret_cast<const float4*>(mean + c);
            is = *reinterpret_cast<const float4*>(invstd + c);
        }
        if (active) {
            // rows r, r+rpi, r+2rpi, r+3rpi per trip: four independent load groups in flight per thread
            auto accum = [&](int64_t r) {
                const size_t off = (size_t)r * C + c;
                if (MODE == 0) {
                    const float4 v = *reinterpret_cast<const float4*>(x + off);
                    a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
                    a2.x = fmaf(v.x, v.x, a2.x); a2.y = fmaf(v.y, v.y, a2.y);
                    a2.z = fmaf(v.z, v.z, a2.z); a2.w = fmaf(v.w, v.w, a2.w);
                } else if (MODE == 2) {
                    const float4 v = *reinterpret_cast<const float4*>(x + off);
                    a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
                } else {
                    float4 g = *reinterpret_cast<const float4*>(dyA + off);
                    if (dyB) {
                        const float4 g2 = *reinterpret_cast<const float4*>(dyB + off);
                        g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
                    }
                    if (out) {
                        const float4 o = *reinterpret_cast<const float4*>(out + off);
                        g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
                        g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
                    }
                    const float4 v = *reinterpret_cast<const float4*>(x + off);
                    a1.x += g.x; a1.y += g.y; a1.z += g.z; a1.w += g.w;
                    a2.x = fmaf(g.x, (v.x - mu.x) * is.x, a2.x); a2.y = fmaf(g.y, (v.y - mu.y) * is.y, a2.y);
                    a2.z = fmaf(g.z, (v.z - mu.z) * is.z, a2.z); a2.w = fmaf(g.w, (v.w - mu.w) * is.w, a2.w);
                }
            };
            int64_t r = r_begin + sub;
            for (; r + 3 * (int64_t)rpi < r_end; r += 4 * (int64_t)rpi) {
                accum(r); accum(r + rpi); accum(r + 2 * (int64_t)rpi); accum(r + 3 * (int64_t)rpi);
            }
            for (; r < r_end; r += rpi) accum(r);
        }
        __syncthreads();
        red[0][threadIdx.x][0] = a1.x; red[0][threadIdx.x][1] = a1.y; red[0][threadIdx.x][2] = a1.z; red[0][threadIdx.x][3] = a1.w;
        red[1][threadIdx.x][0] = a2.x; red[1][threadIdx.x][1] = a2.y; red[1][threadIdx.x][2] = a2.z; red[1][threadIdx.x][3] = a2.w;
        __syncthreads();
        if (sub == 0) {
            double t1[4] = {0, 0, 0, 0}, t2[4] = {0, 0, 0, 0};
            for (int s = 0; s < rpi; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) { t1[q] += red[0][s * tpr + cl][q]; t2[q] += red[1][s * tpr + cl][q]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double* wr = ws + (size_t)(blockIdx.x % (unsigned)wgs_bn_nrep(C)) * 2 * C;
                unsafeAtomicAdd(wr + c + q, t1[q]);
                if (MODE != 2) unsafeAtomicAdd(wr + C + c + q, t2[q]);
            }
        }
    }
}

// totals of a MODE 1 / 2 reduction: out1[c] = sum of replica s1 (dbeta / the column sums), out2[c] = s2 (dgamma; may be NULL);
// the replicas are zeroed again
__global__ __launch_bounds__(256) void ws_collapse_kernel(double* __restrict__ ws, int C, float* __restrict__ out1, float* __restrict__ out2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = 2 * C;
    if (i >= n) return;
    double t = 0.0;
    const int nrep = wgs_bn_nrep(C);
    for (int r = 0; r < nrep; ++r) { t += ws[(size_t)r * n + i]; ws[(size_t)r * n + i] = 0.0; }
    if (i < C) { if (out1) out1[i] = (float)t; }
    else if (out2) out2[i - C] = (float)t;
}

// BN forward finalize (train mode): batch mean / biased var -> mean, invstd; running stats as nn.BatchNorm
// (momentum 0.1, unbiased variance for the running estimate), num_batches_tracked += 1.
__global__ void bn_finalize_kernel(const double* __restrict__ ws, float* __restrict__ mean, float* __restrict__ invstd,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   int64_t* __restrict__ nbt, int64_t N, int C, float eps, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) nbt[0] += 1;
    if (c >= C) return;
    // sums the replicas itself (same order as ws_collapse_kernel): one launch less per BatchNorm forward
    double s1 = 0.0, s2 = 0.0;
    const int nrep = wgs_bn_nrep(C);
    for (int r = 0; r < nrep; ++r) {
        double* pr = const_cast<double*>(ws) + (size_t)r * 2 * C;
        s1 += pr[c]; s2 += pr[C + c];
        pr[c] = 0.0; pr[C + c] = 0.0;               // left zero for the next reduction (no memset launch)
    }
    const double m = s1 / (double)N;
    double var = s2 / (double)N - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = N > 1 ? var * (double)N / (double)(N - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}
// eval mode: mean = running_mean, invstd = rsqrt(running_var + eps)
__global__ void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                     float* __restrict__ mean, float* __restrict__ invstd, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = running_mean[c];
    invstd[c] = 1.f / sqrtf(running_var[c] + eps);
}

// y = relu?( (x - mean)*invstd*gamma + beta (+ residual) )
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ res,
                                                       float* __restrict__ y, int64_t N, int C, int relu) {
    const int c4n = C >> 2;
    const int64_t total = N * c4n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % c4n) * 4;
        const float4 v = reinterpret_cast<const float4*>(x)[e];
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
        const float4 be = *reinterpret_cast<const float4*>(beta + c);
        float4 o;
        o.x = (v.x - mu.x) * is.x * ga.x + be.x; o.y = (v.y - mu.y) * is.y * ga.y + be.y;
        o.z = (v.z - mu.z) * is.z * ga.z + be.z; o.w = (v.w - mu.w) * is.w * ga.w + be.w;
        if (res) {
            const float4 r = reinterpret_cast<const float4*>(res)[e];
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        reinterpret_cast<float4*>(y)[e] = o;
    }
}

// ---- round 5: the statistics are finished in the PROLOGUE of the apply kernels (every workgroup sums the <= 32 KB of replicas for all C
// channels into LDS; workgroup 0 also writes what the finalise / collapse launches wrote), and the scratch is a PAIR: this launch reads
// `ws`, and leaves the OTHER scratch `wz` zero for the next producer (it was dirtied two producers ago; nothing else touches it now) —
// so neither a finalise / collapse launch nor a memset sits between a reduction and its apply.
// Doubles of the OTHER scratch buffer that a fused apply launch zeroes.  The footprint of a producer is wgs_bn_nrep(C') * 2 * C' doubles of ITS
// channel count: <= 4096 for 64 <= C' <= 2048, 2 * C' above.  A scratch pair is shared by BatchNorms of different widths (ResNet-18: 64 .. 512), so
// the extent must not depend on the CURRENT C (zeroing nrep(C) * 2 * C left a tail of a wider / narrower predecessor's sums — silently
// wrong statistics).  Below 64 channels a buffer holds only WGS_BN_WS_DOUBLES(C) = 64 * C doubles, above 2048 the footprint is 2 * C: such calls must
// not share a pair with other widths (include/wgs.h).
__device__ __forceinline__ int bn_zero_extent(int C) { return C >= 64 ? (2 * C > 4096 ? 2 * C : 4096) : wgs_bn_nrep(C) * 2 * C; }

__global__ __launch_bounds__(256) void bn_apply_fused_kernel(const float* __restrict__ x, const double* __restrict__ ws, double* __restrict__ wz,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ res, float* __restrict__ y,
                                                             float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var,
                                                             int64_t* __restrict__ nbt, int64_t N, int C, float eps, float momentum, int relu) {
    extern __shared__ float bn_sm[];            // [C] scale = invstd * gamma | [C] shift = beta - mean * scale
    float* sc = bn_sm;
    float* sh = bn_sm + C;
    const int nrep = wgs_bn_nrep(C);
    for (int c = threadIdx.x; c < C; c += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int r = 0; r < nrep; ++r) { s1 += ws[(size_t)r * 2 * C + c]; s2 += ws[(size_t)r * 2 * C + C + c]; }
        const double m = s1 / (double)N;
        double var = s2 / (double)N - m * m;
        if (var < 0.0) var = 0.0;
        const float mf = (float)m, isf = (float)(1.0 / sqrt(var + (double)eps));
        sc[c] = isf; sh[c] = mf;
        if (blockIdx.x == 0) {
            save_mean[c] = mf; save_invstd[c] = isf;
            if (running_mean) {
                const double unb = N > 1 ? var * (double)N / (double)(N - 1) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
            }
        }
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0 && nbt) nbt[0] += 1;
        for (int i = threadIdx.x; i < bn_zero_extent(C); i += 256) wz[i] = 0.0;
    }
    __syncthreads();
    const int c4n = C >> 2;
    const int64_t total = N * c4n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % c4n) * 4;
        const float4 v = reinterpret_cast<const float4*>(x)[e];
        const float4 is = *reinterpret_cast<const float4*>(sc + c);
        const float4 mu = *reinterpret_cast<const float4*>(sh + c);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
        const float4 be = *reinterpret_cast<const float4*>(beta + c);
        float4 o;      // (the same expression, in the same order, as bn_apply_kernel)
        o.x = (v.x - mu.x) * is.x * ga.x + be.x; o.y = (v.y - mu.y) * is.y * ga.y + be.y;
        o.z = (v.z - mu.z) * is.z * ga.z + be.z; o.w = (v.w - mu.w) * is.w * ga.w + be.w;
        if (res) {
            const float4 r = reinterpret_cast<const float4*>(res)[e];
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        reinterpret_cast<float4*>(y)[e] = o;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_fused_kernel(const float* __restrict__ x, const float* __restrict__ dyA,
                                                                 const float* __restrict__ dyB, const float* __restrict__ out,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma, const double* __restrict__ ws, double* __restrict__ wz,
                                                                 float* __restrict__ dx, float* __restrict__ dres,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t N, int C) {
    extern __shared__ float bn_sm[];            // [C] mean of g | [C] mean of g * xhat
    float* m1 = bn_sm;
    float* m2 = bn_sm + C;
    const int nrep = wgs_bn_nrep(C);
    const double invN = 1.0 / (double)N;
    for (int c = threadIdx.x; c < C; c += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int r = 0; r < nrep; ++r) { s1 += ws[(size_t)r * 2 * C + c]; s2 += ws[(size_t)r * 2 * C + C + c]; }
        // (the separate collapse launch rounds the sums to fp32 — they are dbeta / dgamma — and the apply kernel divides those: same here)
        const float f1 = (float)s1, f2 = (float)s2;
        m1[c] = (float)(f1 * invN); m2[c] = (float)(f2 * invN);
        if (blockIdx.x == 0) { dbeta[c] = f1; dgamma[c] = f2; }
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < bn_zero_extent(C); i += 256) wz[i] = 0.0;
    __syncthreads();
    const int c4n = C >> 2;
    const int64_t total = N * c4n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % c4n) * 4;
        float4 g = reinterpret_cast<const float4*>(dyA)[e];
        if (dyB) {
            const float4 g2 = reinterpret_cast<const float4*>(dyB)[e];
            g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
        }
        if (out) {
            const float4 o = reinterpret_cast<const float4*>(out)[e];
            g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
            g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
        }
        if (dres) reinterpret_cast<float4*>(dres)[e] = g;
        const float4 v = reinterpret_cast<const float4*>(x)[e];
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
        const float4 a1 = *reinterpret_cast<const float4*>(m1 + c), a2 = *reinterpret_cast<const float4*>(m2 + c);
        float4 d;
        d.x = ga.x * is.x * (g.x - a1.x - (v.x - mu.x) * is.x * a2.x);
        d.y = ga.y * is.y * (g.y - a1.y - (v.y - mu.y) * is.y * a2.y);
        d.z = ga.z * is.z * (g.z - a1.z - (v.z - mu.z) * is.z * a2.z);
        d.w = ga.w * is.w * (g.w - a1.w - (v.w - mu.w) * is.w * a2.w);
        reinterpret_cast<float4*>(dx)[e] = d;
    }
}

// g = (dyA + dyB)*(out > 0);  dx = gamma*invstd*(g - s1/N - xhat*s2/N)  (train)  or gamma*invstd*g (eval);
// dres (optional) = g;  block 0 also writes dgamma = s2, dbeta = s1.
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dyA,
                                                           const float* __restrict__ dyB, const float* __restrict__ out,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const double* __restrict__ ws,
                                                           float* __restrict__ dx, float* __restrict__ dres,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t N,
                                                           int C, int train) {
    // (dbeta = sum g and dgamma = sum g*xhat were written by ws_collapse_kernel, the launch in front of this one)
    const int c4n = C >> 2;
    const int64_t total = N * c4n;
    const double invN = 1.0 / (double)N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % c4n) * 4;
        float4 g = reinterpret_cast<const float4*>(dyA)[e];
        if (dyB) {
            const float4 g2 = reinterpret_cast<const float4*>(dyB)[e];
            g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
        }
        if (out) {
            const float4 o = reinterpret_cast<const float4*>(out)[e];
            g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
            g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
        }
        if (dres) reinterpret_cast<float4*>(dres)[e] = g;
        const float4 v = reinterpret_cast<const float4*>(x)[e];
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
        float4 d;
        if (train) {
            const float4 s1 = *reinterpret_cast<const float4*>(dbeta + c), s2 = *reinterpret_cast<const float4*>(dgamma + c);
            const float m1x = (float)(s1.x * invN), m1y = (float)(s1.y * invN), m1z = (float)(s1.z * invN), m1w = (float)(s1.w * invN);
            const float m2x = (float)(s2.x * invN), m2y = (float)(s2.y * invN), m2z = (float)(s2.z * invN), m2w = (float)(s2.w * invN);
            d.x = ga.x * is.x * (g.x - m1x - (v.x - mu.x) * is.x * m2x);
            d.y = ga.y * is.y * (g.y - m1y - (v.y - mu.y) * is.y * m2y);
            d.z = ga.z * is.z * (g.z - m1z - (v.z - mu.z) * is.z * m2z);
            d.w = ga.w * is.w * (g.w - m1w - (v.w - mu.w) * is.w * m2w);
        } else {
            d.x = ga.x * is.x * g.x; d.y = ga.y * is.y * g.y; d.z = ga.z * is.z * g.z; d.w = ga.w * is.w * g.w;
        }
        reinterpret_cast<float4*>(dx)[e] = d;
    }
}

// ---- max pooling k x k / stride s / pad p on NHWC; idx = ky*k+kx of the FIRST maximum (scan order),
// which is how torch's max_pool2d breaks ties (strict '>' update) — ties are common after ReLU.
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          unsigned char* __restrict__ idx, int B, int Hi, int Wi, int C,
                                                          int Ho, int Wo, int k, int s, int p) {
    const int c4n = C >> 2;
    const int64_t total = (int64_t)B * Ho * Wo * c4n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int64_t r = e;
        const int c = (int)(r % c4n) * 4; r /= c4n;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned char mi[4] = {0, 0, 0, 0};
        bool first = true;
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * s + ky - p;
            if (iy < 0 || iy >= Hi) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * s + kx - p;
                if (ix < 0 || ix >= Wi) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)b * Hi + iy) * Wi + ix) * C + c);
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (first || vv[q] > m[q]) { m[q] = vv[q]; mi[q] = (unsigned char)(ky * k + kx); }
                first = false;
            }
        }
        const size_t off = (((size_t)b * Ho + oy) * Wo + ox) * C + c;
        *reinterpret_cast<float4*>(y + off) = make_float4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<uchar4*>(idx + off) = make_uchar4(mi[0], mi[1], mi[2], mi[3]);
    }
}
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                          float* __restrict__ dx, int B, int Hi, int Wi, int C, int Ho,
                                                          int Wo, int k, int s, int p) {
    const int c4n = C >> 2;
    const int64_t total = (int64_t)B * Hi * Wi * c4n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int64_t r = e;
        const int c = (int)(r % c4n) * 4; r /= c4n;
        const int ix = (int)(r % Wi); r /= Wi;
        const int iy = (int)(r % Hi);
        const int b = (int)(r / Hi);
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        // windows (oy,ox) that contain (iy,ix): oy*s - p <= iy <= oy*s - p + k - 1
        for (int oy = (iy + p) / s; oy >= 0 && oy * s + k - 1 - p >= iy; --oy) {
            if (oy >= Ho) continue;
            const int ky = iy + p - oy * s;
            for (int ox = (ix + p) / s; ox >= 0 && ox * s + k - 1 - p >= ix; --ox) {
                if (ox >= Wo) continue;
                const int kx = ix + p - ox * s;
                const size_t off = (((size_t)b * Ho + oy) * Wo + ox) * C + c;
                const uchar4 id = *reinterpret_cast<const uchar4*>(idx + off);
                const float4 g = *reinterpret_cast<const float4*>(dy + off);
                const unsigned char me = (unsigned char)(ky * k + kx);
                if (id.x == me) a[0] += g.x;
                if (id.y == me) a[1] += g.y;
                if (id.z == me) a[2] += g.z;
                if (id.w == me) a[3] += g.w;
            }
        }
        *reinterpret_cast<float4*>(dx + (((size_t)b * Hi + iy) * Wi + ix) * C + c) = make_float4(a[0], a[1], a[2], a[3]);
    }
}

// ---- global average pooling [B,P,C] -> [B,C] and its backward ------------------------------------------
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int P, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (c >= C) return;
    float acc = 0.f;
    for (int p = 0; p < P; ++p) acc += x[((size_t)b * P + p) * C + c];
    y[(size_t)b * C + c] = acc / P;
}
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int P, int C) {
    const int64_t total = (int64_t)B * P * C;
    const float f = 1.f / P;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const int b = (int)(e / ((int64_t)P * C));
        dx[e] = dy[(size_t)b * C + c] * f;
    }
}

// ---- backward of a nearest-neighbour 2x upsample: dx[b,y,x,c] = sum of the 2x2 block of dy ------------------
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int H,
                                                             int W, int C) {
    const int c4n = C >> 2;
    const int64_t total = (int64_t)B * H * W * c4n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int64_t r = e;
        const int c = (int)(r % c4n) * 4; r /= c4n;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        const float* p0 = dy + (((size_t)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C + c;
        const float4 a = *reinterpret_cast<const float4*>(p0), bb = *reinterpret_cast<const float4*>(p0 + C);
        const float4 cc = *reinterpret_cast<const float4*>(p0 + (size_t)2 * W * C), d = *reinterpret_cast<const float4*>(p0 + (size_t)2 * W * C + C);
        *reinterpret_cast<float4*>(dx + e * 4) = make_float4(a.x + bb.x + cc.x + d.x, a.y + bb.y + cc.y + d.y,
                                                             a.z + bb.z + cc.z + d.z, a.w + bb.w + cc.w + d.w);
    }
}

// ---- loss: CrossEntropy(mean) + L1(mean), their gradients, argmax / accuracy ----------------------------
// one block (256 threads) per sample.
__global__ __launch_bounds__(256) void loss_rows_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                        const float* __restrict__ mag_pred, const float* __restrict__ mag_tgt,
                                                        float lambda_cls, float lambda_reg, float* __restrict__ dlogits,
                                                        float* __restrict__ dmag, float* __restrict__ row_ce,
                                                        float* __restrict__ row_l1, int64_t* __restrict__ row_argmax, int B, int K) {
    __shared__ float sval[4];
    __shared__ int sidx[4];
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* lr = logits + (size_t)b * K;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -INFINITY;
    int mi = 0x7fffffff;
    for (int j = threadIdx.x; j < K; j += 256) {
        const float v = lr[j];
        if (v > m || (v == m && j < mi)) { m = v; mi = j; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_xor(m, off, 64);
        const int oi = __shfl_xor(mi, off, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    if (lane == 0) { sval[wave] = m; sidx[wave] = mi; }
    __syncthreads();
    m = sval[0]; mi = sidx[0];
    for (int w = 1; w < 4; ++w)
        if (sval[w] > m || (sval[w] == m && sidx[w] < mi)) { m = sval[w]; mi = sidx[w]; }
    float se = 0.f;
    for (int j = threadIdx.x; j < K; j += 256) se += expf(lr[j] - m);
    se = block_sum<4>(se, red);
    const float lse = m + logf(se);
    const int t = (int)target[b];
    const float invB = 1.f / B;
    for (int j = threadIdx.x; j < K; j += 256) {
        const float pj = expf(lr[j] - lse);
        dlogits[(size_t)b * K + j] = (pj - (j == t ? 1.f : 0.f)) * lambda_cls * invB;
    }
    if (threadIdx.x == 0) {
        row_ce[b] = lse - lr[t];
        const float d = mag_pred[b] - mag_tgt[b];
        row_l1[b] = fabsf(d);
        dmag[b] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * lambda_reg * invB;
        row_argmax[b] = mi;
    }
}
// stats[0..3] = (classification_loss, regression_loss, total_loss, accuracy), lib/trainer.py:245-258
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* __restrict__ row_ce, const float* __restrict__ row_l1,
                                                          const int64_t* __restrict__ row_argmax, const int64_t* __restrict__ target,
                                                          float lambda_cls, float lambda_reg, float* __restrict__ stats, int B) {
    __shared__ float red[4];
    float a = 0.f, l = 0.f, c = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) { a += row_ce[b]; l += row_l1[b]; c += (row_argmax[b] == target[b]) ? 1.f : 0.f; }
    a = block_sum<4>(a, red); l = block_sum<4>(l, red); c = block_sum<4>(c, red);
    if (threadIdx.x == 0) {
        const float ce = a / B, l1 = l / B;
        stats[0] = ce; stats[1] = l1; stats[2] = lambda_cls * ce + lambda_reg * l1; stats[3] = c / B;
    }
}

// ---- Adam (torch.optim.Adam defaults: no amsgrad, no weight decay), one flat buffer ---------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float beta1, float beta2, float eps,
                                                   float step_size, float bc2_sqrt, float grad_scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * grad_scale;
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);          // exp_avg.lerp_(grad, 1-beta1)
        const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;      // exp_avg_sq.mul_(beta2).addcmul_(g,g,1-beta2)
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

// the fused BatchNorm apply kernels pay a prologue per workgroup (<= 32 KB of replica sums): fewer, fatter workgroups
int grid_fused(int64_t work) {
    int g = wgs_cdiv(work, 1024);
    return g > 2048 ? 2048 : (g < 1 ? 1 : g);
}

int grid_for(int64_t work) {
    int g = wgs_cdiv(work, 256);
    return g > 8192 ? 8192 : (g < 1 ? 1 : g);
}

}  // namespace

extern "C" {

int wgs_pack_pair_nhwc(const float* x1, const float* x2, float* y, int B, int c, int HW, int Cp, wgs_stream_t stream) {
    WGS_CHECK_ARG(x1 && x2 && y && B > 0 && c > 0 && HW > 0 && Cp >= 2 * c, "wgs_pack_pair_nhwc: bad arguments");
    WGS_LAUNCH(pack_pair_kernel, dim3(grid_for((int64_t)B * HW)), dim3(256), 0, (hipStream_t)stream, x1, x2, y, B, c, HW, Cp);
    WGS_CHECK_LAUNCH("pack_pair_kernel");
    return WGS_OK;
}
int wgs_pack_pair_s2d(const float* x1, const float* x2, float* y, int B, int c, int H, int W, wgs_stream_t stream) {
    WGS_CHECK_ARG(x1 && x2 && y && B > 0 && c > 0 && 2 * c <= 8 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "wgs_pack_pair_s2d: bad arguments (2c <= 8, even H / W)");
    WGS_LAUNCH(pack_pair_s2d_kernel, dim3(grid_for((int64_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, x1, x2, y, B, c, H, W);
    WGS_CHECK_LAUNCH("pack_pair_s2d_kernel");
    return WGS_OK;
}
int wgs_unpack_pair_s2d_grad(const float* dys, float* d1, float* d2, int B, int c, int H, int W, wgs_stream_t stream) {
    WGS_CHECK_ARG(dys && B > 0 && c > 0 && 2 * c <= 8 && H % 2 == 0 && W % 2 == 0, "wgs_unpack_pair_s2d_grad: bad arguments");
    WGS_LAUNCH(unpack_pair_s2d_kernel, dim3(grid_for((int64_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, dys, d1, d2, B, c, H, W);
    WGS_CHECK_LAUNCH("unpack_pair_s2d_kernel");
    return WGS_OK;
}
int wgs_stem_weight_s2d(const float* src, float* dst, int Co, int Ci, int back, wgs_stream_t stream) {
    WGS_CHECK_ARG(src && dst && Co > 0 && Ci > 0 && Ci <= 8, "wgs_stem_weight_s2d: bad arguments (Ci <= 8)");
    const int n = back ? Co * 49 * Ci : Co * 16 * 32;
    WGS_LAUNCH(stem_weight_s2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, Co, Ci, back);
    WGS_CHECK_LAUNCH("stem_weight_s2d_kernel");
    return WGS_OK;
}

int wgs_unpack_pair_grad(const float* dy, float* d1, float* d2, int B, int c, int HW, int Cp, wgs_stream_t stream) {
    WGS_CHECK_ARG(dy && (d1 || d2) && B > 0 && c > 0 && HW > 0 && Cp >= 2 * c, "wgs_unpack_pair_grad: bad arguments");
    WGS_LAUNCH(unpack_pair_kernel, dim3(grid_for((int64_t)B * HW)), dim3(256), 0, (hipStream_t)stream, dy, d1, d2, B, c, HW, Cp);
    WGS_CHECK_LAUNCH("unpack_pair_kernel");
    return WGS_OK;
}

// WGS_CHECK_WS=1 (debug): the zero-on-entry contract of the BatchNorm / column-sum scratch, checked synchronously before the reduction
// (a dirty buffer — a client following the old per-call-memset contract, an interrupted call, two streams sharing one ws — otherwise
// gives wrong statistics silently; include/wgs.h)
static int ws_is_zero(const double* ws, int C, hipStream_t st, const char* who) {
    if (!wgs_flags().check_ws) return WGS_OK;
    const size_t n = (size_t)WGS_BN_WS_DOUBLES(C);
    double* h = (double*)malloc(n * sizeof(double));
    if (!h) return WGS_OK;
    hipError_t e = hipMemcpyAsync(h, ws, n * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    size_t bad = n;
    if (e == hipSuccess)
        for (size_t i = 0; i < n; ++i)
            if (h[i] != 0.0) { bad = i; break; }
    free(h);
    if (e != hipSuccess) { wgs_set_error("%s: WGS_CHECK_WS copy failed: %s", who, hipGetErrorString(e)); return WGS_ELAUNCH; }
    if (bad != n) { wgs_set_error("%s: ws is not zero on entry (double %zu of %zu): zero it once when allocating it, one stream per buffer", who, bad, n); return WGS_EINVAL; }
    return WGS_OK;
}

static int reduce_rows_per_block(int64_t N, int C) {
    const int c4n = C >> 2;
    const int tpr = c4n < 256 ? c4n : 256;
    const int rpi = 256 / tpr;
    int64_t rpb = (N + 767) / 768;     // ~3 workgroups per CU: enough loads in flight, few same-address fp64 atomics
    if (rpb < rpi * 4) rpb = rpi * 4;
    return (int)rpb;
}

int wgs_bn_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y, float* save_mean,
               float* save_invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, double* ws,
               int64_t N, int C, float eps, float momentum, int relu, int train, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && gamma && beta && y && save_mean && save_invstd && ws, "wgs_bn_fwd: null pointer");
    WGS_CHECK_ARG(N > 0 && C >= 4 && C % 4 == 0, "wgs_bn_fwd: C=%d must be a multiple of 4", C);
    WGS_CHECK_ARG(train || (running_mean && running_var), "wgs_bn_fwd: eval mode needs running stats");
    hipStream_t st = (hipStream_t)stream;
    if (train) {
        if (int rc = ws_is_zero(ws, C, st, "wgs_bn_fwd")) return rc;
        const int rpb = reduce_rows_per_block(N, C);
        WGS_LAUNCH(chan_reduce_kernel<0>, dim3(wgs_cdiv(N, rpb)), dim3(256), 0, st, x, nullptr, nullptr, nullptr,
                           nullptr, nullptr, ws, N, C, rpb);
        WGS_LAUNCH(bn_finalize_kernel, dim3(wgs_cdiv(C, 256)), dim3(256), 0, st, ws, save_mean, save_invstd,
                           running_mean, running_var, num_batches_tracked, N, C, eps, momentum);
    } else {
        WGS_LAUNCH(bn_eval_stats_kernel, dim3(wgs_cdiv(C, 256)), dim3(256), 0, st, running_mean, running_var,
                           save_mean, save_invstd, C, eps);
    }
    WGS_LAUNCH(bn_apply_kernel, dim3(grid_for(N * (C / 4))), dim3(256), 0, st, x, save_mean, save_invstd, gamma, beta,
                       residual, y, N, C, relu);
    WGS_CHECK_LAUNCH("bn_fwd");
    return WGS_OK;
}

int wgs_bn_fwd_sums(const float* x, const float* gamma, const float* beta, const float* residual, float* y, float* save_mean,
                    float* save_invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, double* ws,
                    int64_t N, int C, float eps, float momentum, int relu, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && gamma && beta && y && save_mean && save_invstd && ws, "wgs_bn_fwd_sums: null pointer");
    WGS_CHECK_ARG(N > 0 && C >= 4 && C % 4 == 0, "wgs_bn_fwd_sums: C=%d must be a multiple of 4", C);
    hipStream_t st = (hipStream_t)stream;
    // (ws holds the sums the producing conv's epilogue accumulated — wgs_conv_desc.col_stats — in chan_reduce's replica layout)
    WGS_LAUNCH(bn_finalize_kernel, dim3(wgs_cdiv(C, 256)), dim3(256), 0, st, ws, save_mean, save_invstd,
                       running_mean, running_var, num_batches_tracked, N, C, eps, momentum);
    WGS_LAUNCH(bn_apply_kernel, dim3(grid_for(N * (C / 4))), dim3(256), 0, st, x, save_mean, save_invstd, gamma, beta,
                       residual, y, N, C, relu);
    WGS_CHECK_LAUNCH("bn_fwd_sums");
    return WGS_OK;
}

int wgs_bn_fwd_fused(const float* x, const float* gamma, const float* beta, const float* residual, float* y, float* save_mean,
                     float* save_invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, const double* ws_sums,
                     double* ws_zero, int64_t N, int C, float eps, float momentum, int relu, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && gamma && beta && y && save_mean && save_invstd && ws_sums && ws_zero && ws_sums != ws_zero, "wgs_bn_fwd_fused: null pointer (or one scratch passed twice)");
    WGS_CHECK_ARG(N > 0 && C >= 4 && C % 4 == 0 && C <= 8192, "wgs_bn_fwd_fused: C=%d must be a multiple of 4 (<= 8192)", C);
    WGS_LAUNCH(bn_apply_fused_kernel, dim3(grid_fused(N * (C / 4))), dim3(256), (size_t)2 * C * sizeof(float), (hipStream_t)stream, x, ws_sums, ws_zero,
               gamma, beta, residual, y, save_mean, save_invstd, running_mean, running_var, num_batches_tracked, N, C, eps, momentum, relu);
    WGS_CHECK_LAUNCH("bn_apply_fused_kernel");
    return WGS_OK;
}

int wgs_bn_bwd_fused(const float* x, const float* dyA, const float* dyB, const float* out, const float* save_mean,
                     const float* save_invstd, const float* gamma, float* dx, float* dres, float* dgamma, float* dbeta,
                     double* ws, double* ws_zero, int64_t N, int C, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && dyA && save_mean && save_invstd && gamma && dx && dgamma && dbeta && ws && ws_zero && ws != ws_zero,
                  "wgs_bn_bwd_fused: null pointer (or one scratch passed twice)");
    WGS_CHECK_ARG(N > 0 && C >= 4 && C % 4 == 0 && C <= 8192, "wgs_bn_bwd_fused: C=%d must be a multiple of 4 (<= 8192)", C);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = ws_is_zero(ws, C, st, "wgs_bn_bwd_fused")) return rc;
    const int rpb = reduce_rows_per_block(N, C);
    WGS_LAUNCH(chan_reduce_kernel<1>, dim3(wgs_cdiv(N, rpb)), dim3(256), 0, st, x, dyA, dyB, out, save_mean, save_invstd, ws, N, C, rpb);
    WGS_LAUNCH(bn_bwd_apply_fused_kernel, dim3(grid_fused(N * (C / 4))), dim3(256), (size_t)2 * C * sizeof(float), st, x, dyA, dyB, out, save_mean,
               save_invstd, gamma, ws, ws_zero, dx, dres, dgamma, dbeta, N, C);
    WGS_CHECK_LAUNCH("bn_bwd_fused");
    return WGS_OK;
}

int wgs_bn_bwd(const float* x, const float* dyA, const float* dyB, const float* out, const float* save_mean,
               const float* save_invstd, const float* gamma, float* dx, float* dres, float* dgamma, float* dbeta,
               double* ws, int64_t N, int C, int train, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && dyA && save_mean && save_invstd && gamma && dx && ws, "wgs_bn_bwd: null pointer");
    WGS_CHECK_ARG(N > 0 && C >= 4 && C % 4 == 0, "wgs_bn_bwd: C=%d must be a multiple of 4", C);
    WGS_CHECK_ARG((dgamma && dbeta) || (!train && !dgamma && !dbeta),
                  "wgs_bn_bwd: dgamma and dbeta are required in train mode (the input gradient reads the sums back from them); eval mode takes both or neither");
    hipStream_t st = (hipStream_t)stream;
    const int rpb = reduce_rows_per_block(N, C);
    if (dgamma) {        // (eval mode without parameter gradients — a frozen generator's BatchNorm — needs no reduction at all)
        if (int rc = ws_is_zero(ws, C, st, "wgs_bn_bwd")) return rc;
        WGS_LAUNCH(chan_reduce_kernel<1>, dim3(wgs_cdiv(N, rpb)), dim3(256), 0, st, x, dyA, dyB, out, save_mean,
                           save_invstd, ws, N, C, rpb);
        WGS_LAUNCH(ws_collapse_kernel, dim3(wgs_cdiv(2 * C, 256)), dim3(256), 0, st, ws, C, dbeta, dgamma);
    }
    WGS_LAUNCH(bn_bwd_apply_kernel, dim3(grid_for(N * (C / 4))), dim3(256), 0, st, x, dyA, dyB, out, save_mean,
                       save_invstd, gamma, ws, dx, dres, dgamma, dbeta, N, C, train);
    WGS_CHECK_LAUNCH("bn_bwd");
    return WGS_OK;
}

int wgs_maxpool_fwd(const float* x, float* y, unsigned char* idx, int B, int Hi, int Wi, int C, int k, int s, int p,
                    wgs_stream_t stream) {
    WGS_CHECK_ARG(x && y && idx && B > 0 && Hi > 0 && Wi > 0 && C >= 4 && C % 4 == 0 && k > 0 && k <= 15 && s > 0 && p >= 0,
                  "wgs_maxpool_fwd: bad arguments");
    const int Ho = (Hi + 2 * p - k) / s + 1, Wo = (Wi + 2 * p - k) / s + 1;
    WGS_LAUNCH(maxpool_fwd_kernel, dim3(grid_for((int64_t)B * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y,
                       idx, B, Hi, Wi, C, Ho, Wo, k, s, p);
    WGS_CHECK_LAUNCH("maxpool_fwd_kernel");
    return WGS_OK;
}
int wgs_maxpool_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int Hi, int Wi, int C, int k, int s, int p,
                    wgs_stream_t stream) {
    WGS_CHECK_ARG(dy && idx && dx && B > 0 && Hi > 0 && Wi > 0 && C >= 4 && C % 4 == 0 && k > 0 && s > 0 && p >= 0,
                  "wgs_maxpool_bwd: bad arguments");
    const int Ho = (Hi + 2 * p - k) / s + 1, Wo = (Wi + 2 * p - k) / s + 1;
    WGS_LAUNCH(maxpool_bwd_kernel, dim3(grid_for((int64_t)B * Hi * Wi * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy,
                       idx, dx, B, Hi, Wi, C, Ho, Wo, k, s, p);
    WGS_CHECK_LAUNCH("maxpool_bwd_kernel");
    return WGS_OK;
}

int wgs_avgpool_fwd(const float* x, float* y, int B, int P, int C, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && y && B > 0 && P > 0 && C > 0, "wgs_avgpool_fwd: bad arguments");
    WGS_LAUNCH(avgpool_fwd_kernel, dim3(wgs_cdiv(C, 256), B), dim3(256), 0, (hipStream_t)stream, x, y, P, C);
    WGS_CHECK_LAUNCH("avgpool_fwd_kernel");
    return WGS_OK;
}
int wgs_avgpool_bwd(const float* dy, float* dx, int B, int P, int C, wgs_stream_t stream) {
    WGS_CHECK_ARG(dy && dx && B > 0 && P > 0 && C > 0, "wgs_avgpool_bwd: bad arguments");
    WGS_LAUNCH(avgpool_bwd_kernel, dim3(grid_for((int64_t)B * P * C)), dim3(256), 0, (hipStream_t)stream, dy, dx, B, P, C);
    WGS_CHECK_LAUNCH("avgpool_bwd_kernel");
    return WGS_OK;
}

int wgs_upsample2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, wgs_stream_t stream) {
    WGS_CHECK_ARG(dy && dx && B > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0, "wgs_upsample2x_bwd: bad arguments (C %% 4)");
    WGS_LAUNCH(upsample2x_bwd_kernel, dim3(grid_for((int64_t)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, dx, B, H, W, C);
    WGS_CHECK_LAUNCH("upsample2x_bwd_kernel");
    return WGS_OK;
}

int wgs_colsum(const float* x, float* out, double* ws, int64_t N, int C, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && out && ws && N > 0 && C >= 4 && C % 4 == 0, "wgs_colsum: bad arguments (C %% 4)");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = ws_is_zero(ws, C, st, "wgs_colsum")) return rc;
    const int rpb = reduce_rows_per_block(N, C);
    WGS_LAUNCH(chan_reduce_kernel<2>, dim3(wgs_cdiv(N, rpb)), dim3(256), 0, st, x, nullptr, nullptr, nullptr, nullptr,
                       nullptr, ws, N, C, rpb);
    WGS_LAUNCH(ws_collapse_kernel, dim3(wgs_cdiv(2 * C, 256)), dim3(256), 0, st, ws, C, out, (float*)nullptr);
    WGS_CHECK_LAUNCH("colsum");
    return WGS_OK;
}

int wgs_ce_l1_loss(const float* logits, const int64_t* target, const float* mag_pred, const float* mag_target, float lambda_cls,
                   float lambda_reg, float* dlogits, float* dmag, float* stats, int64_t* argmax, float* ws, int B, int K,
                   wgs_stream_t stream) {
    WGS_CHECK_ARG(logits && target && mag_pred && mag_target && dlogits && dmag && stats && argmax && ws,
                  "wgs_ce_l1_loss: null pointer");
    WGS_CHECK_ARG(B > 0 && K > 0, "wgs_ce_l1_loss: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    WGS_LAUNCH(loss_rows_kernel, dim3(B), dim3(256), 0, st, logits, target, mag_pred, mag_target, lambda_cls, lambda_reg,
                       dlogits, dmag, ws, ws + B, argmax, B, K);
    WGS_LAUNCH(loss_finish_kernel, dim3(1), dim3(256), 0, st, ws, ws + B, argmax, target, lambda_cls, lambda_reg, stats, B);
    WGS_CHECK_LAUNCH("ce_l1_loss");
    return WGS_OK;
}

int wgs_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, int step, float grad_scale, wgs_stream_t stream) {
    WGS_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "wgs_adam_step: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    WGS_LAUNCH(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                       beta1, beta2, eps, step_size, bc2_sqrt, grad_scale);
    WGS_CHECK_LAUNCH("adam_kernel");
    return WGS_OK;
}

}  // extern "C"
