// StyleGAN2 generator glue kernels (everything in models/StyleGAN2/model.py that is not a dense
// 3x3 contraction): PixelNorm :9-15, EqualLinear :110-136 (mapping MLP + per-layer modulation),
// demodulation :194-195, Blur + NoiseInjection + FusedLeakyReLU fused :67-81,231-241,264, ToRGB
// :270-282, and the hand-derived backward of the shifted branch (dgrad only; G is frozen).
// All HBM-/latency-bound: float4 accesses over the channel (NHWC minor) axis, wave64 reductions.
#include "wgs_common.h"
#include "fir4.h"
#include "../../include/wgs.h"

namespace {

constexpr float SQRT2 = 1.41421356237309504880f;

// ---- PixelNorm over rows of length d -------------------------------------------------------------
// Two forms.  Long rows (the 512-d latent codes): one wave per row.  Short rows (ProgGAN feature maps: 16 .. 256 channels per
// pixel, up to 33 M pixels per tensor): LPR = d / 4 lanes per row, each lane one float4, 64 / LPR rows per wave, reductions by
// xor-shuffles inside the lane group — every lane loads and stores 16 bytes, so the pass streams at HBM rate (one wave per
// 16-float row left 48 lanes idle and ran at ~1 TB/s: 51 ms of ProgGAN-1024's 200 ms step).
__global__ __launch_bounds__(256) void pixelnorm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int rows, int d, float eps) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const float* xr = x + (size_t)r * d;
    float s = 0.f;
    for (int j = lane; j < d; j += 64) s = fmaf(xr[j], xr[j], s);
    s = wave_sum(s);
    const float f = rsqrtf(s / d + eps);
    for (int j = lane; j < d; j += 64) y[(size_t)r * d + j] = xr[j] * f;
}
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <int LPR>
__global__ __launch_bounds__(256) void pixelnorm_fwd_vec_kernel(const float* __restrict__ x, float* __restrict__ y, long rows, float eps) {
    constexpr int RPB = 256 / LPR;                     // rows per workgroup pass
    const int sub = threadIdx.x % LPR, rl = threadIdx.x / LPR;
    const float inv_d = 1.f / (4 * LPR);
    for (long r = (long)blockIdx.x * RPB + rl; r < rows; r += (long)gridDim.x * RPB) {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * (4 * LPR) + 4 * sub);
        // the same fma order per lane as the scalar form would need a different tree; the sum is formed in fp32 either way
        const float s = group_sum<LPR>(fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, v.w * v.w))));
        const float f = rsqrtf(s * inv_d + eps);
        *reinterpret_cast<float4*>(y + (size_t)r * (4 * LPR) + 4 * sub) = make_float4(v.x * f, v.y * f, v.z * f, v.w * f);
    }
}
// y = x * f, f = (mean(x^2)+eps)^-1/2  =>  gx = f*gy - x * f^3 * (x.gy)/d
// act_slope != 1: the result is also multiplied by the leaky-relu gate of x (x > 0 ? 1 : act_slope) — x is the activated output of
// the previous block, so this is that block's activation backward folded into the store (ProgGAN, models/ProgGAN/model.py:35-62)
// gx_amax (optional): device scalar raised (atomic max) to max |gx| of the launch — the magnitude bound of the fp16 operand scale of the
// gradient conv that consumes gx (wgs_conv_desc.a_amax)
__global__ __launch_bounds__(256) void pixelnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                            float* __restrict__ gx, int rows, int d, float eps, float act_slope,
                                                            float* __restrict__ gx_amax) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const float* xr = x + (size_t)r * d;
    const float* gr = gy + (size_t)r * d;
    float s = 0.f, t = 0.f;
    for (int j = lane; j < d; j += 64) { s = fmaf(xr[j], xr[j], s); t = fmaf(xr[j], gr[j], t); }
    s = wave_sum(s); t = wave_sum(t);
    const float f = rsqrtf(s / d + eps);
    const float c = f * f * f * t / d;
    float am = 0.f;
    for (int j = lane; j < d; j += 64) {
        const float v = (f * gr[j] - xr[j] * c) * (xr[j] > 0.f ? 1.f : act_slope);
        gx[(size_t)r * d + j] = v;
        am = fmaxf(am, fabsf(v));
    }
    if (gx_amax) {
        am = wave_max(am);
        if (lane == 0) raise_amax(gx_amax, am);
    }
}
template <int LPR>
__global__ __launch_bounds__(256) void pixelnorm_bwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                float* __restrict__ gx, long rows, float eps, float act_slope,
                                                                float* __restrict__ gx_amax) {
    constexpr int RPB = 256 / LPR;
    const int sub = threadIdx.x % LPR, rl = threadIdx.x / LPR;
    const float inv_d = 1.f / (4 * LPR);
    float am = 0.f;
    for (long r = (long)blockIdx.x * RPB + rl; r < rows; r += (long)gridDim.x * RPB) {
        const size_t o = (size_t)r * (4 * LPR) + 4 * sub;
        const float4 v = *reinterpret_cast<const float4*>(x + o);
        const float4 g = *reinterpret_cast<const float4*>(gy + o);
        const float s = group_sum<LPR>(fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, v.w * v.w))));
        const float t = group_sum<LPR>(fmaf(v.x, g.x, fmaf(v.y, g.y, fmaf(v.z, g.z, v.w * g.w))));
        const float f = rsqrtf(s * inv_d + eps);
        const float c = f * f * f * t * inv_d;
        const float4 o4 = make_float4((f * g.x - v.x * c) * (v.x > 0.f ? 1.f : act_slope), (f * g.y - v.y * c) * (v.y > 0.f ? 1.f : act_slope),
                                      (f * g.z - v.z * c) * (v.z > 0.f ? 1.f : act_slope), (f * g.w - v.w * c) * (v.w > 0.f ? 1.f : act_slope));
        *reinterpret_cast<float4*>(gx + o) = o4;
        am = fmaxf(fmaxf(am, fmaxf(fabsf(o4.x), fabsf(o4.y))), fmaxf(fabsf(o4.z), fabsf(o4.w)));
    }
    if (gx_amax) {      // (every lane of the wave reaches this point: the row loop has no early exit)
        am = wave_max(am);
        if ((threadIdx.x & 63) == 0) raise_amax(gx_amax, am);
    }
}

// ---- small dense layers (M = batch rows <= a few hundred) ----------------------------------------
// y[m,n] = epi( wscale * sum_k f(x[m,k]) w[n,k] + bscale*bias[n] ), one wave per output column n.
//   f = identity or square (in_square);  epi: 0 none, 1 leaky-relu(0.2)*sqrt(2), 2 rsqrt(v + eps)
template <int NT>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         int M, int N, int K, int ldx, int ldy, float wscale,
                                                         float bscale, int in_square, int epi, float eps, float out_gain) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    float4 wr[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int k = 4 * lane + 256 * t;
        wr[t] = (k < K) ? *reinterpret_cast<const float4*>(w + (size_t)n * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float b = bias ? bias[n] * bscale : 0.f;
    // eight batch rows per trip: their loads and shuffle reductions are independent, so the latencies overlap
    constexpr int U = 8;
    for (int m0 = 0; m0 < M; m0 += U) {
        float acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m0 + u < M ? m0 + u : M - 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int k = 4 * lane + 256 * t;
                if (k < K) {
                    float4 xv = *reinterpret_cast<const float4*>(x + (size_t)m * ldx + k);
                    if (in_square) { xv.x *= xv.x; xv.y *= xv.y; xv.z *= xv.z; xv.w *= xv.w; }
                    acc[u] = fmaf(xv.x, wr[t].x, acc[u]); acc[u] = fmaf(xv.y, wr[t].y, acc[u]);
                    acc[u] = fmaf(xv.z, wr[t].z, acc[u]); acc[u] = fmaf(xv.w, wr[t].w, acc[u]);
                }
            }
        }
        const float tot = wave_sum8(acc, lane);          // lane 8*j: row m0 + j
        if ((lane & 7) == 0 && m0 + (lane >> 3) < M) {
            float v = fmaf(tot, wscale, b);
            if (epi == 1) v = (v > 0.f ? v : 0.2f * v) * SQRT2;
            else if (epi == 2) v = rsqrtf(v + eps);
            y[(size_t)(m0 + (lane >> 3)) * ldy + n] = v * out_gain;
        }
    }
}

// ---- the whole mapping network in ONE launch: PixelNorm + L x (EqualLinear + fused leaky-relu) (model.py:288-295) --------
// One workgroup (8 waves) carries MLP_RPB = 2 batch rows through all L layers: the rows' activations live in LDS (ping-pong),
// a wave owns d/8 output columns per layer and streams their weight rows (coalesced 2 KB per row) against the two activation
// rows it keeps in registers.  Every layer's output also goes to acts[l+1] (the backward's gates).  Arithmetic = the per-layer
// kernels' (same k-to-lane assignment, same reduction), so results are bit-identical to L launches of linear_fwd_kernel<2>.
// A workgroup streams all L*d*d weights: ~8 MB at the ~150 GB/s one CU sustains from L2 = ~55 us, against L+1 launches of
// ~14 us each; splitting the columns of a layer over workgroups instead would need a grid barrier per layer (4-7 us each).
struct MappingArgs {
    const float* z; float* acts;            // acts: [(L+1)][B][d]
    const float* w[16]; const float* b[16];
    int B, d, L;
    float wscale, lr_mul, eps;
};
constexpr int MLP_RPB = 2, MLP_D = 512;
__global__ __launch_bounds__(512) void mapping_mlp_fwd_kernel(const MappingArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[2][MLP_RPB][MLP_D];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = blockIdx.x * MLP_RPB;
    const size_t plane = (size_t)a.B * MLP_D;
    // PixelNorm of this workgroup's rows (wave r < MLP_RPB), the arithmetic of pixelnorm_fwd_kernel
    if (wave < MLP_RPB) {
        const int r = row0 + wave;
        if (r < a.B) {
            const float* xr = a.z + (size_t)r * MLP_D;
            float s = 0.f;
            for (int j = lane; j < MLP_D; j += 64) s = fmaf(xr[j], xr[j], s);
            s = wave_sum(s);
            const float f = rsqrtf(s / MLP_D + a.eps);
            for (int j = lane; j < MLP_D; j += 64) {
                const float v = xr[j] * f;
                xs[0][wave][j] = v;
                a.acts[(size_t)r * MLP_D + j] = v;
            }
        } else {
            for (int j = lane; j < MLP_D; j += 64) xs[0][wave][j] = 0.f;
        }
    }
    __syncthreads();
    constexpr int NPW = MLP_D / 8;                  // output columns per wave and layer
    for (int l = 0; l < a.L; ++l) {
        const float* __restrict__ w = a.w[l];
        const float* __restrict__ bias = a.b[l];
        const float (*xin)[MLP_D] = xs[l & 1];
        float (*xout)[MLP_D] = xs[(l + 1) & 1];
        float4 xr[MLP_RPB][2];
#pragma unroll
        for (int r = 0; r < MLP_RPB; ++r)
#pragma unroll
            for (int t = 0; t < 2; ++t) xr[r][t] = *reinterpret_cast<const float4*>(&xin[r][4 * lane + 256 * t]);
        float* out = a.acts + (size_t)(l + 1) * plane;
        // weight rows in flight per wave: two groups of G (the next group's 2*G loads are issued before the current group is
        // multiplied and reduced; one row at a time is a memory round trip per output column)
        constexpr int G = 8;
        float4 wq[2][G][2];
        auto load_group = [&](int buf, int i0) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int n = wave * NPW + i0 + g;
                wq[buf][g][0] = *reinterpret_cast<const float4*>(w + (size_t)n * MLP_D + 4 * lane);
                wq[buf][g][1] = *reinterpret_cast<const float4*>(w + (size_t)n * MLP_D + 4 * lane + 256);
            }
        };
        load_group(0, 0);
#pragma unroll
        for (int gi = 0; gi < NPW / G; ++gi) {
            if (gi + 1 < NPW / G) load_group((gi + 1) & 1, (gi + 1) * G);
            float part[MLP_RPB][G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float4 w0 = wq[gi & 1][g][0], w1 = wq[gi & 1][g][1];
#pragma unroll
                for (int r = 0; r < MLP_RPB; ++r) {
                    float s = 0.f;
                    s = fmaf(xr[r][0].x, w0.x, s); s = fmaf(xr[r][0].y, w0.y, s); s = fmaf(xr[r][0].z, w0.z, s); s = fmaf(xr[r][0].w, w0.w, s);
                    s = fmaf(xr[r][1].x, w1.x, s); s = fmaf(xr[r][1].y, w1.y, s); s = fmaf(xr[r][1].z, w1.z, s); s = fmaf(xr[r][1].w, w1.w, s);
                    part[r][g] = s;
                }
            }
#pragma unroll
            for (int r = 0; r < MLP_RPB; ++r) {
                const float tot = wave_sum8(part[r], lane);          // lane 8*g: column gi*G + g of row r
                if ((lane & 7) == 0) {
                    const int n = wave * NPW + gi * G + (lane >> 3);
                    float v = fmaf(tot, a.wscale, bias[n] * a.lr_mul);
                    v = (v > 0.f ? v : 0.2f * v) * SQRT2;
                    xout[r][n] = v;
                    if (row0 + r < a.B) out[(size_t)(row0 + r) * MLP_D + n] = v;
                }
            }
        }
        __syncthreads();
    }
}

// Backward of the same stack in ONE launch: gx_l = wscale * (g_l o gate(act_{l+1})) W_l for l = L-1 .. 0 (gate: the fused leaky-relu's
// slope * sqrt 2), i.e. L x wgs_linear_dgrad.  The per-layer launches are M = 32 rows of a 512 x 512 contraction: 44 us each, nine of
// them at the very end of the generator's backward (the step's critical path).  Here a workgroup carries two batch rows through all
// layers: thread k owns input column k, streams W[:, k] (consecutive k: coalesced 2-KB wave rows) against the gated gradient rows in LDS.
struct MappingBwdArgs {
    const float* gw; const float* acts; float* gx;
    const float* w[16];
    int B, L;
    float wscale;
};
__global__ __launch_bounds__(MLP_D) void mapping_mlp_bwd_kernel(const MappingBwdArgs a) {
    // thread t: in the contraction, column group cg = t % 128 (columns 4 cg .. 4 cg + 3, one 16-byte load per weight row) and row
    // residue ng = t / 128 (weight rows n = ng, ng + 4, ...): four times the bytes in flight of one dword per thread and row (the first
    // version: 30 us per layer); outside it, thread t owns column t (gate, the cross-residue sum)
    __shared__ float gs[MLP_RPB][MLP_D];
    __shared__ __attribute__((aligned(16))) float part[4][MLP_RPB][MLP_D];
    const int k = threadIdx.x, cg = k & 127, ng = k >> 7;
    const int row0 = blockIdx.x * MLP_RPB;
    const size_t plane = (size_t)a.B * MLP_D;
    float g[MLP_RPB];
#pragma unroll
    for (int r = 0; r < MLP_RPB; ++r) g[r] = row0 + r < a.B ? a.gw[(size_t)(row0 + r) * MLP_D + k] : 0.f;
    for (int l = a.L - 1; l >= 0; --l) {
        const float* __restrict__ w = a.w[l];
        const float* act = a.acts + (size_t)(l + 1) * plane;
#pragma unroll
        for (int r = 0; r < MLP_RPB; ++r) {
            const float o = row0 + r < a.B ? act[(size_t)(row0 + r) * MLP_D + k] : 0.f;
            gs[r][k] = g[r] * (o > 0.f ? SQRT2 : 0.2f * SQRT2);
        }
        __syncthreads();
        float4 acc[MLP_RPB];
#pragma unroll
        for (int r = 0; r < MLP_RPB; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 16
        for (int i = 0; i < MLP_D / 4; ++i) {
            const int n = 4 * i + ng;
            const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)n * MLP_D + 4 * cg);
#pragma unroll
            for (int r = 0; r < MLP_RPB; ++r) {
                const float gv = gs[r][n];
                acc[r].x = fmaf(gv, wv.x, acc[r].x); acc[r].y = fmaf(gv, wv.y, acc[r].y);
                acc[r].z = fmaf(gv, wv.z, acc[r].z); acc[r].w = fmaf(gv, wv.w, acc[r].w);
            }
        }
#pragma unroll
        for (int r = 0; r < MLP_RPB; ++r) *reinterpret_cast<float4*>(&part[ng][r][4 * cg]) = acc[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MLP_RPB; ++r) g[r] = ((part[0][r][k] + part[1][r][k]) + (part[2][r][k] + part[3][r][k])) * a.wscale;
        // (the next layer's gs / part writes come after its own barrier: gs is rewritten only after every thread has passed the barrier
        // above, and part only after the next contraction, behind the next barrier)
    }
#pragma unroll
    for (int r = 0; r < MLP_RPB; ++r)
        if (row0 + r < a.B) a.gx[(size_t)(row0 + r) * MLP_D + k] = g[r];
}

// The same for up to 16 independent small layers in ONE launch (blockIdx.y = layer): the per-layer demodulation
// vectors of the synthesis network are 13 such GEMVs per pass, each too small to fill the chip or hide its latency.
struct LinearBatchArgs {
    int n, M, in_square, epi;
    const float* x[16];
    const float* w[16];
    float* y[16];
    int N[16], K[16], ldx[16], ldy[16];
    float wscale[16], eps[16], out_gain[16];
};
__global__ __launch_bounds__(256) void linear_fwd_batch_kernel(const LinearBatchArgs a) {
    const int l = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    const int N = a.N[l], K = a.K[l];
    if (n >= N) return;
    const float* x = a.x[l];
    const float* w = a.w[l];
    const int ldx = a.ldx[l];
    float4 wr[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int k = 4 * lane + 256 * t;
        wr[t] = (k < K) ? *reinterpret_cast<const float4*>(w + (size_t)n * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int m0 = 0; m0 < a.M; m0 += 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + u < a.M ? m0 + u : a.M - 1;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int k = 4 * lane + 256 * t;
                if (k < K) {
                    float4 xv = *reinterpret_cast<const float4*>(x + (size_t)m * ldx + k);
                    if (a.in_square) { xv.x *= xv.x; xv.y *= xv.y; xv.z *= xv.z; xv.w *= xv.w; }
                    acc[u] = fmaf(xv.x, wr[t].x, acc[u]); acc[u] = fmaf(xv.y, wr[t].y, acc[u]);
                    acc[u] = fmaf(xv.z, wr[t].z, acc[u]); acc[u] = fmaf(xv.w, wr[t].w, acc[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = wave_sum(acc[u]);
        if (lane < 4 && m0 + lane < a.M) {
            float v = (lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3]) * a.wscale[l];
            if (a.epi == 1) v = (v > 0.f ? v : 0.2f * v) * SQRT2;
            else if (a.epi == 2) v = rsqrtf(v + a.eps[l]);
            a.y[l][(size_t)(m0 + lane) * a.ldy[l] + n] = v * a.out_gain[l];
        }
    }
}

// gx[m,k] (+)= wscale * sum_n g'[m,n] w[n,k],  g' = gy * (gate ? (gate[m,n] > 0 ? gain : gain*slope) : 1)
// Block = 16 k-columns x 16 n-groups; every thread accumulates its n-stripe in fp64 and the 16 stripes are
// combined in fp64 through LDS (these reductions run over up to ~6000 terms — all modulation layers — and
// feed the ill-conditioned mapping-network Jacobian: partial sums must not be rounded to fp32).
__global__ __launch_bounds__(256) void linear_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                           const float* __restrict__ gate, float* __restrict__ gx,
                                                           int M, int N, int K, int ldg, int ldx, float wscale,
                                                           float slope, float gain, int accumulate) {
    __shared__ double red[16][17];
    const int kl = threadIdx.x & 15, ng = threadIdx.x >> 4;
    const int k = blockIdx.x * 16 + kl;
    const int m = blockIdx.y;
    const float* g = gy + (size_t)m * ldg;
    const float* gt = gate ? gate + (size_t)m * ldg : nullptr;
    double acc = 0.0;
    if (k < K) {
        // eight loads in flight per lane (the fp64 chain itself is short: the loop was bound by one exposed load latency per trip);
        // same order of additions
        int n = ng;
        for (; n + 7 * 16 < N; n += 8 * 16) {
            float gv[8], wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                gv[u] = g[n + u * 16];
                if (gt) gv[u] *= (gt[n + u * 16] > 0.f ? gain : gain * slope);
                wv[u] = w[(size_t)(n + u * 16) * K + k];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fma((double)gv[u], (double)wv[u], acc);
        }
        for (; n < N; n += 16) {
            float gv = g[n];
            if (gt) gv *= (gt[n] > 0.f ? gain : gain * slope);
            acc = fma((double)gv, (double)w[(size_t)n * K + k], acc);
        }
    }
    red[ng][kl] = acc;
    __syncthreads();
    if (ng == 0 && k < K) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][kl];
        float* o = gx + (size_t)m * ldx + k;
        const float r = (float)(t * (double)wscale);
        *o = accumulate ? (*o + r) : r;
    }
}

// dW[n,k] = sum_m gy[m,n] x[m,k] ; db[n] = sum_m gy[m,n]   (Reconstructor heads; M = batch)
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                           float* __restrict__ dw, float* __restrict__ db, int M,
                                                           int N, int K) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (k >= K) return;
    float acc = 0.f, accb = 0.f;
    for (int m = 0; m < M; ++m) {
        const float g = gy[(size_t)m * N + n];
        acc = fmaf(g, x[(size_t)m * K + k], acc);
        accb += g;
    }
    dw[(size_t)n * K + k] = acc;
    if (db && k == 0) db[n] = accb;
}

// ---- Blur(4x4, pad (1,1)) + noise + bias + leaky-relu*sqrt(2) on NHWC: fir4.h (sliding-window FIR, EPI = true) ----

// ---- ToRGB: 1x1 modulated conv to 3 channels, no demod, + bias + up-sampled skip -> NCHW image ------
// img[b,o,p] = wscale * sum_c x[b,p,c] s[b,c] W[o,c] + bias[o] + skip[b,o,p]
// One block = 256 consecutive pixels of one sample; each wave walks its 64 pixels with LPP lanes per
// pixel (float4 over channels), shuffle-reduces, stages results in LDS, then writes coalesced planes.
__global__ __launch_bounds__(256) void torgb_fwd_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ skip, float* __restrict__ img,
                                                        int P, int C, float wscale, int ppb,
                                                        const float* __restrict__ skip_lo, const float* __restrict__ upk, int Wimg, int s_ld) {
    // ppb = pixels per block (64, 128 or 256): small maps use small blocks so that the launch still fills the chip
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* wm = sm;               // [3][C] modulated weights W[o,c]*s[b,c]*wscale
    float* res = sm + 3 * C;      // [3][256]
    float* upk_s = res + 3 * 256; // [16] flipped up-sampling taps (fetched here: in the epilogue each would be a dependent global load)
    if (skip_lo && threadIdx.x < 16) upk_s[threadIdx.x] = upk[15 - threadIdx.x];
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * ppb;
    const int ppwave = ppb >> 2;
    for (int i = threadIdx.x; i < 3 * C; i += 256) wm[i] = w[i] * s[(size_t)b * s_ld + (i % C)] * wscale;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c4n = C >> 2;
    const int lpp = c4n < 64 ? c4n : 64;   // lanes per pixel (power of two: C in {32..512})
    const int ppw = 64 / lpp;              // pixels per wave iteration
    const int sub = lane / lpp, cl = lane % lpp;
    // a lane owns the same one or two channel quads for every pixel: its modulated weights stay in registers, and four pixels'
    // loads are issued before the first is reduced (one pixel at a time left the kernel latency-bound at 4.0 TB/s)
    const int nc = C / (lpp * 4);          // 1 (C <= 256) or 2 (C = 512)
    float4 W0[2], W1[2], W2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = (cl + j * lpp) * 4;
        const bool ok = j < nc;
        W0[j] = ok ? *reinterpret_cast<const float4*>(wm + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        W1[j] = ok ? *reinterpret_cast<const float4*>(wm + C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        W2[j] = ok ? *reinterpret_cast<const float4*>(wm + 2 * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    constexpr int U = 4;
    for (int it = 0; it < ppwave; it += ppw * U) {
        float4 v[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pl = wave * ppwave + it + u * ppw + sub;
            const int p = p0 + pl;
            const bool live = it + u * ppw < ppwave && p < P;
            const float* xp = x + ((size_t)b * P + (live ? p : 0)) * C;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                v[u][j] = (live && j < nc) ? *reinterpret_cast<const float4*>(xp + (cl + j * lpp) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (it + u * ppw >= ppwave) break;
            const int pl = wave * ppwave + it + u * ppw + sub;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 t = v[u][j];
                a0 += t.x * W0[j].x + t.y * W0[j].y + t.z * W0[j].z + t.w * W0[j].w;
                a1 += t.x * W1[j].x + t.y * W1[j].y + t.z * W1[j].z + t.w * W1[j].w;
                a2 += t.x * W2[j].x + t.y * W2[j].y + t.z * W2[j].z + t.w * W2[j].w;
            }
            for (int off = lpp >> 1; off > 0; off >>= 1) {
                a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64); a2 += __shfl_xor(a2, off, 64);
            }
            if (cl == 0) { res[pl] = a0; res[256 + pl] = a1; res[512 + pl] = a2; }
        }
    }
    __syncthreads();
    const int p = p0 + threadIdx.x;
    if (p < P && threadIdx.x < ppb) {
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const size_t off = ((size_t)b * 3 + o) * P + p;
            float v = res[o * 256 + threadIdx.x] + bias[o];
            if (skip) v += skip[off];
            if (skip_lo) {
                // Upsample(skip) of model.py:257-262,279-281 = upfirdn2d(skip, k*4, up=2, pad=(2,1)) evaluated in place: of the 4x4
                // taps only the 2x2 that land on the zero-inserted grid's samples contribute
                const int oy = p / Wimg, ox = p - oy * Wimg, Hl = (P / Wimg) >> 1, Wl = Wimg >> 1;
                const float* sl = skip_lo + ((size_t)b * 3 + o) * Hl * Wl;
                float up = 0.f;
#pragma unroll
                for (int ky = 0; ky < 4; ++ky) {
                    const int Y = oy + ky - 2, i = Y >> 1;
                    if ((Y & 1) || i < 0 || i >= Hl) continue;
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const int X = ox + kx - 2, j = X >> 1;
                        if ((X & 1) || j < 0 || j >= Wl) continue;
                        up = fmaf(upk_s[ky * 4 + kx], sl[i * Wl + j], up);
                    }
                }
                v += up;
            }
            img[off] = v;
        }
    }
}

// ---- backward of one StyledConv's output tensor ----------------------------------------------------
// `out` [B,P,C] is the saved post-activation output of a StyledConv, consumed by (A) the next modulated
// conv with style sA [B,C], whose dgrad produced the UN-scaled gA [B,P,C], and optionally (R) the
// ToRGB at this resolution (style sR, weight wR [3,C]*rscale) with image gradient drgb [B,3,P].
//   dOut = sA*gA + sR * (sum_o drgb[b,o,p] wR[o,c] rscale)
//   dy   = dOut * (out > 0 ? sqrt2 : 0.2*sqrt2)                               -> written [B,P,C]
//   ypre = (out > 0 ? out/sqrt2 : out/(0.2 sqrt2)) - nw*noise[p] - bias[c]      (pre-noise conv output)
//   num[b,c]  += sum_p dy*ypre      (d demod numerator: ddemod = num/demod)
//   dsA[b,c]  += sum_p out*gA       (direct style gradient of the consumer conv)
//   dsR[b,c]  += sum_p out*gR       (direct style gradient of the ToRGB)
// grid = (pixel chunks, B); thread -> fixed float4 channel group, strides over the chunk's pixels.
template <bool UNR4>      // UNR4: four pixels' loads in flight per thread (the launches with a ToRGB branch), see the loop
__global__ __launch_bounds__(256) void sg2_act_bwd_kernel(
    const float* __restrict__ out, const float* __restrict__ gA, const float* __restrict__ sA,
    const float* __restrict__ drgb, const float* __restrict__ wR, const float* __restrict__ sR, float rscale,
    const float* __restrict__ noise, const float* __restrict__ noise_w, const float* __restrict__ bias,
    float* __restrict__ dy, float* __restrict__ num, float* __restrict__ dsA, float* __restrict__ dsR,
    const float* __restrict__ post_scale, float* __restrict__ dy_amax, int P, int C, int chunk, int s_ld,
    unsigned short* __restrict__ dy_h, const float* __restrict__ dy_bound) {
    __shared__ double red[3][256][4];
    // dy_h: store the gradient ONLY as the fp16 operand plane of the dgrad conv that consumes it (f16_rn(dy * 2^k), k from the
    // a-priori bound dy_bound >= max |stored dy|, conv_scheme.h) instead of the fp32 tensor
    float h_mult = 1.f, h_inv = 1.f;
    if (dy_h) wgsconv::operand_scale(dy_bound, nullptr, 1.f, h_mult, h_inv);
    float amax = 0.f;          // max |stored dy| seen by this thread
    const int b = blockIdx.y;
    const int c4n = C >> 2;
    const int tpp = c4n < 256 ? c4n : 256;  // threads per pixel
    const int ppi = 256 / tpp;              // pixels per block iteration
    const int cl = threadIdx.x % tpp, sub = threadIdx.x / tpp;
    const int p_begin = blockIdx.x * chunk, p_end = min(P, p_begin + chunk);
    const float nw = noise ? noise_w[0] : 0.f;
    for (int c = cl * 4; c < C; c += tpp * 4) {
        float4 sa = gA ? *reinterpret_cast<const float4*>(sA + (size_t)b * s_ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 sr = make_float4(0.f, 0.f, 0.f, 0.f), w0 = sr, w1 = sr, w2 = sr;
        if (drgb) {
            sr = *reinterpret_cast<const float4*>(sR + (size_t)b * s_ld + c);
            w0 = *reinterpret_cast<const float4*>(wR + c);
            w1 = *reinterpret_cast<const float4*>(wR + C + c);
            w2 = *reinterpret_cast<const float4*>(wR + 2 * C + c);
        }
        const float4 bv = *reinterpret_cast<const float4*>(bias + c);
        const float4 ps = post_scale ? *reinterpret_cast<const float4*>(post_scale + (size_t)b * C + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        // fp64 partial sums: these reductions run over up to 65536 pixels with heavy cancellation
        double r_num[4] = {0, 0, 0, 0}, r_a[4] = {0, 0, 0, 0}, r_r[4] = {0, 0, 0, 0};
        // one pixel: (o, ga, raw drgb triple, raw noise) -> dy store + the three partial sums.  Layers WITH a ToRGB branch (five dependent-free
        // loads per pixel, three of them 4-byte broadcasts) keep FOUR pixels' loads in flight per thread and finish them in order — the same
        // additions in the same order as the one-pixel loop: 580 -> 533 us at 128 ch @256^2, 197 -> 182 us at 512 ch @64^2 (tools/bench_actbwd.py);
        // layers without it are at 5.0 - 5.5 TB/s with one pixel per trip and lose 1 - 4 % to the larger register footprint: they keep it
        auto body = [&](int p, const float4 o, const float4 ga, float e0, float e1, float e2, float nzr) {
            const size_t off = ((size_t)b * P + p) * C + c;
            float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
            if (drgb) {
                const float d0 = e0 * rscale, d1 = e1 * rscale, d2 = e2 * rscale;
                gr.x = d0 * w0.x + d1 * w1.x + d2 * w2.x; gr.y = d0 * w0.y + d1 * w1.y + d2 * w2.y;
                gr.z = d0 * w0.z + d1 * w1.z + d2 * w2.z; gr.w = d0 * w0.w + d1 * w1.w + d2 * w2.w;
            }
            const float nz = noise ? nw * nzr : 0.f;
            float4 d;
#define WGS_ONE(f, q)                                                               \
    {                                                                               \
        const float dout = sa.f * ga.f + sr.f * gr.f;                               \
        const bool pos = o.f > 0.f;                                                 \
        d.f = dout * (pos ? SQRT2 : 0.2f * SQRT2);                                  \
        const float ypre = (pos ? o.f * (1.f / SQRT2) : o.f * (1.f / (0.2f * SQRT2))) - nz - bv.f; \
        r_num[q] += (double)(d.f * ypre);                                           \
        r_a[q] += (double)(o.f * ga.f);                                             \
        r_r[q] += (double)(o.f * gr.f);                                             \
    }
            WGS_ONE(x, 0) WGS_ONE(y, 1) WGS_ONE(z, 2) WGS_ONE(w, 3)
#undef WGS_ONE
            // stored gradient = dy * post_scale (the dgrad conv's demodulation factor: rounded fp32 product, exactly what the
            // conv kernels' A-operand prologue computes when it is handed dy and the factor separately)
            d.x = __fmul_rn(d.x, ps.x); d.y = __fmul_rn(d.y, ps.y); d.z = __fmul_rn(d.z, ps.z); d.w = __fmul_rn(d.w, ps.w);
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(d.x), fabsf(d.y))), fmaxf(fabsf(d.z), fabsf(d.w)));
            if (dy_h) {
                const wgsconv::sch_f32x4 f = {d.x * h_mult, d.y * h_mult, d.z * h_mult, d.w * h_mult};
                uint2 h, l;
                wgsconv::Scheme<1>::cvt4(f, h, l);
                *reinterpret_cast<uint2*>(dy_h + off) = h;
            } else *reinterpret_cast<float4*>(dy + off) = d;
        };
        auto fetch = [&](int p, float4& o, float4& ga, float& e0, float& e1, float& e2, float& nzr) {
            const size_t off = ((size_t)b * P + p) * C + c;
            o = *reinterpret_cast<const float4*>(out + off);
            ga = gA ? *reinterpret_cast<const float4*>(gA + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            e0 = e1 = e2 = 0.f;
            if (drgb) {
                e0 = drgb[((size_t)b * 3 + 0) * P + p]; e1 = drgb[((size_t)b * 3 + 1) * P + p]; e2 = drgb[((size_t)b * 3 + 2) * P + p];
            }
            nzr = noise ? noise[p] : 0.f;
        };
        int p = p_begin + sub;
        for (; UNR4 && p + 3 * ppi < p_end; p += 4 * ppi) {
            float4 o[4], ga[4];
            float e0[4], e1[4], e2[4], nzr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) fetch(p + u * ppi, o[u], ga[u], e0[u], e1[u], e2[u], nzr[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) body(p + u * ppi, o[u], ga[u], e0[u], e1[u], e2[u], nzr[u]);
        }
        for (; p < p_end; p += ppi) {
            float4 o, ga;
            float e0, e1, e2, nzr;
            fetch(p, o, ga, e0, e1, e2, nzr);
            body(p, o, ga, e0, e1, e2, nzr);
        }
        // combine the `ppi` pixel sub-streams that share this channel group
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            red[0][threadIdx.x][q] = r_num[q]; red[1][threadIdx.x][q] = r_a[q]; red[2][threadIdx.x][q] = r_r[q];
        }
        __syncthreads();
        if (sub == 0) {
            for (int s2 = 1; s2 < ppi; ++s2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    r_num[q] += red[0][s2 * tpp + cl][q];
                    r_a[q] += red[1][s2 * tpp + cl][q];
                    r_r[q] += red[2][s2 * tpp + cl][q];
                }
            }
            float* pn = num + (size_t)b * C + c;
#pragma unroll
            for (int q = 0; q < 4; ++q) unsafeAtomicAdd(pn + q, (float)r_num[q]);
            if (gA) {
                float* pa = dsA + (size_t)b * C + c;
#pragma unroll
                for (int q = 0; q < 4; ++q) unsafeAtomicAdd(pa + q, (float)r_a[q]);
            }
            if (drgb) {
                float* pr = dsR + (size_t)b * C + c;
#pragma unroll
                for (int q = 0; q < 4; ++q) unsafeAtomicAdd(pr + q, (float)r_r[q]);
            }
        }
    }
    if (dy_amax) {      // non-negative floats order like their bit patterns
        amax = wave_max(amax);
        if ((threadIdx.x & 63) == 0) raise_amax(dy_amax, amax);
    }
}

// ds[b,c] += sum_p x[(xb ? b : 0), p, c] * g[b,p,c]   (direct style gradient of the first conv, whose
// input is the batch-independent ConstantInput)
__global__ __launch_bounds__(256) void xg_reduce_kernel(const float* __restrict__ x, int x_batched,
                                                        const float* __restrict__ g, float* __restrict__ ds, int P,
                                                        int C) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* xp = x + (x_batched ? (size_t)b * P * C : 0);
    const float* gp = g + (size_t)b * P * C;
    float acc = 0.f;
    for (int p = 0; p < P; ++p) acc = fmaf(xp[(size_t)p * C + c], gp[(size_t)p * C + c], acc);
    ds[(size_t)b * C + c] += acc;
}

// dstyle[b,i] = dsdir[b,i] - s[b,i]*scale2 * sum_o num[b,o]*demod[b,o]^2 * wsq[o,i]
// (style gradient of a demodulated conv: direct term + the path through demod = rsqrt(scale2*sum s^2 wsq + eps))
__global__ __launch_bounds__(256) void sg2_style_grad_kernel(const float* __restrict__ num, const float* __restrict__ demod,
                                                             const float* __restrict__ s, const float* __restrict__ dsdir,
                                                             const float* __restrict__ wsq, float scale2,
                                                             float* __restrict__ dstyle, int Co, int Ci, int lds_,
                                                             int ldo) {
    // block = 64 input channels x 4 partitions of the output-channel sum (fp64, combined through LDS)
    __shared__ double red[4][64];
    const int il = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    const int b = blockIdx.y;
    double acc0 = 0.0, acc1 = 0.0;
    if (demod && i < Ci) {
        int o = part;
        // four trips' loads in flight (same two chains, same order of additions): one exposed load latency per trip bound the loop
        for (; o + 28 < Co; o += 32) {
            float dv[8], nv[8], wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                dv[u] = demod[(size_t)b * Co + o + 4 * u];
                nv[u] = num[(size_t)b * Co + o + 4 * u];
                wv[u] = wsq[(size_t)(o + 4 * u) * Ci + i];
            }
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const double d0 = dv[u], d1 = dv[u + 1];
                acc0 = fma((double)nv[u] * d0 * d0, (double)wv[u], acc0);
                acc1 = fma((double)nv[u + 1] * d1 * d1, (double)wv[u + 1], acc1);
            }
        }
        for (; o + 4 < Co; o += 8) {
            const double d0 = demod[(size_t)b * Co + o], d1 = demod[(size_t)b * Co + o + 4];
            acc0 = fma((double)num[(size_t)b * Co + o] * d0 * d0, (double)wsq[(size_t)o * Ci + i], acc0);
            acc1 = fma((double)num[(size_t)b * Co + o + 4] * d1 * d1, (double)wsq[(size_t)(o + 4) * Ci + i], acc1);
        }
        for (; o < Co; o += 4) {
            const double d0 = demod[(size_t)b * Co + o];
            acc0 = fma((double)num[(size_t)b * Co + o] * d0 * d0, (double)wsq[(size_t)o * Ci + i], acc0);
        }
    }
    red[part][il] = acc0 + acc1;
    __syncthreads();
    if (part == 0 && i < Ci) {
        const double acc = red[0][il] + red[1][il] + red[2][il] + red[3][il];
        dstyle[(size_t)b * ldo + i] = (float)((double)dsdir[(size_t)b * Ci + i] - (double)s[(size_t)b * lds_ + i] * scale2 * acc);
    }
}

// The same for up to 24 layers in ONE launch (blockIdx.z = layer): the synthesis backward has one such reduction per modulated
// conv and per ToRGB, each far too small to fill the chip; none of their results is needed before the pass's last kernel.
struct StyleGradBatchArgs {
    int n, ld_s, ld_out;
    const float* num[24]; const float* demod[24]; const float* s[24]; const float* dsdir[24]; const float* wsq[24];
    float* dstyle[24];
    int Co[24], Ci[24];
    float scale2[24];
};
__global__ __launch_bounds__(256) void sg2_style_grad_batch_kernel(const StyleGradBatchArgs a) {
    __shared__ double red[4][64];
    const int l = blockIdx.z;
    const int Co = a.Co[l], Ci = a.Ci[l];
    if ((int)blockIdx.x * 64 >= Ci) return;
    const float* __restrict__ num = a.num[l];
    const float* __restrict__ demod = a.demod[l];
    const float* __restrict__ wsq = a.wsq[l];
    const int il = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    const int b = blockIdx.y;
    double acc0 = 0.0, acc1 = 0.0;
    if (demod && i < Ci) {
        int o = part;
        // four trips' loads in flight (same two chains, same order of additions): one exposed load latency per trip bound the loop
        for (; o + 28 < Co; o += 32) {
            float dv[8], nv[8], wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                dv[u] = demod[(size_t)b * Co + o + 4 * u];
                nv[u] = num[(size_t)b * Co + o + 4 * u];
                wv[u] = wsq[(size_t)(o + 4 * u) * Ci + i];
            }
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const double d0 = dv[u], d1 = dv[u + 1];
                acc0 = fma((double)nv[u] * d0 * d0, (double)wv[u], acc0);
                acc1 = fma((double)nv[u + 1] * d1 * d1, (double)wv[u + 1], acc1);
            }
        }
        for (; o + 4 < Co; o += 8) {
            const double d0 = demod[(size_t)b * Co + o], d1 = demod[(size_t)b * Co + o + 4];
            acc0 = fma((double)num[(size_t)b * Co + o] * d0 * d0, (double)wsq[(size_t)o * Ci + i], acc0);
            acc1 = fma((double)num[(size_t)b * Co + o + 4] * d1 * d1, (double)wsq[(size_t)(o + 4) * Ci + i], acc1);
        }
        for (; o < Co; o += 4) {
            const double d0 = demod[(size_t)b * Co + o];
            acc0 = fma((double)num[(size_t)b * Co + o] * d0 * d0, (double)wsq[(size_t)o * Ci + i], acc0);
        }
    }
    red[part][il] = acc0 + acc1;
    __syncthreads();
    if (part == 0 && i < Ci) {
        const double acc = red[0][il] + red[1][il] + red[2][il] + red[3][il];
        a.dstyle[l][(size_t)b * a.ld_out + i] =
            (float)((double)a.dsdir[l][(size_t)b * Ci + i] - (double)a.s[l][(size_t)b * a.ld_s + i] * a.scale2[l] * acc);
    }
}

// wsq[o,i] = sum_t w[o,t,i]^2   (w packed [Co,T,Ci]; one-off for the frozen generator)
__global__ __launch_bounds__(256) void wsq_kernel(const float* __restrict__ w, float* __restrict__ wsq, int Co, int T, int Ci) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int o = blockIdx.y;
    if (i >= Ci) return;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) { const float v = w[((size_t)o * T + t) * Ci + i]; acc = fmaf(v, v, acc); }
    wsq[(size_t)o * Ci + i] = acc;
}

}  // namespace

extern "C" {

int wgs_pixelnorm_fwd(const float* x, float* y, int rows, int d, float eps, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && y && rows > 0 && d > 0, "wgs_pixelnorm_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
#define WGS_PN(LPR) { const long nb = ((long)rows + 256 / LPR - 1) / (256 / LPR); \
        WGS_LAUNCH(pixelnorm_fwd_vec_kernel<LPR>, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, st, x, y, (long)rows, eps); }
    if (d == 16) WGS_PN(4) else if (d == 32) WGS_PN(8) else if (d == 64) WGS_PN(16) else if (d == 128) WGS_PN(32) else if (d == 256) WGS_PN(64)
    else WGS_LAUNCH(pixelnorm_fwd_kernel, dim3(wgs_cdiv(rows, 4)), dim3(256), 0, st, x, y, rows, d, eps);
#undef WGS_PN
    WGS_CHECK_LAUNCH("pixelnorm_fwd_kernel");
    return WGS_OK;
}
static int pixelnorm_bwd_launch(const float* x, const float* gy, float* gx, int rows, int d, float eps, float act_slope, float* gx_amax, hipStream_t st) {
#define WGS_PN(LPR) { const long nb = ((long)rows + 256 / LPR - 1) / (256 / LPR); \
        WGS_LAUNCH(pixelnorm_bwd_vec_kernel<LPR>, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, st, x, gy, gx, (long)rows, eps, act_slope, gx_amax); }
    if (d == 16) WGS_PN(4) else if (d == 32) WGS_PN(8) else if (d == 64) WGS_PN(16) else if (d == 128) WGS_PN(32) else if (d == 256) WGS_PN(64)
    else WGS_LAUNCH(pixelnorm_bwd_kernel, dim3(wgs_cdiv(rows, 4)), dim3(256), 0, st, x, gy, gx, rows, d, eps, act_slope, gx_amax);
#undef WGS_PN
    WGS_CHECK_LAUNCH("pixelnorm_bwd_kernel");
    return WGS_OK;
}
int wgs_pixelnorm_bwd(const float* x, const float* gy, float* gx, int rows, int d, float eps, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && gy && gx && rows > 0 && d > 0, "wgs_pixelnorm_bwd: bad arguments");
    return pixelnorm_bwd_launch(x, gy, gx, rows, d, eps, 1.f, nullptr, (hipStream_t)stream);
}
int wgs_pixelnorm_bwd_act(const float* x, const float* gy, float* gx, int rows, int d, float eps, float act_slope, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && gy && gx && rows > 0 && d > 0, "wgs_pixelnorm_bwd_act: bad arguments");
    return pixelnorm_bwd_launch(x, gy, gx, rows, d, eps, act_slope, nullptr, (hipStream_t)stream);
}
int wgs_pixelnorm_bwd_act_amax(const float* x, const float* gy, float* gx, float* gx_amax, int rows, int d, float eps, float act_slope, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && gy && gx && gx_amax && rows > 0 && d > 0, "wgs_pixelnorm_bwd_act_amax: bad arguments");
    return pixelnorm_bwd_launch(x, gy, gx, rows, d, eps, act_slope, gx_amax, (hipStream_t)stream);
}

int wgs_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int ldx,
                   int ldy, float wscale, float bscale, int in_square, int epilogue, float eps, float out_gain,
                   wgs_stream_t stream) {
    WGS_CHECK_ARG(x && w && y, "wgs_linear_fwd: null pointer");
    WGS_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 4 == 0 && K <= 2048 && ldx % 4 == 0,
                  "wgs_linear_fwd: need K %% 4 == 0, K <= 2048, ldx %% 4 == 0 (M=%d N=%d K=%d ldx=%d)", M, N, K, ldx);
    dim3 grid(wgs_cdiv(N, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define WGS_LIN(NT) WGS_LAUNCH(linear_fwd_kernel<NT>, grid, block, 0, st, x, w, bias, y, M, N, K, ldx, ldy, wscale, bscale, in_square, epilogue, eps, out_gain)
    if (K <= 256) WGS_LIN(1); else if (K <= 512) WGS_LIN(2); else if (K <= 1024) WGS_LIN(4); else WGS_LIN(8);
#undef WGS_LIN
    WGS_CHECK_LAUNCH("linear_fwd_kernel");
    return WGS_OK;
}

int wgs_mapping_mlp_fwd(const float* z, const float* const* w, const float* const* bias, float* acts, int B, int d, int L,
                        float wscale, float lr_mul, float eps, wgs_stream_t stream) {
    WGS_CHECK_ARG(z && w && bias && acts && B > 0, "wgs_mapping_mlp_fwd: bad arguments");
    WGS_CHECK_ARG(d == MLP_D && L >= 1 && L <= 16, "wgs_mapping_mlp_fwd: d = %d (must be %d), L = %d (1..16)", d, MLP_D, L);
    MappingArgs a;
    a.z = z; a.acts = acts; a.B = B; a.d = d; a.L = L; a.wscale = wscale; a.lr_mul = lr_mul; a.eps = eps;
    for (int l = 0; l < 16; ++l) { a.w[l] = l < L ? w[l] : nullptr; a.b[l] = l < L ? bias[l] : nullptr; }
    for (int l = 0; l < L; ++l) WGS_CHECK_ARG(a.w[l] && a.b[l], "wgs_mapping_mlp_fwd: null layer %d", l);
    WGS_LAUNCH(mapping_mlp_fwd_kernel, dim3(wgs_cdiv(B, MLP_RPB)), dim3(512), 0, (hipStream_t)stream, a);
    WGS_CHECK_LAUNCH("mapping_mlp_fwd_kernel");
    return WGS_OK;
}

int wgs_mapping_mlp_bwd(const float* gw, const float* const* w, const float* acts, float* gx, int B, int d, int L, float wscale,
                        wgs_stream_t stream) {
    WGS_CHECK_ARG(gw && w && acts && gx && B > 0, "wgs_mapping_mlp_bwd: null pointer");
    WGS_CHECK_ARG(d == MLP_D && L >= 1 && L <= 16, "wgs_mapping_mlp_bwd: d = %d (must be %d), L = %d (1..16)", d, MLP_D, L);
    MappingBwdArgs a;
    a.gw = gw; a.acts = acts; a.gx = gx; a.B = B; a.L = L; a.wscale = wscale;
    for (int l = 0; l < 16; ++l) a.w[l] = l < L ? w[l] : nullptr;
    for (int l = 0; l < L; ++l) WGS_CHECK_ARG(a.w[l], "wgs_mapping_mlp_bwd: null layer %d", l);
    WGS_LAUNCH(mapping_mlp_bwd_kernel, dim3(wgs_cdiv(B, MLP_RPB)), dim3(MLP_D), 0, (hipStream_t)stream, a);
    WGS_CHECK_LAUNCH("mapping_mlp_bwd_kernel");
    return WGS_OK;
}

int wgs_linear_fwd_batch(const wgs_linear_batch* b, wgs_stream_t stream) {
    WGS_CHECK_ARG(b && b->n > 0 && b->n <= 16 && b->M > 0, "wgs_linear_fwd_batch: 1..16 layers");
    LinearBatchArgs a;
    a.n = b->n; a.M = b->M; a.in_square = b->in_square; a.epi = b->epilogue;
    int nmax = 0;
    for (int i = 0; i < b->n; ++i) {
        WGS_CHECK_ARG(b->x[i] && b->w[i] && b->y[i] && b->N[i] > 0 && b->K[i] > 0 && b->K[i] % 4 == 0 && b->K[i] <= 512 && b->ldx[i] % 4 == 0,
                      "wgs_linear_fwd_batch: layer %d needs K %% 4 == 0, K <= 512, ldx %% 4 == 0", i);
        a.x[i] = b->x[i]; a.w[i] = b->w[i]; a.y[i] = b->y[i]; a.N[i] = b->N[i]; a.K[i] = b->K[i]; a.ldx[i] = b->ldx[i]; a.ldy[i] = b->ldy[i];
        a.wscale[i] = b->wscale[i]; a.eps[i] = b->eps[i]; a.out_gain[i] = b->out_gain[i];
        nmax = b->N[i] > nmax ? b->N[i] : nmax;
    }
    WGS_LAUNCH(linear_fwd_batch_kernel, dim3(wgs_cdiv(nmax, 4), b->n), dim3(256), 0, (hipStream_t)stream, a);
    WGS_CHECK_LAUNCH("linear_fwd_batch_kernel");
    return WGS_OK;
}

int wgs_linear_dgrad(const float* gy, const float* w, const float* gate_y, float* gx, int M, int N, int K, int ldg,
                     int ldx, float wscale, float gate_slope, float gate_gain, int accumulate, wgs_stream_t stream) {
    WGS_CHECK_ARG(gy && w && gx && M > 0 && N > 0 && K > 0, "wgs_linear_dgrad: bad arguments");
    WGS_LAUNCH(linear_dgrad_kernel, dim3(wgs_cdiv(K, 16), M), dim3(256), 0, (hipStream_t)stream, gy, w, gate_y,
                       gx, M, N, K, ldg, ldx, wscale, gate_slope, gate_gain, accumulate);
    WGS_CHECK_LAUNCH("linear_dgrad_kernel");
    return WGS_OK;
}

int wgs_linear_wgrad(const float* gy, const float* x, float* dw, float* db, int M, int N, int K, wgs_stream_t stream) {
    WGS_CHECK_ARG(gy && x && dw && M > 0 && N > 0 && K > 0, "wgs_linear_wgrad: bad arguments");
    WGS_LAUNCH(linear_wgrad_kernel, dim3(wgs_cdiv(K, 256), N), dim3(256), 0, (hipStream_t)stream, gy, x, dw, db, M, N, K);
    WGS_CHECK_LAUNCH("linear_wgrad_kernel");
    return WGS_OK;
}

int wgs_sg2_blur_noise_bias_act(const float* x, const float* kernel4x4, const float* noise, const float* noise_w,
                                const float* bias, float* y, float* y_amax, int B, int Ho, int Wo, int C, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && kernel4x4 && bias && y, "wgs_sg2_blur_noise_bias_act: null pointer");
    WGS_CHECK_ARG(B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0, "wgs_sg2_blur_noise_bias_act: bad sizes (C %% 4)");
    WGS_CHECK_ARG(!noise || noise_w, "wgs_sg2_blur_noise_bias_act: noise needs noise_w");
    wgsfir::launch_fir4<true>(x, kernel4x4, y, B, Ho + 1, Wo + 1, Ho, Wo, C, 1, 1, noise, noise_w, bias, (hipStream_t)stream, y_amax);
    WGS_CHECK_LAUNCH("fir4_kernel<epilogue>");
    return WGS_OK;
}

int wgs_sg2_blur_bwd_f16(const float* dy, const float* kernel4x4, uint16_t* dt_hi, const float* a_amax, float a_bound,
                         int B, int H, int W, int C, wgs_stream_t stream) {
    WGS_CHECK_ARG(dy && kernel4x4 && dt_hi && a_amax, "wgs_sg2_blur_bwd_f16: null pointer");
    WGS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "wgs_sg2_blur_bwd_f16: bad sizes (C %% 4)");
    WGS_CHECK_ARG(a_bound > 0.f, "wgs_sg2_blur_bwd_f16: a_bound must be positive");
    wgsfir::launch_fir4<false, true>(dy, kernel4x4, reinterpret_cast<float*>(dt_hi), B, H, W, H + 1, W + 1, C, 2, 2, nullptr, nullptr,
                                     nullptr, (hipStream_t)stream, nullptr, a_amax, a_bound);
    WGS_CHECK_LAUNCH("fir4_kernel<f16 plane>");
    return WGS_OK;
}

int wgs_sg2_blur_bwd_f16_x16(const uint16_t* dy_hi, const float* kernel4x4, uint16_t* dt_hi, const float* a_amax, float a_bound,
                             int B, int H, int W, int C, wgs_stream_t stream) {
    WGS_CHECK_ARG(dy_hi && kernel4x4 && dt_hi && a_amax, "wgs_sg2_blur_bwd_f16_x16: null pointer");
    WGS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "wgs_sg2_blur_bwd_f16_x16: bad sizes (C %% 4)");
    WGS_CHECK_ARG(a_bound > 0.f, "wgs_sg2_blur_bwd_f16_x16: a_bound must be positive");
    wgsfir::launch_fir4<false, true, true>(reinterpret_cast<const float*>(dy_hi), kernel4x4, reinterpret_cast<float*>(dt_hi), B, H, W, H + 1, W + 1, C,
                                           2, 2, nullptr, nullptr, nullptr, (hipStream_t)stream, nullptr, a_amax, a_bound);
    WGS_CHECK_LAUNCH("fir4_kernel<f16 plane from an f16 plane>");
    return WGS_OK;
}

int wgs_sg2_torgb_fwd(const float* x, const float* s, const float* w, const float* bias, const float* skip, float* img,
                      int B, int P, int C, float wscale, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && s && w && bias && img, "wgs_sg2_torgb_fwd: null pointer");
    WGS_CHECK_ARG(B > 0 && P > 0 && C >= 4 && C <= 512 && (C & (C - 1)) == 0, "wgs_sg2_torgb_fwd: C=%d must be a power of two in [4, 512]", C);
    const size_t smem = (size_t)(3 * C + 3 * 256 + 16) * sizeof(float);
    int ppb = 256;
    while (ppb > 64 && (long)wgs_cdiv(P, ppb) * B < 8192) ppb >>= 1;      // >= 4 rounds of ~8 workgroups per CU: prologue / epilogue
                                                                          // of one round under the streaming of the next
    if (C < 32) ppb = 256;                // a wave iteration covers 64 / (C/4) pixels: needs ppb/4 >= that
    WGS_LAUNCH(torgb_fwd_kernel, dim3(wgs_cdiv(P, ppb), B), dim3(256), smem, (hipStream_t)stream, x, s, w, bias, skip, img, P, C, wscale, ppb,
                       (const float*)nullptr, (const float*)nullptr, 0, C);
    WGS_CHECK_LAUNCH("torgb_fwd_kernel");
    return WGS_OK;
}

int wgs_sg2_torgb_up_fwd(const float* x, const float* s, int s_ld, const float* w, const float* bias, const float* skip_lo,
                         const float* up_kernel4x4, float* img, int B, int H, int W, int C, float wscale, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && s && w && bias && img && skip_lo && up_kernel4x4, "wgs_sg2_torgb_up_fwd: null pointer");
    WGS_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C >= 4 && C <= 512 && (C & (C - 1)) == 0,
                  "wgs_sg2_torgb_up_fwd: even H, W and a power-of-two C >= 4 (H=%d W=%d C=%d)", H, W, C);
    const int P = H * W;
    const size_t smem = (size_t)(3 * C + 3 * 256 + 16) * sizeof(float);
    int ppb = 256;
    while (ppb > 64 && (long)wgs_cdiv(P, ppb) * B < 8192) ppb >>= 1;      // >= 4 rounds of ~8 workgroups per CU: prologue / epilogue
                                                                          // of one round under the streaming of the next
    if (C < 32) ppb = 256;
    WGS_LAUNCH(torgb_fwd_kernel, dim3(wgs_cdiv(P, ppb), B), dim3(256), smem, (hipStream_t)stream, x, s, w, bias,
                       (const float*)nullptr, img, P, C, wscale, ppb, skip_lo, up_kernel4x4, W, s_ld > 0 ? s_ld : C);
    WGS_CHECK_LAUNCH("torgb_fwd_kernel<up>");
    return WGS_OK;
}

static int sg2_act_bwd_launch(const char* name, const float* out, const float* gA, const float* sA, const float* drgb, const float* wR,
                              const float* sR, float rscale, const float* noise, const float* noise_w, const float* bias,
                              float* dy, float* num, float* dsA, float* dsR, const float* post_scale, float* dy_amax, int B, int P, int C,
                              int s_ld, uint16_t* dy_h, const float* dy_bound, wgs_stream_t stream) {
    WGS_CHECK_ARG(out && bias && (dy || dy_h) && num, "%s: null pointer", name);
    WGS_CHECK_ARG(gA || drgb, "%s: needs at least one gradient source", name);
    WGS_CHECK_ARG(!gA || (sA && dsA), "%s: gA needs sA and dsA", name);
    WGS_CHECK_ARG(!drgb || (wR && sR && dsR), "%s: drgb needs wR, sR, dsR", name);
    WGS_CHECK_ARG(!noise || noise_w, "%s: noise needs noise_w", name);
    WGS_CHECK_ARG(B > 0 && P > 0 && C >= 4 && (C & (C - 1)) == 0, "%s: C=%d must be a power of two >= 4", name, C);
    // ~2048 blocks in total; each block owns `chunk` pixels of one sample
    int chunks = wgs_cdiv(2048, B);
    int chunk = wgs_cdiv(P, chunks);
    if (chunk < 16) chunk = 16;
    chunks = wgs_cdiv(P, chunk);
    if (drgb)
        WGS_LAUNCH(sg2_act_bwd_kernel<true>, dim3(chunks, B), dim3(256), 0, (hipStream_t)stream, out, gA, sA, drgb, wR, sR,
                   rscale, noise, noise_w, bias, dy, num, dsA, dsR, post_scale, dy_amax, P, C, chunk, s_ld > 0 ? s_ld : C,
                   reinterpret_cast<unsigned short*>(dy_h), dy_bound);
    else
        WGS_LAUNCH(sg2_act_bwd_kernel<false>, dim3(chunks, B), dim3(256), 0, (hipStream_t)stream, out, gA, sA, drgb, wR, sR,
                   rscale, noise, noise_w, bias, dy, num, dsA, dsR, post_scale, dy_amax, P, C, chunk, s_ld > 0 ? s_ld : C,
                   reinterpret_cast<unsigned short*>(dy_h), dy_bound);
    WGS_CHECK_LAUNCH("sg2_act_bwd_kernel");
    return WGS_OK;
}

int wgs_sg2_act_bwd(const float* out, const float* gA, const float* sA, const float* drgb, const float* wR,
                    const float* sR, float rscale, const float* noise, const float* noise_w, const float* bias,
                    float* dy, float* num, float* dsA, float* dsR, const float* post_scale, float* dy_amax, int B, int P, int C,
                    int s_ld, wgs_stream_t stream) {
    WGS_CHECK_ARG(dy, "wgs_sg2_act_bwd: null pointer");
    return sg2_act_bwd_launch("wgs_sg2_act_bwd", out, gA, sA, drgb, wR, sR, rscale, noise, noise_w, bias, dy, num, dsA, dsR, post_scale,
                              dy_amax, B, P, C, s_ld, nullptr, nullptr, stream);
}

int wgs_sg2_act_bwd_f16(const float* out, const float* gA, const float* sA, const float* drgb, const float* wR,
                        const float* sR, float rscale, const float* noise, const float* noise_w, const float* bias,
                        uint16_t* dy_hi, const float* dy_bound, float* num, float* dsA, float* dsR, const float* post_scale,
                        float* dy_amax, int B, int P, int C, int s_ld, wgs_stream_t stream) {
    WGS_CHECK_ARG(dy_hi && dy_bound, "wgs_sg2_act_bwd_f16: null pointer");
    return sg2_act_bwd_launch("wgs_sg2_act_bwd_f16", out, gA, sA, drgb, wR, sR, rscale, noise, noise_w, bias, nullptr, num, dsA, dsR,
                              post_scale, dy_amax, B, P, C, s_ld, dy_hi, dy_bound, stream);
}

// bound[0] = sqrt(2) * max_{b,c} post_scale[b,c] * ( |sA[b,c]| * gA_amax + |sR[b,c]| * rscale * drgb_amax * drgb_factor * sum_o |wR[o,c]| )
// >= max |dy * post_scale| of the sg2_act_bwd launch with the same operands (|lrelu'| * sqrt(2) <= sqrt(2)): one workgroup.
__global__ __launch_bounds__(1024) void sg2_dy_bound_kernel(const float* __restrict__ gA_amax, const float* __restrict__ sA,
                                                            const float* __restrict__ drgb_amax, float drgb_factor,
                                                            const float* __restrict__ wR, const float* __restrict__ sR, float rscale,
                                                            const float* __restrict__ post_scale, float* __restrict__ bound,
                                                            int B, int C, int s_ld) {
    // one workgroup of 16 waves, four elements per lane and trip with their loads issued together: the kernel sits in the generator
    // backward's dependent chain three times per step, and with 256 lanes taking one element per trip (64 trips of 3 - 6 dependent
    // loads at B = 32, C = 512) it took 50 - 95 us.  A maximum: any order gives the same value.
    __shared__ float red[16];
    const float a = gA_amax ? gA_amax[0] : 0.f;
    const float r = drgb_amax ? drgb_amax[0] * drgb_factor * rscale : 0.f;
    const int n = B * C;
    float m = 0.f;
    for (int e0 = threadIdx.x; e0 < n; e0 += 4 * 1024) {
        float va[4], vr[4], vw[4], vp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 1024;
            const bool ok = e < n;
            const int b = ok ? e / C : 0, c = ok ? e - b * C : 0;
            va[u] = (ok && gA_amax) ? fabsf(sA[(size_t)b * s_ld + c]) : 0.f;
            vr[u] = (ok && drgb_amax) ? fabsf(sR[(size_t)b * s_ld + c]) : 0.f;
            vw[u] = (ok && drgb_amax) ? fabsf(wR[c]) + fabsf(wR[C + c]) + fabsf(wR[2 * C + c]) : 0.f;
            vp[u] = (ok && post_scale) ? fabsf(post_scale[(size_t)b * C + c]) : (ok ? 1.f : 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = 0.f;
            if (gA_amax) v += va[u] * a;
            if (drgb_amax) v += vr[u] * r * vw[u];
            v *= vp[u];
            m = fmaxf(m, v);
        }
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = red[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) t = fmaxf(t, red[i]);
        bound[0] = t * SQRT2;
    }
}

int wgs_sg2_dy_bound(const float* gA_amax, const float* sA, const float* drgb_amax, float drgb_factor, const float* wR,
                     const float* sR, float rscale, const float* post_scale, float* bound, int B, int C, int s_ld,
                     wgs_stream_t stream) {
    WGS_CHECK_ARG(bound && (gA_amax || drgb_amax), "wgs_sg2_dy_bound: null pointer");
    WGS_CHECK_ARG(!gA_amax || sA, "wgs_sg2_dy_bound: gA_amax needs sA");
    WGS_CHECK_ARG(!drgb_amax || (wR && sR), "wgs_sg2_dy_bound: drgb_amax needs wR and sR");
    WGS_CHECK_ARG(B > 0 && C > 0, "wgs_sg2_dy_bound: bad sizes");
    WGS_LAUNCH(sg2_dy_bound_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, gA_amax, sA, drgb_amax, drgb_factor, wR, sR,
                       rscale, post_scale, bound, B, C, s_ld > 0 ? s_ld : C);
    WGS_CHECK_LAUNCH("sg2_dy_bound_kernel");
    return WGS_OK;
}

int wgs_xg_reduce(const float* x, int x_batched, const float* g, float* ds, int B, int P, int C, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && g && ds && B > 0 && P > 0 && C > 0, "wgs_xg_reduce: bad arguments");
    WGS_LAUNCH(xg_reduce_kernel, dim3(wgs_cdiv(C, 256), B), dim3(256), 0, (hipStream_t)stream, x, x_batched, g, ds, P, C);
    WGS_CHECK_LAUNCH("xg_reduce_kernel");
    return WGS_OK;
}

int wgs_sg2_style_grad(const float* num, const float* demod, const float* s, const float* dsdir, const float* wsq,
                       float scale2, float* dstyle, int B, int Co, int Ci, int ld_s, int ld_out, wgs_stream_t stream) {
    WGS_CHECK_ARG(s && dsdir && dstyle && B > 0 && Co > 0 && Ci > 0, "wgs_sg2_style_grad: bad arguments");
    WGS_CHECK_ARG(!demod || (num && wsq), "wgs_sg2_style_grad: demod needs num and wsq");
    WGS_LAUNCH(sg2_style_grad_kernel, dim3(wgs_cdiv(Ci, 64), B), dim3(256), 0, (hipStream_t)stream, num, demod, s,
                       dsdir, wsq, scale2, dstyle, Co, Ci, ld_s, ld_out);
    WGS_CHECK_LAUNCH("sg2_style_grad_kernel");
    return WGS_OK;
}

int wgs_sg2_style_grad_batch(const wgs_style_grad_batch* d, wgs_stream_t stream) {
    WGS_CHECK_ARG(d && d->n > 0 && d->n <= 24 && d->B > 0, "wgs_sg2_style_grad_batch: 1..24 layers");
    StyleGradBatchArgs a;
    a.n = d->n; a.ld_s = d->ld_s; a.ld_out = d->ld_out;
    int cimax = 0;
    for (int l = 0; l < d->n; ++l) {
        WGS_CHECK_ARG(d->s[l] && d->dsdir[l] && d->dstyle[l] && d->Co[l] > 0 && d->Ci[l] > 0, "wgs_sg2_style_grad_batch: bad layer %d", l);
        WGS_CHECK_ARG(!d->demod[l] || (d->num[l] && d->wsq[l]), "wgs_sg2_style_grad_batch: layer %d: demod needs num and wsq", l);
        a.num[l] = d->num[l]; a.demod[l] = d->demod[l]; a.s[l] = d->s[l]; a.dsdir[l] = d->dsdir[l]; a.wsq[l] = d->wsq[l];
        a.dstyle[l] = d->dstyle[l]; a.Co[l] = d->Co[l]; a.Ci[l] = d->Ci[l]; a.scale2[l] = d->scale2[l];
        cimax = d->Ci[l] > cimax ? d->Ci[l] : cimax;
    }
    WGS_LAUNCH(sg2_style_grad_batch_kernel, dim3(wgs_cdiv(cimax, 64), d->B, d->n), dim3(256), 0, (hipStream_t)stream, a);
    WGS_CHECK_LAUNCH("sg2_style_grad_batch_kernel");
    return WGS_OK;
}

int wgs_sg2_wsq(const float* w_packed, float* wsq, int Co, int T, int Ci, wgs_stream_t stream) {
    WGS_CHECK_ARG(w_packed && wsq && Co > 0 && T > 0 && Ci > 0, "wgs_sg2_wsq: bad arguments");
    WGS_LAUNCH(wsq_kernel, dim3(wgs_cdiv(Ci, 256), Co), dim3(256), 0, (hipStream_t)stream, w_packed, wsq, Co, T, Ci);
    WGS_CHECK_LAUNCH("wsq_kernel");
    return WGS_OK;
}

}  // extern "C"
