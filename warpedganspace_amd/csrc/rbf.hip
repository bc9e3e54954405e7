// RBF warping field on gfx950: SupportSets.forward / backward / all-path traversal.
// Reference semantics: lib/support_sets.py:81-101 (forward), its autograd backward, and the walk
// loops of traverse_latent_space.py:361-438.  HBM-bound: one row of SUPPORT_SETS (n2*d floats) is
// streamed once per sample with 16-B coalesced loads, squared distances are reduced with wave64
// shuffles, cross-wave partial fields are combined through LDS.
//
// Work decomposition (training fwd/bwd): grid = (S splits of the n2 support vectors, B samples),
// 256 threads = 4 waves; wave w of split sp owns vectors sp*VPS + w, +4, ...  Lane l owns the
// float4s at element offsets 4*l + 256*t, t < NT (so d <= 256*NT).  B*S workgroups keep all 256
// CUs streaming even at B = 32.
#include "wgs_common.h"
#include "../../include/wgs.h"

namespace {

constexpr int RBF_THREADS = 256;
constexpr int RBF_WAVES = 4;

inline int rbf_splits(int B, int n2) {
    // >= 4 vectors per split (one per wave); cap the grid near 2048 workgroups.
    int S = (n2 + 3) / 4;
    while (S > 1 && (long)S * B > 2048) S = (S + 1) / 2;
    return S < 1 ? 1 : S;
}

struct RbfWs {
    float* g_raw;    // [B,d]   un-normalised field
    float* gnorm;    // [B]
    float* r2;       // [B,n2]  squared distances
    float* partial;  // [B,S,d]
};
__host__ __device__ inline RbfWs rbf_ws(float* ws, int B, int n2, int d) {
    RbfWs w;
    w.g_raw = ws;
    w.gnorm = w.g_raw + (size_t)B * d;
    w.r2 = w.gnorm + ((B + 3) & ~3);
    w.partial = w.r2 + (((size_t)B * n2 + 3) & ~(size_t)3);
    return w;
}

template <int NT>
__device__ __forceinline__ void load_vec(float4 (&v)[NT], const float* __restrict__ p, int lane, int d) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = 4 * lane + 256 * t;
        v[t] = (j < d) ? *reinterpret_cast<const float4*>(p + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ void fma4(float4& acc, float c, float4 v) {
    acc.x = fmaf(c, v.x, acc.x); acc.y = fmaf(c, v.y, acc.y); acc.z = fmaf(c, v.z, acc.z); acc.w = fmaf(c, v.w, acc.w);
}

// Accumulate sum_i c_i * (z - s_i) over the vectors [i0,i1) owned by this workgroup into acc (per
// wave), where c_i = -2*alpha_i*gamma*exp(-gamma*|z - s_i|^2).  `srow` may point to global or LDS.
template <int NT, bool SAVE_R2>
__device__ __forceinline__ void rbf_accumulate(float4 (&acc)[NT], const float4 (&zr)[NT],
                                               const float* __restrict__ srow,
                                               const float* __restrict__ arow, float gamma, int i0,
                                               int i1, int d, int wave, int lane, float* r2_out) {
    for (int i = i0 + wave; i < i1; i += RBF_WAVES) {
        float4 sv[NT];
        load_vec<NT>(sv, srow + (size_t)i * d, lane, d);
        float p = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            sv[t] = sub4(zr[t], sv[t]);  // D = z - s
            p += dot4(sv[t], sv[t]);
        }
        const float r2 = wave_sum(p);
        const float c = -2.f * arow[i] * gamma * expf(-gamma * r2);
#pragma unroll
        for (int t = 0; t < NT; ++t) fma4(acc[t], c, sv[t]);
        if (SAVE_R2 && lane == 0) r2_out[i] = r2;
    }
}

// ---------------------------------------------------------------------------------------------
// forward, stage 1: partial field of one split of the support vectors.
template <int NT>
__global__ __launch_bounds__(RBF_THREADS) void rbf_fwd_partial(
    const float* __restrict__ table, const float* __restrict__ alphas,
    const float* __restrict__ loggamma, float gamma_c, const int64_t* __restrict__ idx,
    const float* __restrict__ z, float* __restrict__ ws, int B, int n2, int d, int S) {
    __shared__ float4 red[RBF_WAVES][64 * NT];
    const int sp = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = (int)idx[b];
    const float gamma = loggamma ? expf(loggamma[k]) : gamma_c;
    const RbfWs w = rbf_ws(ws, B, n2, d);
    const int vps = (n2 + S - 1) / S;
    const int i0 = sp * vps, i1 = min(n2, i0 + vps);

    float4 zr[NT], acc[NT];
    load_vec<NT>(zr, z + (size_t)b * d, lane, d);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);

    rbf_accumulate<NT, true>(acc, zr, table + (size_t)k * n2 * d, alphas + (size_t)k * n2, gamma, i0,
                             i1, d, wave, lane, w.r2 + (size_t)b * n2);

#pragma unroll
    for (int t = 0; t < NT; ++t) red[wave][lane + 64 * t] = acc[t];
    __syncthreads();
    // 256 threads sum the 4 wave partials; thread j owns float4 slot j, j+256, ...
    float* pout = w.partial + ((size_t)b * S + sp) * d;
    for (int slot = threadIdx.x; slot < 64 * NT; slot += RBF_THREADS) {
        const int l = slot & 63, t = slot >> 6;
        const int j = 4 * l + 256 * t;
        if (j < d) {
            float4 s = red[0][slot];
#pragma unroll
            for (int ww = 1; ww < RBF_WAVES; ++ww) {
                const float4 o = red[ww][slot];
                s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
            }
            *reinterpret_cast<float4*>(pout + j) = s;
        }
    }
}

// forward, stage 2: combine splits, normalise, optional per-sample scale.
__global__ __launch_bounds__(RBF_THREADS) void rbf_fwd_finish(float* __restrict__ ws,
                                                              const float* __restrict__ scale,
                                                              float* __restrict__ out, int B, int n2,
                                                              int d, int S) {
    __shared__ float red[RBF_WAVES];
    const int b = blockIdx.x;
    const RbfWs w = rbf_ws(ws, B, n2, d);
    const float* p = w.partial + (size_t)b * S * d;
    // each thread owns elements j = tid, tid+256, ... (d <= 2048 -> <= 8 per thread)
    float g[8];
    float nn = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int j = threadIdx.x + RBF_THREADS * t;
        float s = 0.f;
        if (j < d)
            for (int sp = 0; sp < S; ++sp) s += p[(size_t)sp * d + j];
        g[t] = s;
        nn = fmaf(s, s, nn);
    }
    const float norm = sqrtf(block_sum<RBF_WAVES>(nn, red));
    const float f = (scale ? scale[b] : 1.f) / norm;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int j = threadIdx.x + RBF_THREADS * t;
        if (j < d) {
            w.g_raw[(size_t)b * d + j] = g[t];
            out[(size_t)b * d + j] = g[t] * f;
        }
    }
    if (threadIdx.x == 0) w.gnorm[b] = norm;
}

// forward in ONE launch: a workgroup of 16 waves owns a sample — wave w streams support vectors w, w + 16, ... of the selected row (each a
// coalesced 16-B-per-lane read), the 16 partial fields meet in LDS, the norm is a block reduction.  The two-launch form above spreads a
// sample over S workgroups to keep all CUs streaming, but at the training batch the whole table row is 131 KB per sample and the step is
// launch latency, not bandwidth: 2 launches of ~8 us against one.
constexpr int RBF1_WAVES = 16;
template <int NT>
__global__ __launch_bounds__(64 * RBF1_WAVES) void rbf_fwd_one(
    const float* __restrict__ table, const float* __restrict__ alphas, const float* __restrict__ loggamma, float gamma_c,
    const int64_t* __restrict__ idx, const float* __restrict__ z, const float* __restrict__ scale, float* __restrict__ out,
    float* __restrict__ ws, int B, int n2, int d) {
    __shared__ float4 red[RBF1_WAVES][64 * NT];
    __shared__ float nred[RBF1_WAVES];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = (int)idx[b];
    const float gamma = loggamma ? expf(loggamma[k]) : gamma_c;
    const RbfWs w = rbf_ws(ws, B, n2, d);
    const float* srow = table + (size_t)k * n2 * d;
    const float* arow = alphas + (size_t)k * n2;
    float4 zr[NT], acc[NT];
    load_vec<NT>(zr, z + (size_t)b * d, lane, d);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = wave; i < n2; i += RBF1_WAVES) {
        float4 sv[NT];
        load_vec<NT>(sv, srow + (size_t)i * d, lane, d);
        float p = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            sv[t] = sub4(zr[t], sv[t]);
            p += dot4(sv[t], sv[t]);
        }
        const float r2 = wave_sum(p);
        const float c = -2.f * arow[i] * gamma * expf(-gamma * r2);
#pragma unroll
        for (int t = 0; t < NT; ++t) fma4(acc[t], c, sv[t]);
        if (lane == 0) w.r2[(size_t)b * n2 + i] = r2;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) red[wave][lane + 64 * t] = acc[t];
    __syncthreads();
    // threads 0 .. 64 NT - 1 each finish one float4 slot of the field; everybody joins the norm reduction
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int slot = threadIdx.x;
    const int jl = 4 * (slot & 63) + 256 * (slot >> 6);
    const bool live = slot < 64 * NT && jl < d;
    if (live) {
        g4 = red[0][slot];
#pragma unroll
        for (int ww = 1; ww < RBF1_WAVES; ++ww) {
            const float4 o = red[ww][slot];
            g4.x += o.x; g4.y += o.y; g4.z += o.z; g4.w += o.w;
        }
    }
    float nn = wave_sum(dot4(g4, g4));
    if (lane == 0) nred[wave] = nn;
    __syncthreads();
    nn = 0.f;
#pragma unroll
    for (int ww = 0; ww < RBF1_WAVES; ++ww) nn += nred[ww];
    const float norm = sqrtf(nn);
    const float f = (scale ? scale[b] : 1.f) / norm;
    if (live) {
        *reinterpret_cast<float4*>(w.g_raw + (size_t)b * d + jl) = g4;
        *reinterpret_cast<float4*>(out + (size_t)b * d + jl) = make_float4(g4.x * f, g4.y * f, g4.z * f, g4.w * f);
    }
    if (threadIdx.x == 0) w.gnorm[b] = norm;
}

// ---------------------------------------------------------------------------------------------
// backward.  With u = g/|g| and go = dL/du (times scale[b]):  h = dL/dg = (go - u (u.go)) / |g|.
//   dL/ds_i      =  2 a_i y e_i (h - 2 y (h.D_i) D_i)        (y = gamma, D_i = z - s_i)
//   dL/dalpha_i  = -2 y e_i (h.D_i)
//   dL/dloggamma =  y * sum_i -2 a_i (h.D_i) e_i (1 - y r2_i)
//   dL/dz        = -sum_i dL/ds_i
template <int NT>
__global__ __launch_bounds__(RBF_THREADS) void rbf_bwd_kernel(
    const float* __restrict__ table, const float* __restrict__ alphas,
    const float* __restrict__ loggamma, float gamma_c, const int64_t* __restrict__ idx,
    const float* __restrict__ z, const float* __restrict__ scale, const float* __restrict__ gout,
    const float* __restrict__ ws_c, float* __restrict__ dtable, float* __restrict__ dloggamma,
    float* __restrict__ dalphas, float* __restrict__ dz, int B, int n2, int d, int S) {
    __shared__ float red[RBF_WAVES];
    __shared__ float4 zred[RBF_WAVES][64 * NT];
    const int sp = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = (int)idx[b];
    const float gamma = loggamma ? expf(loggamma[k]) : gamma_c;
    const RbfWs w = rbf_ws(const_cast<float*>(ws_c), B, n2, d);
    const int vps = (n2 + S - 1) / S;
    const int i0 = sp * vps, i1 = min(n2, i0 + vps);
    const float gnorm = w.gnorm[b];
    const float inv = 1.f / gnorm;
    const float sc = scale ? scale[b] : 1.f;

    float4 zr[NT], h[NT], u[NT];
    load_vec<NT>(zr, z + (size_t)b * d, lane, d);
    load_vec<NT>(u, w.g_raw + (size_t)b * d, lane, d);
    load_vec<NT>(h, gout + (size_t)b * d, lane, d);
    float p = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        u[t].x *= inv; u[t].y *= inv; u[t].z *= inv; u[t].w *= inv;
        h[t].x *= sc; h[t].y *= sc; h[t].z *= sc; h[t].w *= sc;
        p += dot4(u[t], h[t]);
    }
    // every wave holds the full vector, so a wave reduction already gives u.go
    const float ugo = wave_sum(p);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        h[t].x = (h[t].x - u[t].x * ugo) * inv;
        h[t].y = (h[t].y - u[t].y * ugo) * inv;
        h[t].z = (h[t].z - u[t].z * ugo) * inv;
        h[t].w = (h[t].w - u[t].w * ugo) * inv;
    }

    const float* srow = table + (size_t)k * n2 * d;
    const float* arow = alphas + (size_t)k * n2;
    float* drow = dtable + (size_t)k * n2 * d;
    float4 dzacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) dzacc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    float dgam = 0.f;

    for (int i = i0 + wave; i < i1; i += RBF_WAVES) {
        float4 D[NT];
        load_vec<NT>(D, srow + (size_t)i * d, lane, d);
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            D[t] = sub4(zr[t], D[t]);
            q += dot4(h[t], D[t]);
        }
        const float hd = wave_sum(q);
        const float r2 = w.r2[(size_t)b * n2 + i];
        const float e = expf(-gamma * r2);
        const float a = arow[i];
        const float coef = 2.f * a * gamma * e;
        const float c2 = -2.f * gamma * hd;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = 4 * lane + 256 * t;
            if (j < d) {
                float4 ds;
                ds.x = coef * fmaf(c2, D[t].x, h[t].x);
                ds.y = coef * fmaf(c2, D[t].y, h[t].y);
                ds.z = coef * fmaf(c2, D[t].z, h[t].z);
                ds.w = coef * fmaf(c2, D[t].w, h[t].w);
                float* dp = drow + (size_t)i * d + j;
                unsafeAtomicAdd(dp + 0, ds.x);
                unsafeAtomicAdd(dp + 1, ds.y);
                unsafeAtomicAdd(dp + 2, ds.z);
                unsafeAtomicAdd(dp + 3, ds.w);
                dzacc[t].x -= ds.x; dzacc[t].y -= ds.y; dzacc[t].z -= ds.z; dzacc[t].w -= ds.w;
            }
        }
        dgam += -2.f * a * hd * e * (1.f - gamma * r2);
        if (dalphas && lane == 0) unsafeAtomicAdd(dalphas + (size_t)k * n2 + i, -2.f * gamma * e * hd);
    }
    if (dloggamma && loggamma && lane == 0 && dgam != 0.f) unsafeAtomicAdd(dloggamma + k, gamma * dgam);
    (void)red;
    if (dz) {
#pragma unroll
        for (int t = 0; t < NT; ++t) zred[wave][lane + 64 * t] = dzacc[t];
        __syncthreads();
        for (int slot = threadIdx.x; slot < 64 * NT; slot += RBF_THREADS) {
            const int l = slot & 63, t = slot >> 6;
            const int j = 4 * l + 256 * t;
            if (j < d) {
                float4 s = zred[0][slot];
#pragma unroll
                for (int ww = 1; ww < RBF_WAVES; ++ww) {
                    const float4 o = zred[ww][slot];
                    s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
                }
                float* dp = dz + (size_t)b * d + j;
                unsafeAtomicAdd(dp + 0, s.x);
                unsafeAtomicAdd(dp + 1, s.y);
                unsafeAtomicAdd(dp + 2, s.z);
                unsafeAtomicAdd(dp + 3, s.w);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// traversal: one workgroup per (path k, code c, direction); T sequential steps with the support
// set resident in LDS (when it fits) so HBM is touched once per walk instead of once per step.
template <int NT, bool IN_LDS>
__global__ __launch_bounds__(RBF_THREADS) void rbf_traverse_kernel(
    const float* __restrict__ table, const float* __restrict__ alphas,
    const float* __restrict__ loggamma, float gamma_c, const float* __restrict__ codes, float eps,
    int T, float* __restrict__ path, float* __restrict__ shift, int n_codes, int K, int n2, int d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // layout: red[4][256*NT] floats | (IN_LDS) support set [n2*d]
    float4* red = reinterpret_cast<float4*>(smem);
    float* sset = smem + RBF_WAVES * 256 * NT;
    __shared__ float nred[RBF_WAVES];

    const int k = blockIdx.x, c = blockIdx.y, dir = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float sign = dir == 0 ? 1.f : -1.f;
    const float gamma = loggamma ? expf(loggamma[k]) : gamma_c;
    const float* srow_g = table + (size_t)k * n2 * d;
    const float* arow = alphas + (size_t)k * n2;
    const float* srow = srow_g;
    if (IN_LDS) {
        const int n4 = n2 * d / 4;
        for (int q = threadIdx.x; q < n4; q += RBF_THREADS)
            reinterpret_cast<float4*>(sset)[q] = reinterpret_cast<const float4*>(srow_g)[q];
        srow = sset;
        __syncthreads();
    }
    const int L = 2 * T + 1;
    float* prow = path + ((size_t)c * K + k) * L * d;
    float* hrow = shift + ((size_t)c * K + k) * L * d;

    float4 zr[NT];
    load_vec<NT>(zr, codes + (size_t)c * d, lane, d);
    if (dir == 0 && wave == 0) {  // centre entry: start code, zero shift
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = 4 * lane + 256 * t;
            if (j < d) {
                *reinterpret_cast<float4*>(prow + (size_t)T * d + j) = zr[t];
                *reinterpret_cast<float4*>(hrow + (size_t)T * d + j) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    for (int step = 1; step <= T; ++step) {
        float4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        rbf_accumulate<NT, false>(acc, zr, srow, arow, gamma, 0, n2, d, wave, lane, nullptr);
        __syncthreads();  // previous step's readers of `red` are done
#pragma unroll
        for (int t = 0; t < NT; ++t) red[wave * 64 * NT + lane + 64 * t] = acc[t];
        __syncthreads();
        float nn = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float4 s = red[lane + 64 * t];
#pragma unroll
            for (int ww = 1; ww < RBF_WAVES; ++ww) {
                const float4 o = red[ww * 64 * NT + lane + 64 * t];
                s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
            }
            acc[t] = s;  // every wave now holds the full field
            nn += dot4(s, s);
        }
        const float f = sign * eps / sqrtf(wave_sum(nn));
        const int pos = dir == 0 ? T + step : T - step;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float4 sh = make_float4(acc[t].x * f, acc[t].y * f, acc[t].z * f, acc[t].w * f);
            zr[t].x += sh.x; zr[t].y += sh.y; zr[t].z += sh.z; zr[t].w += sh.w;
            const int j = 4 * lane + 256 * t;
            if (wave == 0 && j < d) {
                *reinterpret_cast<float4*>(prow + (size_t)pos * d + j) = zr[t];
                *reinterpret_cast<float4*>(hrow + (size_t)pos * d + j) = sh;
            }
        }
    }
    (void)nred; (void)n_codes;
}

int rbf_nt(int d) { return d <= 256 ? 1 : d <= 512 ? 2 : d <= 1024 ? 4 : 8; }

}  // namespace

extern "C" {

int64_t wgs_rbf_ws_floats(int B, int n2, int d) {
    const int S = rbf_splits(B, n2);
    const int64_t head = (int64_t)B * d + ((B + 3) & ~3) + (((int64_t)B * n2 + 3) & ~(int64_t)3);
    return head + (int64_t)B * S * d;
}

int wgs_rbf_fwd(const float* table, const float* alphas, const float* loggamma, float gamma,
                const int64_t* idx, const float* z, const float* scale, float* out, float* ws, int B,
                int K, int n2, int d, wgs_stream_t stream) {
    WGS_CHECK_ARG(table && alphas && idx && z && out && ws, "wgs_rbf_fwd: null pointer");
    WGS_CHECK_ARG(B > 0 && K > 0 && n2 > 0, "wgs_rbf_fwd: bad sizes B=%d K=%d n2=%d", B, K, n2);
    WGS_CHECK_ARG(d > 0 && d % 4 == 0 && d <= 2048, "wgs_rbf_fwd: d=%d must be a multiple of 4 and <= 2048", d);
    hipStream_t st = (hipStream_t)stream;
    if (!wgs_flags().rbf_split && d <= 1024) {       // one launch: a 16-wave workgroup per sample (d <= 1024: NT <= 4, 64 KB of LDS)
        dim3 g1(B), b1(64 * RBF1_WAVES);
        switch (rbf_nt(d)) {
            case 1: WGS_LAUNCH(rbf_fwd_one<1>, g1, b1, 0, st, table, alphas, loggamma, gamma, idx, z, scale, out, ws, B, n2, d); break;
            case 2: WGS_LAUNCH(rbf_fwd_one<2>, g1, b1, 0, st, table, alphas, loggamma, gamma, idx, z, scale, out, ws, B, n2, d); break;
            default: WGS_LAUNCH(rbf_fwd_one<4>, g1, b1, 0, st, table, alphas, loggamma, gamma, idx, z, scale, out, ws, B, n2, d); break;
        }
        WGS_CHECK_LAUNCH("rbf_fwd_one");
        return WGS_OK;
    }
    const int S = rbf_splits(B, n2);
    dim3 grid(S, B), block(RBF_THREADS);
    switch (rbf_nt(d)) {
        case 1: WGS_LAUNCH(rbf_fwd_partial<1>, grid, block, 0, st, table, alphas, loggamma, gamma, idx, z, ws, B, n2, d, S); break;
        case 2: WGS_LAUNCH(rbf_fwd_partial<2>, grid, block, 0, st, table, alphas, loggamma, gamma, idx, z, ws, B, n2, d, S); break;
        case 4: WGS_LAUNCH(rbf_fwd_partial<4>, grid, block, 0, st, table, alphas, loggamma, gamma, idx, z, ws, B, n2, d, S); break;
        default: WGS_LAUNCH(rbf_fwd_partial<8>, grid, block, 0, st, table, alphas, loggamma, gamma, idx, z, ws, B, n2, d, S); break;
    }
    WGS_CHECK_LAUNCH("rbf_fwd_partial");
    WGS_LAUNCH(rbf_fwd_finish, dim3(B), block, 0, st, ws, scale, out, B, n2, d, S);
    WGS_CHECK_LAUNCH("rbf_fwd_finish");
    return WGS_OK;
}

int wgs_rbf_bwd(const float* table, const float* alphas, const float* loggamma, float gamma,
                const int64_t* idx, const float* z, const float* scale, const float* gout,
                const float* ws, float* dtable, float* dloggamma, float* dalphas, float* dz, int B,
                int K, int n2, int d, wgs_stream_t stream) {
    WGS_CHECK_ARG(table && alphas && idx && z && gout && ws && dtable, "wgs_rbf_bwd: null pointer");
    WGS_CHECK_ARG(B > 0 && K > 0 && n2 > 0, "wgs_rbf_bwd: bad sizes B=%d K=%d n2=%d", B, K, n2);
    WGS_CHECK_ARG(d > 0 && d % 4 == 0 && d <= 2048, "wgs_rbf_bwd: d=%d must be a multiple of 4 and <= 2048", d);
    hipStream_t st = (hipStream_t)stream;
    const int S = rbf_splits(B, n2);
    dim3 grid(S, B), block(RBF_THREADS);
#define WGS_RBF_BWD(NT)                                                                             \
    WGS_LAUNCH(rbf_bwd_kernel<NT>, grid, block, 0, st, table, alphas, loggamma, gamma, idx, \
                       z, scale, gout, ws, dtable, dloggamma, dalphas, dz, B, n2, d, S)
    switch (rbf_nt(d)) {
        case 1: WGS_RBF_BWD(1); break;
        case 2: WGS_RBF_BWD(2); break;
        case 4: WGS_RBF_BWD(4); break;
        default: WGS_RBF_BWD(8); break;
    }
#undef WGS_RBF_BWD
    WGS_CHECK_LAUNCH("rbf_bwd_kernel");
    return WGS_OK;
}

int wgs_rbf_traverse(const float* table, const float* alphas, const float* loggamma, float gamma,
                     const float* codes, float eps, int T, float* path, float* shift, int n_codes,
                     int K, int n2, int d, wgs_stream_t stream) {
    WGS_CHECK_ARG(table && alphas && codes && path && shift, "wgs_rbf_traverse: null pointer");
    WGS_CHECK_ARG(n_codes > 0 && K > 0 && n2 > 0 && T >= 0, "wgs_rbf_traverse: bad sizes");
    WGS_CHECK_ARG(d > 0 && d % 4 == 0 && d <= 2048, "wgs_rbf_traverse: d=%d must be a multiple of 4 and <= 2048", d);
    hipStream_t st = (hipStream_t)stream;
    const int NT = rbf_nt(d);
    const size_t red_bytes = (size_t)RBF_WAVES * 256 * NT * sizeof(float);
    const size_t set_bytes = (size_t)n2 * d * sizeof(float);
    const bool in_lds = red_bytes + set_bytes + 64 <= 160 * 1024;
    const size_t smem = red_bytes + (in_lds ? set_bytes : 0);
    dim3 grid(K, n_codes, 2), block(RBF_THREADS);
#define WGS_RBF_TRV(NT_, L_)                                                                         \
    do {                                                                                             \
        auto kfn = rbf_traverse_kernel<NT_, L_>;                                                     \
        if (smem > 48 * 1024)                                                                        \
            (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        WGS_LAUNCH(kfn, grid, block, smem, st, table, alphas, loggamma, gamma, codes, eps, T, \
                           path, shift, n_codes, K, n2, d);                                          \
    } while (0)
    if (in_lds) {
        switch (NT) { case 1: WGS_RBF_TRV(1, true); break; case 2: WGS_RBF_TRV(2, true); break;
                      case 4: WGS_RBF_TRV(4, true); break; default: WGS_RBF_TRV(8, true); break; }
    } else {
        switch (NT) { case 1: WGS_RBF_TRV(1, false); break; case 2: WGS_RBF_TRV(2, false); break;
                      case 4: WGS_RBF_TRV(4, false); break; default: WGS_RBF_TRV(8, false); break; }
    }
#undef WGS_RBF_TRV
    WGS_CHECK_LAUNCH("rbf_traverse_kernel");
    return WGS_OK;
}

}  // extern "C"
