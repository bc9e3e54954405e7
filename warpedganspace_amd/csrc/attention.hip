// Self-attention core of BigGAN's non-local block (models/BigGAN/layers.py:157-166), fused and batched:
//     beta[b,q,:] = softmax_k( theta[b,q,:] . phi[b,k,:] ),      o[b,q,:] = sum_k beta[b,q,k] * g[b,k,:]
// forward and backward, exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32).  The reference materialises the [B, Pq, Pk]
// score and attention tensors (torch.bmm + F.softmax: 268 MB each at B = 16, 64 x 64 queries, 32 x 32 keys); here they exist
// only as 32 x 32 register blocks:
//   * forward: a wave owns 32 queries.  Pass 1 walks the key blocks and keeps a running row maximum / sum (log-sum-exp `lse`,
//     saved for the backward); pass 2 recomputes each score block, normalises it with the final lse and accumulates o — no
//     rescaling of accumulators, no LDS, no barrier.
//   * backward: P = exp(S - lse) is recomputed per block from the saved lse;  dP = do . g^T,  dS = P * (dP - D) with
//     D[q] = sum_c do[q,c] * o[q,c];  d theta = dS . phi (one workgroup per 32 queries, its four waves split the keys),
//     d phi = dS^T . theta and d g = P^T . do (one workgroup per 32 keys, its four waves split the queries).
// Register-level trick used throughout: the accumulator layout of a 32 x 32 MFMA block (lane = column, 16 registers = rows
// (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) IS the A-operand layout of the transposed block, so a score block computed as
// S^T = phi_blk . theta^T feeds  o = P . g  /  d theta = dS . phi  directly, and one computed as S = theta_blk . phi^T feeds
// d g = P^T . do  /  d phi = dS^T . theta  directly: no transposes through LDS.  Contractions pair their k indices freely
// (lanes 0-31 take the first half of a channel vector, lanes 32-63 the second), so operand fragments are plain 16-byte loads.
#include "wgs_common.h"
#include "../../include/wgs.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ int rowmap(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

template <int N>
__device__ __forceinline__ void ld_vec(float (&d)[N], const float* __restrict__ p) {      // N % 4 == 0, p 16-byte aligned
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        const float4 v = reinterpret_cast<const float4*>(p)[i];
        d[4 * i] = v.x; d[4 * i + 1] = v.y; d[4 * i + 2] = v.z; d[4 * i + 3] = v.w;
    }
}

// channel owned by (lane column l31, column block j) of an N-channel result: N % 32 == 0 -> NJ = N / 32 consecutive channels
// per lane (vector loads / stores); otherwise (24, 48: ch / 8 of the smaller architectures) block j holds channels 32 j + lane
// and the columns past N idle
template <int N> struct ChanMap {
    static constexpr int NJ = (N + 31) / 32;
    static __device__ __forceinline__ int ch(int l31, int j) { return N % 32 == 0 ? NJ * l31 + j : 32 * j + l31; }
    static __device__ __forceinline__ bool ok(int l31, int j) { return N % 32 == 0 || 32 * j + l31 < N; }
};

// S block: acc[r] = sum_kk a[kk] * b[kk] over the paired channel halves (C8H MFMAs)
template <int C8H>
__device__ __forceinline__ f32x16 dot_block(const float (&a)[C8H], const float (&b)[C8H]) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < C8H; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], s, 0, 0, 0);
    return s;
}

// ------------------------------------------------------------------------------------------------------------------------
template <int C8, int C2>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ theta, const float* __restrict__ phi,
                                                       const float* __restrict__ g, float* __restrict__ o, float* __restrict__ lse,
                                                       int Pq, int Pk) {
    constexpr int C8H = C8 / 2, NJ = C2 / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, q0 = ((int)blockIdx.x * 4 + wave) * 32;
    if (q0 >= Pq) return;
    float bt[C8H];                                           // theta row of query q0 + l31, channel half lh: the B operand of S^T
    ld_vec<C8H>(bt, theta + ((size_t)b * Pq + q0 + l31) * C8 + lh * C8H);
    const float* ph = phi + (size_t)b * Pk * C8 + (size_t)l31 * C8 + lh * C8H;       // + key block * 32 * C8
    const float* gb = g + (size_t)b * Pk * C2 + NJ * l31;                             // + key * C2 + j
    const int nkb = Pk / 32;
    // ---- pass 1: log-sum-exp of every score row (this lane: query q0 + l31, the keys rowmap(r, lh) of each block)
    float m = -INFINITY, l = 0.f;
    float a0[C8H], a1[C8H];
    ld_vec<C8H>(a0, ph);
    auto lse_step = [&](const float (&a)[C8H]) {
        const f32x16 s = dot_block<C8H>(a, bt);
        float bm = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) bm = fmaxf(bm, s[r]);
        const float mn = fmaxf(m, bm);
        float acc = l * expf(m - mn);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += expf(s[r] - mn);
        l = acc; m = mn;
    };
    for (int kb = 0; kb < nkb; kb += 2) {
        ld_vec<C8H>(a1, ph + (size_t)(kb + 1) * 32 * C8);
        lse_step(a0);
        if (kb + 2 < nkb) ld_vec<C8H>(a0, ph + (size_t)(kb + 2) * 32 * C8);
        lse_step(a1);
    }
    {   // the two half-waves hold the two halves of each row's keys
        const float m2 = __shfl_xor(m, 32, 64), l2 = __shfl_xor(l, 32, 64);
        const float mm = fmaxf(m, m2);
        l = l * expf(m - mm) + l2 * expf(m2 - mm);
        m = mm;
    }
    const float lse_q = m + logf(l);
    if (lh == 0) lse[(size_t)b * Pq + q0 + l31] = lse_q;
    // ---- pass 2: o = sum over key blocks of P_blk . g_blk with P = exp(S - lse) (already normalised)
    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    ld_vec<C8H>(a0, ph);
    auto o_step = [&](const float (&a)[C8H], int kb) {
        const f32x16 s = dot_block<C8H>(a, bt);
        const float* gk = gb + (size_t)kb * 32 * C2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = expf(s[r] - lse_q);
            const float* gr = gk + (size_t)rowmap(r, lh) * C2;
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(p, gr[j], acc[j], 0, 0, 0);
        }
    };
    for (int kb = 0; kb < nkb; kb += 2) {
        ld_vec<C8H>(a1, ph + (size_t)(kb + 1) * 32 * C8);
        o_step(a0, kb);
        if (kb + 2 < nkb) ld_vec<C8H>(a0, ph + (size_t)(kb + 2) * 32 * C8);
        o_step(a1, kb + 1);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float* orow = o + ((size_t)b * Pq + q0 + rowmap(r, lh)) * C2 + NJ * l31;
#pragma unroll
        for (int j = 0; j < NJ; ++j) orow[j] = acc[j][r];
    }
}

// D[b,q] = sum_c do[b,q,c] * o[b,q,c]: one wave per row
__global__ __launch_bounds__(256) void attn_dsum_kernel(const float* __restrict__ dO, const float* __restrict__ o, float* __restrict__ D,
                                                        long rows, int C2) {
    const int lane = threadIdx.x & 63;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane; c < C2; c += 64) s = fmaf(dO[r * C2 + c], o[r * C2 + c], s);
        s = wave_sum(s);
        if (lane == 0) D[r] = s;
    }
}

// Sum the four waves' partial [32][W] tiles (registers, MFMA accumulator layout over ChanMap<N> column blocks) in LDS, in wave
// order (deterministic), and store the total as rows of `out` (row stride N floats).
template <int N, int NJ>
__device__ __forceinline__ void reduce_store(const f32x16 (&acc)[NJ], float* red, int wave, int l31, int lh, float* __restrict__ out) {
    constexpr int LD = N + 1;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (!ChanMap<N>::ok(l31, j)) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* d = red + rowmap(r, lh) * LD + ChanMap<N>::ch(l31, j);
                    *d = (w == 0 ? 0.f : *d) + acc[j][r];
                }
            }
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < 32 * N; e += 256) out[e] = red[(e / N) * LD + (e % N)];
    __syncthreads();
}

// ---- d theta: workgroup = 32 queries of one sample, waves split the key blocks ---------------------------------------------
template <int C8, int C2>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float* __restrict__ theta, const float* __restrict__ phi,
                                                         const float* __restrict__ g, const float* __restrict__ dO,
                                                         const float* __restrict__ lse, const float* __restrict__ D,
                                                         float* __restrict__ dtheta, int Pq, int Pk) {
    constexpr int C8H = C8 / 2, C2H = C2 / 2, NJ8 = ChanMap<C8>::NJ, LDO = C2 + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dOt = smem;                       // [32][LDO]: the tile's do rows, B operand of dP^T
    float* red = smem + 32 * LDO;            // [32][C8 + 1]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, q0 = (int)blockIdx.x * 32;
    for (int e = threadIdx.x; e < 32 * (C2 / 4); e += 256) {
        const int row = e / (C2 / 4), c4 = e % (C2 / 4);
        *reinterpret_cast<float4*>(dOt + row * LDO + c4 * 4) = *reinterpret_cast<const float4*>(dO + ((size_t)b * Pq + q0 + row) * C2 + c4 * 4);
    }
    float bt[C8H];
    ld_vec<C8H>(bt, theta + ((size_t)b * Pq + q0 + l31) * C8 + lh * C8H);
    const float lse_q = lse[(size_t)b * Pq + q0 + l31], D_q = D[(size_t)b * Pq + q0 + l31];
    __syncthreads();
    f32x16 acc[NJ8];
#pragma unroll
    for (int j = 0; j < NJ8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* bdo = dOt + l31 * LDO + lh * C2H;
    for (int kb = wave; kb < Pk / 32; kb += 4) {
        const size_t key = (size_t)b * Pk + kb * 32;
        float a[C8H];
        ld_vec<C8H>(a, phi + (key + l31) * C8 + lh * C8H);
        const f32x16 s = dot_block<C8H>(a, bt);                      // S^T[key rowmap(r, lh)][q = l31]
        f32x16 dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = 0.f;
        const float* ga = g + (key + l31) * C2 + lh * C2H;          // A operand of dP^T = g_blk . do^T
#pragma unroll 4
        for (int c4 = 0; c4 < C2H / 4; ++c4) {
            const float4 av = *reinterpret_cast<const float4*>(ga + c4 * 4);
            const float4 bv = *reinterpret_cast<const float4*>(bdo + c4 * 4);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ds = expf(s[r] - lse_q) * (dp[r] - D_q);     // dS[q = l31][key rowmap(r, lh)]: A operand of dS . phi
            const float* pr = phi + (key + rowmap(r, lh)) * C8;
#pragma unroll
            for (int j = 0; j < NJ8; ++j) {
                const float bv = ChanMap<C8>::ok(l31, j) ? pr[ChanMap<C8>::ch(l31, j)] : 0.f;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds, bv, acc[j], 0, 0, 0);
            }
        }
    }
    reduce_store<C8, NJ8>(acc, red, wave, l31, lh, dtheta + ((size_t)b * Pq + q0) * C8);
}

// ---- d phi, d g: workgroup = 32 keys of one sample, waves split the query blocks -------------------------------------------
template <int C8, int C2>
__global__ __launch_bounds__(256) void attn_bwd_k_kernel(const float* __restrict__ theta, const float* __restrict__ phi,
                                                         const float* __restrict__ g, const float* __restrict__ dO,
                                                         const float* __restrict__ lse, const float* __restrict__ D,
                                                         float* __restrict__ dphi, float* __restrict__ dg, int Pq, int Pk) {
    constexpr int C8H = C8 / 2, C2H = C2 / 2, NJ = C2 / 32, NJ8 = ChanMap<C8>::NJ, LDG = C2 + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* gt = smem;                        // [32][LDG]: the tile's g rows, B operand of dP
    float* red = smem + 32 * LDG;            // [32][C2 + 1] (re-used for d phi)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, k0 = (int)blockIdx.x * 32;
    for (int e = threadIdx.x; e < 32 * (C2 / 4); e += 256) {
        const int row = e / (C2 / 4), c4 = e % (C2 / 4);
        *reinterpret_cast<float4*>(gt + row * LDG + c4 * 4) = *reinterpret_cast<const float4*>(g + ((size_t)b * Pk + k0 + row) * C2 + c4 * 4);
    }
    float bp[C8H];                                           // phi row of key k0 + l31: the B operand of S
    ld_vec<C8H>(bp, phi + ((size_t)b * Pk + k0 + l31) * C8 + lh * C8H);
    __syncthreads();
    f32x16 ag[NJ], ap[NJ8];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) ag[j][r] = 0.f;
#pragma unroll
        for (int j = 0; j < NJ8; ++j) ap[j][r] = 0.f;
    }
    const float* bg = gt + l31 * LDG + lh * C2H;
    for (int qb = wave; qb < Pq / 32; qb += 4) {
        const size_t q = (size_t)b * Pq + qb * 32;
        float a[C8H];
        ld_vec<C8H>(a, theta + (q + l31) * C8 + lh * C8H);
        const f32x16 s = dot_block<C8H>(a, bp);                      // S[q rowmap(r, lh)][key = l31]
        f32x16 dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = 0.f;
        const float* da = dO + (q + l31) * C2 + lh * C2H;            // A operand of dP = do_blk . g^T
#pragma unroll 4
        for (int c4 = 0; c4 < C2H / 4; ++c4) {
            const float4 av = *reinterpret_cast<const float4*>(da + c4 * 4);
            const float4 bv = *reinterpret_cast<const float4*>(bg + c4 * 4);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t qr = q + rowmap(r, lh);
            const float p = expf(s[r] - lse[qr]);                    // P[q][key = l31]: A operand of P^T . do
            const float ds = p * (dp[r] - D[qr]);                    // dS[q][key]:      A operand of dS^T . theta
            const float* dor = dO + qr * C2 + NJ * l31;
#pragma unroll
            for (int j = 0; j < NJ; ++j) ag[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(p, dor[j], ag[j], 0, 0, 0);
            const float* tr = theta + qr * C8;
#pragma unroll
            for (int j = 0; j < NJ8; ++j) {
                const float bv = ChanMap<C8>::ok(l31, j) ? tr[ChanMap<C8>::ch(l31, j)] : 0.f;
                ap[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds, bv, ap[j], 0, 0, 0);
            }
        }
    }
    reduce_store<C2, NJ>(ag, red, wave, l31, lh, dg + ((size_t)b * Pk + k0) * C2);
    reduce_store<C8, NJ8>(ap, red, wave, l31, lh, dphi + ((size_t)b * Pk + k0) * C8);
}

template <int C8, int C2>
int launch_fwd(const float* theta, const float* phi, const float* g, float* o, float* lse, int B, int Pq, int Pk, hipStream_t st) {
    WGS_LAUNCH((attn_fwd_kernel<C8, C2>), dim3((unsigned)((Pq / 32 + 3) / 4), (unsigned)B), dim3(256), 0, st, theta, phi, g, o, lse, Pq, Pk);
    return 0;
}

template <int C8, int C2>
int launch_bwd(const float* theta, const float* phi, const float* g, const float* o, const float* lse, const float* dO, float* D,
               float* dtheta, float* dphi, float* dg, int B, int Pq, int Pk, hipStream_t st) {
    const long rows = (long)B * Pq;
    WGS_LAUNCH(attn_dsum_kernel, dim3((unsigned)(rows / 4 < 4096 ? (rows + 3) / 4 : 4096)), dim3(256), 0, st, dO, o, D, rows, C2);
    const size_t sm_q = (size_t)(32 * (C2 + 4) + 32 * (C8 + 1)) * 4, sm_k = (size_t)(32 * (C2 + 4) + 32 * (C2 + 1)) * 4;
    auto kq = attn_bwd_q_kernel<C8, C2>;
    auto kk = attn_bwd_k_kernel<C8, C2>;
    (void)hipFuncSetAttribute((const void*)kq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_q);
    (void)hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_k);
    WGS_LAUNCH(kq, dim3((unsigned)(Pq / 32), (unsigned)B), dim3(256), sm_q, st, theta, phi, g, dO, lse, (const float*)D, dtheta, Pq, Pk);
    WGS_LAUNCH(kk, dim3((unsigned)(Pk / 32), (unsigned)B), dim3(256), sm_k, st, theta, phi, g, dO, lse, (const float*)D, dphi, dg, Pq, Pk);
    return 0;
}

bool shape_ok(int B, int Pq, int Pk, int c8, int c2) {
    return B > 0 && Pq > 0 && Pk > 0 && Pq % 128 == 0 && Pk % 64 == 0 && ((c8 == 24 && c2 == 96) || (c8 == 96 && c2 == 384) || (c8 == 48 && c2 == 192));
}

}  // namespace

extern "C" {

int wgs_attn_supported(int B, int Pq, int Pk, int c8, int c2) { return shape_ok(B, Pq, Pk, c8, c2) ? 1 : 0; }

int wgs_attn_fwd(const float* theta, const float* phi, const float* g, float* o, float* lse, int B, int Pq, int Pk, int c8, int c2,
                 wgs_stream_t stream) {
    WGS_CHECK_ARG(theta && phi && g && o && lse, "wgs_attn_fwd: null pointer");
    WGS_CHECK_ARG(shape_ok(B, Pq, Pk, c8, c2), "wgs_attn_fwd: unsupported shape B=%d Pq=%d (%%128) Pk=%d (%%64) c8=%d c2=%d ((24,96), (48,192), (96,384))",
                  B, Pq, Pk, c8, c2);
    hipStream_t st = (hipStream_t)stream;
    if (c8 == 24) launch_fwd<24, 96>(theta, phi, g, o, lse, B, Pq, Pk, st);
    else if (c8 == 48) launch_fwd<48, 192>(theta, phi, g, o, lse, B, Pq, Pk, st);
    else launch_fwd<96, 384>(theta, phi, g, o, lse, B, Pq, Pk, st);
    WGS_CHECK_LAUNCH("attn_fwd_kernel");
    return WGS_OK;
}

int wgs_attn_bwd(const float* theta, const float* phi, const float* g, const float* o, const float* lse, const float* d_o, float* ws,
                 float* dtheta, float* dphi, float* dg, int B, int Pq, int Pk, int c8, int c2, wgs_stream_t stream) {
    WGS_CHECK_ARG(theta && phi && g && o && lse && d_o && ws && dtheta && dphi && dg, "wgs_attn_bwd: null pointer");
    WGS_CHECK_ARG(shape_ok(B, Pq, Pk, c8, c2), "wgs_attn_bwd: unsupported shape B=%d Pq=%d Pk=%d c8=%d c2=%d", B, Pq, Pk, c8, c2);
    hipStream_t st = (hipStream_t)stream;
    if (c8 == 24) launch_bwd<24, 96>(theta, phi, g, o, lse, d_o, ws, dtheta, dphi, dg, B, Pq, Pk, st);
    else if (c8 == 48) launch_bwd<48, 192>(theta, phi, g, o, lse, d_o, ws, dtheta, dphi, dg, B, Pq, Pk, st);
    else launch_bwd<96, 384>(theta, phi, g, o, lse, d_o, ws, dtheta, dphi, dg, B, Pq, Pk, st);
    WGS_CHECK_LAUNCH("attn_bwd kernels");
    return WGS_OK;
}

}  // extern "C"
