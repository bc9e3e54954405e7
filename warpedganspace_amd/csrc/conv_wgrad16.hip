// Weight gradient of a conv on the bf16 matrix cores with fp32-class accuracy (split-bf16 x3, conv_scheme.h Scheme<0>):
//     dW[co][t][ci] (+)= sum over output pixels m of dy[m][co] * x[pix_t(m)][ci]
// GEMM view: M' = Cout, N' = Cin, K' = pixels.  Both operands are k-MAJOR in memory (a pixel row is contiguous over
// channels) while a 16-bit MFMA fragment wants 8 consecutive k per lane, so the tile is transposed IN REGISTERS while it is
// staged: a thread loads the same 4 channels of 8 consecutive pixels (8 coalesced float4 loads: the lanes of a wave cover
// 512 contiguous bytes of each pixel row), splits the 32 values into bf16 hi / lo and writes, per channel, the 8 k-values
// as one 16-byte LDS store into the [channel][k] image (rows of 32 k = 64 B padded to 80 B: conflict-free ds_read_b128 of
// the fragments, exactly the image of conv_igemm_bf16.hip).  Main loop = that kernel's: 3 x v_mfma_f32_32x32x16_bf16 per
// product block, fp32 accumulate.  grid = (co tiles x ci tiles, taps, K splits); K splits combine with atomicAdd into the
// caller-zeroed gradient, like the exact-fp32 kernel (conv_igemm.hip) this one replaces for the Reconstructor's 3x3 / 1x1
// convs: 63 TFLOP/s -> see DESIGN.md.
#include "wgs_common.h"
#include "conv_scheme.h"
#include "../../include/wgs.h"

namespace {

typedef wgsconv::sch_f32x16 f32x16;
typedef wgsconv::sch_f32x4 f32x4;
typedef wgsconv::Scheme<0> SC;
typedef SC::frag frag;

struct Wgrad16Args {
    const float* x;   // [B,Hi,Wi,Ci]
    const float* dy;  // [B,Ho,Wo,Co]
    float* dw;        // dw[co*row_stride + wt[t]*tap_stride + ci]
    int B, Hi, Wi, Ci, Ho, Wo, Co, isy, isx, ntaps, M, ksplit;
    long w_tap_stride, w_row_stride;
    signed char dy_[64], dx_[64];
    short wt[64];
    int x_s2d;        // FLAT form only: x is the space-to-depth tensor [B, Hi/2, Wi/2, 4 * Ci] (wgs_pack_pair_s2d; Ci == 8)
    int oct_row;      // Wo % 8 == 0: the 8 pixels of a staging unit share an image row (one pixel -> (b, oy, ox) split per unit and chunk)
};

constexpr int BK = 32;          // pixels per chunk
constexpr int ROWB = 80;        // bytes per LDS row: 32 bf16 + 16 B pad

// BM x BN output tile (BM, BN in {64, 128}), 4 waves as 2 x 2.
// FLAT (few input channels: the Reconstructor's stem, Ci = 8, 49 taps): the GEMM columns are the flattened (tap, ci) pairs, so one
// launch reads dy once for all taps (grid.y = 1) — the form of igemm_wgrad_kernel<.., true>, on the bf16 matrix cores.  A staging
// unit's four columns lie inside one tap (Ci % 4 == 0); the host guarantees Wo % 8 == 0 (a pixel octet never straddles image rows).
template <int BM, int BN, bool FLAT = false>
__global__ __launch_bounds__(256, 2) void igemm_wgrad16_kernel(const Wgrad16Args p) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int STAGE = 2 * A_BYTES + 2 * B_BYTES;            // A_hi | A_lo | B_hi | B_lo
    // staging units: (channel quad, pixel octet); A has BM/4 * 4 units, B has BN/4 * 4
    constexpr int UA = BM, UB = BN, UT = UA + UB;               // units per chunk (BM/4*4 + BN/4*4)
    constexpr int UPT = (UT + 255) / 256;                       // units per thread (1 for 128+128)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ncol = FLAT ? p.ntaps * p.Ci : p.Ci;
    const int ntn = (ncol + BN - 1) / BN;
    const int co0 = (blockIdx.x / ntn) * BM, ci0 = (blockIdx.x % ntn) * BN;
    const int t = FLAT ? 0 : blockIdx.y;
    const int dyt = p.dy_[t], dxt = p.dx_[t];
    const int nchunks = (p.M + BK - 1) / BK;
    const int per = (nchunks + p.ksplit - 1) / p.ksplit;
    const int c_begin = blockIdx.z * per, c_end = min(nchunks, c_begin + per);
    if (c_begin >= c_end) return;

    // this thread's staging units
    int u_isA[UPT], u_cq[UPT], u_po[UPT];
    int u_dy[UPT], u_dx[UPT], u_ci[UPT];        // FLAT: tap offsets and first channel of a B unit's column quad
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
        const int e = tid + u * 256;
        const bool isA = e < UA;
        const int f = isA ? e : e - UA;
        const int nq = (isA ? BM : BN) / 4;
        u_isA[u] = e < UT ? (isA ? 1 : 0) : -1;
        u_cq[u] = f % nq;
        u_po[u] = f / nq;
        u_dy[u] = dyt; u_dx[u] = dxt; u_ci[u] = ci0 + u_cq[u] * 4;
        if (FLAT && !isA) {
            const int col = ci0 + u_cq[u] * 4;
            const int tc = col < ncol ? col / p.Ci : 0;
            u_dy[u] = p.dy_[tc]; u_dx[u] = p.dx_[tc];
            u_ci[u] = col < ncol ? col - tc * p.Ci : -1;      // -1: column past the end (zeros)
        }
    }
    float4 rg[UPT][8];
    auto load_chunk = [&](int c) {
        const int mbase = c * BK;
        bool o_ok[UPT];
        const float* o_row[UPT];
        int o_ix0[UPT];
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            o_ok[u] = false; o_row[u] = p.x; o_ix0[u] = 0;
            if (p.oct_row && u_isA[u] == 0) {
                const int m0 = mbase + u_po[u] * 8, ci = u_ci[u];
                if (m0 < p.M && ci >= 0 && ci < p.Ci) {
                    const int ox = m0 % p.Wo;
                    const int tt = m0 / p.Wo;
                    const int oy = tt % p.Ho;
                    const int b = tt / p.Ho;
                    const int iy = oy * p.isy + u_dy[u];
                    o_ix0[u] = ox * p.isx + u_dx[u];
                    o_ok[u] = iy >= 0 && iy < p.Hi;
                    if (o_ok[u])
                        o_row[u] = (FLAT && p.x_s2d) ? p.x + (((size_t)(b * (p.Hi >> 1) + (iy >> 1)) * (p.Wi >> 1)) * 4 + (iy & 1) * 2) * p.Ci + ci
                                                     : p.x + ((size_t)(b * p.Hi + iy) * p.Wi) * p.Ci + ci;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            if (u_isA[u] < 0) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = mbase + u_po[u] * 8 + j;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (u_isA[u] == 1) {
                    const int co = co0 + u_cq[u] * 4;
                    if (m < p.M && co < p.Co) v = *reinterpret_cast<const float4*>(p.dy + (size_t)m * p.Co + co);
                } else if (!p.oct_row) {
                    const int ci = u_ci[u];
                    if (m < p.M && ci >= 0 && ci < p.Ci) {
                        const int ox = m % p.Wo;
                        const int tt = m / p.Wo;
                        const int oy = tt % p.Ho;
                        const int b = tt / p.Ho;
                        const int iy = oy * p.isy + u_dy[u], ix = ox * p.isx + u_dx[u];
                        if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) {
                            if (FLAT && p.x_s2d)
                                v = *reinterpret_cast<const float4*>(p.x + (((size_t)(b * (p.Hi >> 1) + (iy >> 1)) * (p.Wi >> 1) + (ix >> 1)) * 4 + ((iy & 1) * 2 + (ix & 1))) * p.Ci + ci);
                            else
                                v = *reinterpret_cast<const float4*>(p.x + ((size_t)(b * p.Hi + iy) * p.Wi + ix) * p.Ci + ci);
                        }
                    }
                } else {
                    // the octet's row was split once (o_ok / o_row / o_ix0 above): a 32-bit division is ~35 vector instructions, and
                    // eight of them per staged float4 made this kernel instruction-bound on its gather
                    const int ix = o_ix0[u] + j * p.isx;
                    if (o_ok[u] && ix >= 0 && ix < p.Wi) {
                        if (FLAT && p.x_s2d)
                            v = *reinterpret_cast<const float4*>(o_row[u] + ((size_t)(ix >> 1) * 4 + (ix & 1)) * p.Ci);
                        else
                            v = *reinterpret_cast<const float4*>(o_row[u] + (size_t)ix * p.Ci);
                    }
                }
                rg[u][j] = v;
            }
        }
    };
    // register transpose + split + 16-byte LDS stores: channel c of the quad gets the 8 k-values {rg[j].c}
    auto store_chunk = [&](int buf) {
        unsigned char* base = smem_b + buf * STAGE;
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            if (u_isA[u] < 0) continue;
            unsigned char* plane = base + (u_isA[u] == 1 ? 0 : 2 * A_BYTES);
            const int pbytes = u_isA[u] == 1 ? A_BYTES : B_BYTES;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                float e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = cc == 0 ? rg[u][j].x : (cc == 1 ? rg[u][j].y : (cc == 2 ? rg[u][j].z : rg[u][j].w));
                uint2 h0, l0, h1, l1;
                SC::cvt4(f32x4{e[0], e[1], e[2], e[3]}, h0, l0);
                SC::cvt4(f32x4{e[4], e[5], e[6], e[7]}, h1, l1);
                const int off = (u_cq[u] * 4 + cc) * ROWB + u_po[u] * 16;
                *reinterpret_cast<uint4*>(plane + off) = make_uint4(h0.x, h0.y, h1.x, h1.y);
                *reinterpret_cast<uint4*>(plane + pbytes + off) = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    load_chunk(c_begin);
    store_chunk(0);
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        if (c + 1 < c_end) load_chunk(c + 1);
        const unsigned char* base = smem_b + cur * STAGE;
        const unsigned char* a_hi = base + (wm * WM + l31) * ROWB + lh * 16;
        const unsigned char* b_hi = base + 2 * A_BYTES + (wn * WN + l31) * ROWB + lh * 16;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            frag bf[TN][2], af[TM][2];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bf[j][0] = *reinterpret_cast<const frag*>(b_hi + j * 32 * ROWB + ks * 32);
                bf[j][1] = *reinterpret_cast<const frag*>(b_hi + B_BYTES + j * 32 * ROWB + ks * 32);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                af[i][0] = *reinterpret_cast<const frag*>(a_hi + i * 32 * ROWB + ks * 32);
                af[i][1] = *reinterpret_cast<const frag*>(a_hi + A_BYTES + i * 32 * ROWB + ks * 32);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = SC::mma(af[i], bf[j], acc[i][j]);
        }
        if (c + 1 < c_end) store_chunk(cur ^ 1);
        __syncthreads();
    }
    float* out = p.dw + (FLAT ? (size_t)0 : (size_t)p.wt[t] * p.w_tap_stride);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ci = ci0 + wn * WN + j * 32 + l31;
        const bool cok = ci < ncol;
        size_t coff = ci;
        if (FLAT && cok) { const int tc = ci / p.Ci; coff = (size_t)p.wt[tc] * p.w_tap_stride + (ci - tc * p.Ci); }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < p.Co && cok) unsafeAtomicAdd(out + (size_t)co * p.w_row_stride + coff, acc[i][j][r]);
            }
        }
    }
}

// ---- three taps of one kernel row per workgroup --------------------------------------------------------------------------
// The per-tap kernel above moves 32 KB of operands per 96 MFMAs (333 B / MFMA) and sits at 62-90 TFLOP/s against the vector
// memory path.  For a stride-1 conv the three taps (ky, kx = 0..2) of a kernel row read the SAME dy chunk and the same input
// row shifted by one pixel: a workgroup that owns all three stages dy once and 10 input pixels per 8-pixel octet (instead of
// 3 x 8), writes three shifted [channel][k] images of the input tile, and runs three accumulator sets: BM x 64 tiles, 26 KB per
// 144 MFMAs at BM = 128 (180 B / MFMA).  grid = (co tiles x ci tiles, kernel rows, K splits).
template <int BM>
__global__ __launch_bounds__(256, 1) void igemm_wgrad16_row_kernel(const Wgrad16Args p) {
    constexpr int BN = 64, NT = 3;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int STAGE = 2 * A_BYTES + NT * 2 * B_BYTES;       // A_hi | A_lo | tap 0: B_hi | B_lo | tap 1 ... | tap 2 ...
    constexpr int UA = BM, UB = BN;                             // staging units (channel quad, pixel octet): one per thread
    static_assert(UA + UB <= 256, "one staging unit per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = p.Ci / BN;
    const int co0 = (blockIdx.x / ntn) * BM, ci0 = (blockIdx.x % ntn) * BN;
    const int t0 = blockIdx.y * NT;                             // taps t0 .. t0+2: same dy, dx = dx0, dx0+1, dx0+2
    const int dyt = p.dy_[t0], dx0 = p.dx_[t0];
    const int nchunks = p.M / BK;                               // the host guarantees Wo % 8 == 0 (octets never straddle rows) and M % 32 == 0
    const int per = (nchunks + p.ksplit - 1) / p.ksplit;
    const int c_begin = blockIdx.z * per, c_end = min(nchunks, c_begin + per);
    if (c_begin >= c_end) return;

    const bool isA = tid < UA, isB = !isA && tid < UA + UB;
    const int f = isA ? tid : tid - UA;
    const int nq = (isA ? BM : BN) / 4;
    const int u_cq = f % nq, u_po = f / nq;                     // channel quad, pixel octet (0..3)
    float4 rg[10];
    auto load_chunk = [&](int c) {
        const int m = c * BK + u_po * 8;                        // first pixel of the octet
        if (isA) {
#pragma unroll
            for (int j = 0; j < 8; ++j) rg[j] = *reinterpret_cast<const float4*>(p.dy + (size_t)(m + j) * p.Co + co0 + u_cq * 4);
        } else if (isB) {
            const int ox = m % p.Wo;
            const int tt = m / p.Wo;
            const int oy = tt % p.Ho;
            const int b = tt / p.Ho;
            const int iy = oy + dyt;
            const bool rowok = iy >= 0 && iy < p.Hi;
            const float* row = p.x + ((size_t)(b * p.Hi + (rowok ? iy : 0)) * p.Wi) * p.Ci + ci0 + u_cq * 4;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int ix = ox + dx0 + j;
                rg[j] = (rowok && ix >= 0 && ix < p.Wi) ? *reinterpret_cast<const float4*>(row + (size_t)ix * p.Ci) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_chunk = [&](int buf) {
        unsigned char* base = smem_b + buf * STAGE;
        if (isA) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                float e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = cc == 0 ? rg[j].x : (cc == 1 ? rg[j].y : (cc == 2 ? rg[j].z : rg[j].w));
                uint2 h0, l0, h1, l1;
                SC::cvt4(f32x4{e[0], e[1], e[2], e[3]}, h0, l0);
                SC::cvt4(f32x4{e[4], e[5], e[6], e[7]}, h1, l1);
                const int off = (u_cq * 4 + cc) * ROWB + u_po * 16;
                *reinterpret_cast<uint4*>(base + off) = make_uint4(h0.x, h0.y, h1.x, h1.y);
                *reinterpret_cast<uint4*>(base + A_BYTES + off) = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
        } else if (isB) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                float e[10];
#pragma unroll
                for (int j = 0; j < 10; ++j) e[j] = cc == 0 ? rg[j].x : (cc == 1 ? rg[j].y : (cc == 2 ? rg[j].z : rg[j].w));
                // split once, then three shifted windows of the 16-bit values
                unsigned short hs[10], ls[10];
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const __bf16 h = (__bf16)e[j];
                    const __bf16 l = (__bf16)(e[j] - (float)h);
                    hs[j] = __builtin_bit_cast(unsigned short, h);
                    ls[j] = __builtin_bit_cast(unsigned short, l);
                }
                const int off = (u_cq * 4 + cc) * ROWB + u_po * 16;
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    unsigned char* pl = base + 2 * A_BYTES + q * 2 * B_BYTES;
                    *reinterpret_cast<uint4*>(pl + off) = make_uint4(hs[q] | (hs[q + 1] << 16), hs[q + 2] | (hs[q + 3] << 16),
                                                                     hs[q + 4] | (hs[q + 5] << 16), hs[q + 6] | (hs[q + 7] << 16));
                    *reinterpret_cast<uint4*>(pl + B_BYTES + off) = make_uint4(ls[q] | (ls[q + 1] << 16), ls[q + 2] | (ls[q + 3] << 16),
                                                                               ls[q + 4] | (ls[q + 5] << 16), ls[q + 6] | (ls[q + 7] << 16));
                }
            }
        }
    };

    f32x16 acc[NT][TM];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][i][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    load_chunk(c_begin);
    store_chunk(0);
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        if (c + 1 < c_end) load_chunk(c + 1);
        const unsigned char* base = smem_b + cur * STAGE;
        const unsigned char* a_hi = base + (wm * WM + l31) * ROWB + lh * 16;
        const unsigned char* b_hi = base + 2 * A_BYTES + (wn * WN + l31) * ROWB + lh * 16;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            frag af[TM][2];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                af[i][0] = *reinterpret_cast<const frag*>(a_hi + i * 32 * ROWB + ks * 32);
                af[i][1] = *reinterpret_cast<const frag*>(a_hi + A_BYTES + i * 32 * ROWB + ks * 32);
            }
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                frag bf[2];
                bf[0] = *reinterpret_cast<const frag*>(b_hi + q * 2 * B_BYTES + ks * 32);
                bf[1] = *reinterpret_cast<const frag*>(b_hi + q * 2 * B_BYTES + B_BYTES + ks * 32);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[q][i] = SC::mma(af[i], bf, acc[q][i]);
            }
        }
        if (c + 1 < c_end) store_chunk(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        float* out = p.dw + (size_t)p.wt[t0 + q] * p.w_tap_stride;
        const int ci = ci0 + wn * WN + l31;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                unsafeAtomicAdd(out + (size_t)co * p.w_row_stride + ci, acc[q][i][r]);
            }
    }
}

template <int BM>
void launch16_row(const Wgrad16Args& a, dim3 grid, hipStream_t st) {
    const size_t sm = (size_t)2 * (2 * BM + 3 * 2 * 64) * ROWB;
    auto k = igemm_wgrad16_row_kernel<BM>;
    wgs_note_kernel("igemm_wgrad16_row_kernel<%d>", BM);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    WGS_LAUNCH(k, grid, dim3(256), sm, st, a);
}

template <int BM, int BN, bool FLAT = false>
void launch16(const Wgrad16Args& a, dim3 grid, hipStream_t st) {
    const size_t sm = (size_t)2 * (2 * BM + 2 * BN) * ROWB;
    auto k = igemm_wgrad16_kernel<BM, BN, FLAT>;
    wgs_note_kernel(FLAT ? "igemm_wgrad16_kernel<%d, %d, true>" : "igemm_wgrad16_kernel<%d, %d>", BM, BN);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    WGS_LAUNCH(k, grid, dim3(256), sm, st, a);
}

}  // namespace

// precision-1 form of wgs_conv_wgrad (called from conv_igemm.hip); returns 0 when it took the launch, 1 when the shape is
// left to the exact kernel (few input channels / flattened taps, channel counts that are not multiples of 64)
int wgs_conv_wgrad16(const wgs_wgrad_desc* d, hipStream_t st) {
    const bool flat = d->Ci <= 32 && d->Ci % 4 == 0 && d->ntaps * d->Ci >= 64 && d->Co % 64 == 0 && d->Wo % 8 == 0 && d->ntaps <= 64;
    if (!flat && (d->Ci % 64 != 0 || d->Co % 64 != 0 || d->ntaps > 64 || d->x_s2d)) return 1;
    Wgrad16Args a;
    a.x = d->x; a.dy = d->dy; a.dw = d->dw;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co;
    a.isy = d->isy; a.isx = d->isx; a.ntaps = d->ntaps; a.M = d->B * d->Ho * d->Wo;
    a.w_tap_stride = d->w_tap_stride; a.w_row_stride = d->w_row_stride;
    a.x_s2d = d->x_s2d;
    a.oct_row = d->Wo % 8 == 0 ? 1 : 0;
    for (int t = 0; t < d->ntaps; ++t) { a.dy_[t] = d->dy_t[t]; a.dx_[t] = d->dx_t[t]; a.wt[t] = d->wt[t]; }
    if (flat) {
        // few input channels (the stem: 49 taps x 8 channels = 392 columns): one pass over dy for all taps; 64 x 128 tiles, K = pixels
        const int ncol = d->ntaps * d->Ci;
        const int bm = d->Co >= 128 ? 128 : 64;
        const int tiles = (d->Co / bm) * ((ncol + 127) / 128);
        const int nchunks = (a.M + BK - 1) / BK;
        int ks = d->ksplit;
        if (ks <= 0) { ks = (1024 + tiles - 1) / tiles; if (ks > nchunks / 8) ks = nchunks / 8; if (ks < 1) ks = 1; }
        a.ksplit = ks;
        dim3 grid((unsigned)tiles, 1, (unsigned)ks);
        if (bm == 128) launch16<128, 128, true>(a, grid, st); else launch16<64, 128, true>(a, grid, st);
        return 0;
    }
    // stride-1 convs whose taps come as kernel rows (dy equal, dx consecutive within each group of three): the row kernel
    bool rows = d->isx == 1 && d->isy == 1 && d->ntaps % 3 == 0 && d->Wo % 8 == 0 && a.M % BK == 0 && d->Hi == d->Ho && d->Wi == d->Wo &&
                !wgs_flags().wgrad_per_tap;
    // measured (tools/bench_wgrad.py, B = 32, with the split cap below): 64 / 128 / 256 / 512 channels 105 / 102 / 109 / 113 TFLOP/s
    // against 51 / 86 / 90 / 86 for the per-tap tiles (and 65-70 for the exact fp32 kernel)
    for (int t = 0; rows && t < d->ntaps; t += 3)
        rows = d->dy_t[t + 1] == d->dy_t[t] && d->dy_t[t + 2] == d->dy_t[t] && d->dx_t[t + 1] == d->dx_t[t] + 1 && d->dx_t[t + 2] == d->dx_t[t] + 2;
    if (rows) {
        const int bm = d->Co >= 128 ? 128 : 64;
        const int tiles = (d->Co / bm) * (d->Ci / 64), nrow = d->ntaps / 3;
        const int nchunks = a.M / BK;
        int ks = d->ksplit;
        if (ks <= 0) {
            ks = (768 + tiles * nrow - 1) / (tiles * nrow);       // ~3 workgroups per CU (one resident at a time: 100 KB of LDS)
            // every split adds Co*Ci*taps fp32 atomics: ~19 M of them cost ~100 us whatever the layer (K-split sweep in DESIGN.md),
            // so wide layers take fewer, longer splits as long as the chip stays covered (512 channels @8x8: 8 -> 2 splits, 107 -> 86 us)
            const int cap = (int)(5000000L / ((long)d->Co * d->Ci * d->ntaps));
            if (cap >= 1 && ks > cap && tiles * nrow * cap >= 190) ks = cap;
            if (ks > nchunks / 4) ks = nchunks / 4;
            if (ks < 1) ks = 1;
        }
        a.ksplit = ks;
        dim3 grid((unsigned)tiles, (unsigned)nrow, (unsigned)ks);
        if (bm == 128) launch16_row<128>(a, grid, st); else launch16_row<64>(a, grid, st);
        return 0;
    }
    const int BM = d->Co >= 128 ? 128 : 64, BN = d->Ci >= 128 ? 128 : 64;
    const int tiles = ((d->Co + BM - 1) / BM) * ((d->Ci + BN - 1) / BN);
    const int nchunks = (a.M + BK - 1) / BK;
    int ks = d->ksplit;
    if (ks <= 0) {
        ks = (1024 + tiles * d->ntaps - 1) / (tiles * d->ntaps);
        const int cap = wgs_flags().wgrad_per_tap ? 0 : (int)(5000000L / ((long)d->Co * d->Ci * d->ntaps));      // as in the row form above
        if (cap >= 1 && ks > cap && tiles * d->ntaps * cap >= 190) ks = cap;
        if (ks > nchunks / 4) ks = nchunks / 4;
        if (ks < 1) ks = 1;
    }
    a.ksplit = ks;
    dim3 grid((unsigned)tiles, (unsigned)d->ntaps, (unsigned)ks);
    if (BM == 128 && BN == 128) launch16<128, 128>(a, grid, st);
    else if (BM == 128) launch16<128, 64>(a, grid, st);
    else if (BN == 128) launch16<64, 128>(a, grid, st);
    else launch16<64, 64>(a, grid, st);
    return 0;
}
