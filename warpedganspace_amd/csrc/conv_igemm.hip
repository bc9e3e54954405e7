// Implicit-GEMM convolution on the gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// One kernel family covers every dense contraction of the hot path:
//   * plain / strided conv forward              (Reconstructor, generator convs)
//   * stride-2 transposed conv as 4 sub-pixel phase GEMMs (StyleGAN2 up-convs, models/StyleGAN2/
//     model.py:201-212) and dgrad of strided convs (same structure)
//   * dgrad of stride-1 convs (taps mirrored, weights packed [tap][Cin][Cout])
//   * weight gradients (igemm_wgrad below), contracting over pixels
// GEMM view (forward): M = B*Hg*Wg pixels, N = Cout, K = taps*Cin.  Activations are NHWC so a K-chunk
// of one tap is a contiguous BK-float run per pixel row; weights are addressed as
// w[wt[t]*tap_stride + n*row_stride + k] so PyTorch's channels_last [Cout,kh,kw,Cin] storage is used
// in place.  The StyleGAN2 modulation is folded in: A rows are scaled by style[b,ci] while being
// staged (a_scale), accumulators by demod[b,co] in the epilogue (col_scale) — the per-sample
// weight tensor of model.py:190-199 is never materialised.
//
// Tiling: 256 threads = 4 waves; block tile BM x BN (128x128 default), each wave a (BM/WAVES_M) x
// (BN/WAVES_N) sub-tile of 32x32 MFMA tiles (64 accumulator VGPRs at 64x64).  K is consumed in
// BK-float chunks, double-buffered in LDS with a register-staged prefetch of the next chunk issued
// before the MFMAs of the current one (one barrier per chunk).  LDS rows are padded to BK+1 floats:
// the per-lane operand reads (lane -> row, fixed k) are then bank-conflict free.
#include "wgs_common.h"
#include "conv_args.h"
#include "../../include/wgs.h"

using wgsconv::ConvArgs;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool ASCALE>
__global__ __launch_bounds__(256) void igemm_nt_kernel(const ConvArgs p) {
    constexpr int LD = BK + 1;
    constexpr int CPR = BK / 4;       // float4 chunks per tile row
    constexpr int RPP = 256 / CPR;    // tile rows filled per pass of the 256 threads
    constexpr int PA = (BM + RPP - 1) / RPP;
    constexpr int PB = (BN + RPP - 1) / RPP;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "bad wave layout");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][BM*LD]
    float* Bs = smem + 2 * BM * LD;   // [2][BN*LD]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ntn = (p.Co + BN - 1) / BN;
    const int bid = blockIdx.x;
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;

    const int q = tid % CPR, r0 = tid / CPR;
    int a_iy0[PA], a_ix0[PA], a_pix[PA], a_b[PA];
#pragma unroll
    for (int pa = 0; pa < PA; ++pa) {
        const int row = r0 + pa * RPP;
        const int m = m0 + row;
        const bool ok = (row < BM) && (m < p.M);
        const int mm = ok ? m : 0;
        const int gx = mm % p.Wg;
        const int t = mm / p.Wg;
        const int gy = t % p.Hg;
        const int b = t / p.Hg;
        a_iy0[pa] = ok ? gy * p.isy : -100000;  // invalid rows fail every bounds test
        a_ix0[pa] = gx * p.isx;
        a_pix[pa] = b * p.Hi * p.Wi;
        a_b[pa] = b;
    }

    // Staging registers.  Loads are issued UNCONDITIONALLY from clamped (always valid) addresses and the
    // padding / tail predicate is applied when the chunk is written to LDS: no branches and no s_waitcnt
    // between the 8..12 global loads of a chunk, so they are all in flight while the MFMAs of the previous
    // chunk run (the style multiply also waits until store time).
    float4 ra[PA], rb[PB], rs[ASCALE ? PA : 1];
    unsigned amask = 0, bmask = 0;
    const int cpt = p.Ci / BK;  // K-chunks per tap
    // split-K (see wgs_conv_desc.ws): workgroup (tile, blockIdx.y) contracts chunks [kbeg, kend)
    const int nk_all = p.ntaps * cpt;
    const int kper = (nk_all + p.ksplit - 1) / p.ksplit;
    const int kbeg = (int)blockIdx.y * kper;
    const int kend = min(nk_all, kbeg + kper);

    auto load_tile = [&](int kt) {
        const int t = kt / cpt;
        const int ci0 = (kt - t * cpt) * BK + q * 4;
        const int yx = p.tap_yx[t];       // dword tables: scalar loads (see conv_args.h)
        const int dy = (int)(short)(yx & 0xffff), dx = yx >> 16;
        amask = 0;
#pragma unroll
        for (int pa = 0; pa < PA; ++pa) {
            const int iy = a_iy0[pa] + dy, ix = a_ix0[pa] + dx;
            const bool v = (r0 + pa * RPP < BM) && iy >= 0 && iy < (p.Hi << p.ups) && ix >= 0 && ix < (p.Wi << p.ups);
            const size_t off = v ? ((size_t)(a_pix[pa] + (iy >> p.ups) * p.Wi + (ix >> p.ups))) * p.Ci + ci0 : (size_t)ci0;
            ra[pa] = *reinterpret_cast<const float4*>(p.x + off);
            if (ASCALE) rs[pa] = *reinterpret_cast<const float4*>(p.a_scale + (size_t)a_b[pa] * p.a_ld + ci0);
            amask |= (v ? 1u : 0u) << pa;
        }
        const float* wt = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.w) + p.tap_w[t]) + ci0;
        bmask = 0;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const int n = r0 + pb * RPP;
            const bool v = (n < BN) && (n0 + n < p.Co);
            rb[pb] = *reinterpret_cast<const float4*>(wt + (size_t)(v ? n0 + n : 0) * p.w_row_stride);
            bmask |= (v ? 1u : 0u) << pb;
        }
    };
    auto store_tile = [&](int buf) {
        float* a = As + buf * BM * LD;
        float* b = Bs + buf * BN * LD;
#pragma unroll
        for (int pa = 0; pa < PA; ++pa) {
            const int row = r0 + pa * RPP;
            if (row < BM) {
                float4 v = ra[pa];
                if (ASCALE) { v.x *= rs[pa].x; v.y *= rs[pa].y; v.z *= rs[pa].z; v.w *= rs[pa].w; }
                const bool ok = (amask >> pa) & 1u;
                float* d = a + row * LD + q * 4;
                d[0] = ok ? v.x : 0.f; d[1] = ok ? v.y : 0.f; d[2] = ok ? v.z : 0.f; d[3] = ok ? v.w : 0.f;
            }
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const int row = r0 + pb * RPP;
            if (row < BN) {
                const bool ok = (bmask >> pb) & 1u;
                float* d = b + row * LD + q * 4;
                d[0] = ok ? rb[pb].x : 0.f; d[1] = ok ? rb[pb].y : 0.f; d[2] = ok ? rb[pb].z : 0.f; d[3] = ok ? rb[pb].w : 0.f;
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

    load_tile(kbeg);
    store_tile(0);
    __syncthreads();

    const int l31 = lane & 31, lh = lane >> 5;
    for (int kt = kbeg; kt < kend; ++kt) {
        const int cur = (kt - kbeg) & 1;
        if (kt + 1 < kend) load_tile(kt + 1);
        const float* a = As + cur * BM * LD + (wm * WM + l31) * LD + lh;
        const float* b = Bs + cur * BN * LD + (wn * WN + l31) * LD + lh;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = a[i * 32 * LD + 2 * kk];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = b[j * 32 * LD + 2 * kk];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < kend) store_tile(cur ^ 1);
        __syncthreads();
    }

    if (p.ksplit > 1) {
        // raw partial tile -> ws[split][m][n]; wgsconv::launch_splitk_epilogue reduces and finishes
        float* part = p.ws + (size_t)blockIdx.y * p.M * p.Co;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + l31;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < p.M && n < p.Co) part[(size_t)m * p.Co + n] = acc[i][j][r];
                }
        }
        return;
    }
    // ---- epilogue: per-row output addressing (and the noise term) staged once in LDS ------------------
    int* r_pix = reinterpret_cast<int*>(smem);   // output pixel index (b*Ho+oy)*Wo+ox, or -1
    int* r_b = r_pix + BM;                       // sample index
    float* r_nz = reinterpret_cast<float*>(r_b + BM);   // noise_w * noise[oy*Wo+ox]
    int* r_add = reinterpret_cast<int*>(r_nz + BM);     // pixel index into the (possibly lower-resolution) addend
    if (tid < BM) {
        const int m = m0 + tid;
        int pix = -1, bb = 0, ap = 0;
        float nz = 0.f;
        if (m < p.M) {
            const int gx = m % p.Wg;
            const int t = m / p.Wg;
            const int gy = t % p.Hg;
            bb = t / p.Hg;
            const int hw = (gy * p.osy + p.oy0) * p.Wo + gx * p.osx + p.ox0;
            pix = bb * p.Ho * p.Wo + hw;
            if (p.noise && p.noise_w) nz = p.noise_w[0] * p.noise[hw];
            const int oy = gy * p.osy + p.oy0, ox = gx * p.osx + p.ox0;
            ap = (bb * (p.Ho >> p.add_ups) + (oy >> p.add_ups)) * (p.Wo >> p.add_ups) + (ox >> p.add_ups);
        }
        r_pix[tid] = pix; r_b[tid] = bb; r_nz[tid] = nz; r_add[tid] = ap;
    }
    __syncthreads();
    // demodulation factors: a tile usually covers one or two samples -> two registers per column
    const int b_lo = r_b[0];
    const int m_last = min(m0 + BM, p.M) - 1;
    const int b_hi = (m_last / p.Wg) / p.Hg;
    const bool cs_fast = p.col_scale && (b_hi - b_lo <= 1);
    float vmax = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + l31;
        const bool nok = n < p.Co;
        const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
        float cs0 = 1.f, cs1 = 1.f;
        if (cs_fast && nok) {
            cs0 = p.col_scale[(size_t)b_lo * p.col_ld + n];
            cs1 = p.col_scale[(size_t)b_hi * p.col_ld + n];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int pix = r_pix[row];
                if (pix >= 0 && nok) {
                    float v = acc[i][j][r] * p.alpha;
                    if (p.col_scale) v *= cs_fast ? (r_b[row] == b_lo ? cs0 : cs1) : p.col_scale[(size_t)r_b[row] * p.col_ld + n];
                    v += r_nz[row] + bias;
                    if (p.addend) v += p.addend[(size_t)r_add[row] * p.Co + n];
                    v = (p.act == 1) ? tanhf(v) : (v > 0.f ? v : v * p.act_slope) * p.gain;
                    p.y[(size_t)pix * p.Co + n] = v;
                    vmax = fmaxf(vmax, fabsf(v));
                }
            }
        }
    }
    if (p.y_amax) {      // magnitude bound for a consumer's fp16 operand scale (a 16-bit launch that fell back to this kernel)
        vmax = wave_max(vmax);
        if (lane == 0) raise_amax(p.y_amax, vmax);
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient: dW[co][t][ci] (+)= sum over output pixels m of dy[m][co] * x[pix_t(m)][ci].
// GEMM view: M' = Cout, N' = Cin, K' = pixels.  Both operands are "k-major" in memory (a pixel row is
// contiguous over channels), so tiles are staged as [k][i] rows: float4 global loads, float4 LDS
// writes, conflict-free per-lane reads.  grid = (co tiles * ci tiles, taps, K splits); K splits
// combine with atomicAdd into the caller-zeroed gradient.
struct WgradArgs {
    const float* x;   // [B,Hi,Wi,Ci]
    const float* dy;  // [B,Ho,Wo,Co]
    float* dw;        // dw[co*row_stride + wt[t]*tap_stride + ci]
    int B, Hi, Wi, Ci, Ho, Wo, Co, isy, isx, ntaps, M, ksplit;
    long w_tap_stride, w_row_stride;
    signed char dy_[64], dx_[64];
    short wt[64];
    unsigned magic_wo, magic_ho, magic_ci;      // wgs_div_magic of Wo / Ho / Ci, or 0: plain division (an operand would overflow the 32-bit product)
    int x_s2d;        // x is stored space-to-depth: [B, Hi/2, Wi/2, 4 * Ci], channel (py*2 + px)*Ci + c (wgs_pack_pair_s2d; Ci == 8)
};

// n / d with the launch's precomputed reciprocal (three vector instructions) instead of a ~35-instruction integer division: the pixel
// -> (b, oy, ox) split runs once per staged float4, and next to the fp32 MFMA every vector instruction costs matrix-pipe time
__device__ __forceinline__ int wgrad_div(int n, int d, unsigned magic) { return magic ? wgs_div_fast(n, magic) : n / d; }

// FLAT (few input channels, e.g. ResNet conv1 with Ci = 8): the GEMM columns are the flattened (tap, ci) pairs, so one
// launch reads dy once for all taps instead of once per tap; grid.y = 1.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool FLAT = false>
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(const WgradArgs p) {
    constexpr int BK = 32;                 // pixels per chunk
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
    constexpr int NW = WAVES_M * WAVES_N;  // waves that own accumulators (<= 4)
    constexpr int CA = BM / 4, CB = BN / 4;             // float4 per staged row
    constexpr int PA = (BK * CA + 255) / 256, PB = (BK * CB + 255) / 256;
    __shared__ __attribute__((aligned(16))) float As[2][BK * BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ntn = ((FLAT ? p.ntaps * p.Ci : p.Ci) + BN - 1) / BN;
    const int co0 = (blockIdx.x / ntn) * BM, ci0 = (blockIdx.x % ntn) * BN;
    const int t = blockIdx.y;
    const int dyt = p.dy_[t], dxt = p.dx_[t];
    __shared__ int tap_yx[64];
    if (FLAT) {
        if (tid < p.ntaps) tap_yx[tid] = ((int)p.dy_[tid] & 0xffff) | ((int)p.dx_[tid] << 16);
        __syncthreads();
    }
    const int ncol = FLAT ? p.ntaps * p.Ci : p.Ci;
    const int nchunks = (p.M + BK - 1) / BK;
    const int per = (nchunks + p.ksplit - 1) / p.ksplit;
    const int c_begin = blockIdx.z * per, c_end = min(nchunks, c_begin + per);
    if (c_begin >= c_end) return;

    float4 ra[PA], rb[PB];
    auto load_tile = [&](int c) {
        const int mbase = c * BK;
#pragma unroll
        for (int pa = 0; pa < PA; ++pa) {
            const int e = tid + pa * 256;
            const int k = e / CA, c4 = e % CA;
            const int m = mbase + k;
            const int co = co0 + c4 * 4;
            const bool v = (e < BK * CA) && (m < p.M) && (co < p.Co);
            ra[pa] = v ? *reinterpret_cast<const float4*>(p.dy + (size_t)m * p.Co + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const int e = tid + pb * 256;
            const int k = e / CB, c4 = e % CB;
            const int m = mbase + k;
            const int ci = ci0 + c4 * 4;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((e < BK * CB) && (m < p.M) && (ci < ncol)) {
                const int tt = wgrad_div(m, p.Wo, p.magic_wo);
                const int ox = m - tt * p.Wo;
                const int b = wgrad_div(tt, p.Ho, p.magic_ho);
                const int oy = tt - b * p.Ho;
                int dyc = dyt, dxc = dxt, cic = ci;
                if (FLAT) {
                    const int tc = wgrad_div(ci, p.Ci, p.magic_ci);
                    cic = ci - tc * p.Ci;
                    const int yx = tap_yx[tc];
                    dyc = (int)(short)(yx & 0xffff); dxc = yx >> 16;
                }
                const int iy = oy * p.isy + dyc, ix = ox * p.isx + dxc;
                if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) {
                    if (p.x_s2d)
                        val = *reinterpret_cast<const float4*>(p.x + (((size_t)(b * (p.Hi >> 1) + (iy >> 1)) * (p.Wi >> 1) + (ix >> 1)) * 4 + ((iy & 1) * 2 + (ix & 1))) * p.Ci + cic);
                    else
                        val = *reinterpret_cast<const float4*>(p.x + ((size_t)(b * p.Hi + iy) * p.Wi + ix) * p.Ci + cic);
                }
            }
            rb[pb] = val;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int pa = 0; pa < PA; ++pa) {
            const int e = tid + pa * 256;
            if (e < BK * CA) *reinterpret_cast<float4*>(&As[buf][(e / CA) * BM + (e % CA) * 4]) = ra[pa];
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const int e = tid + pb * 256;
            if (e < BK * CB) *reinterpret_cast<float4*>(&Bs[buf][(e / CB) * BN + (e % CB) * 4]) = rb[pb];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile(c_begin);
    store_tile(0);
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        if (c + 1 < c_end) load_tile(c + 1);
        if (wave < NW) {
            const float* a = &As[cur][lh * BM + wm * WM + l31];
            const float* b = &Bs[cur][lh * BN + wn * WN + l31];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                float av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = a[2 * kk * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = b[2 * kk * BN + j * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
        if (c + 1 < c_end) store_tile(cur ^ 1);
        __syncthreads();
    }
    if (wave >= NW) return;
    float* out = p.dw + (FLAT ? (size_t)0 : (size_t)p.wt[t] * p.w_tap_stride);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int ci = ci0 + wn * WN + j * 32 + l31;
        const bool cok = ci < ncol;
        size_t coff = ci;
        if (FLAT && cok) { const int tc = wgrad_div(ci, p.Ci, p.magic_ci); coff = (size_t)p.wt[tc] * p.w_tap_stride + (ci - tc * p.Ci); }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < p.Co && cok) {
                    float* d = out + (size_t)co * p.w_row_stride + coff;
                    unsafeAtomicAdd(d, acc[i][j][r]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Narrow implicit GEMM: Cout <= 8 and Ci == 64 (the image-gradient dgrad of ResNet conv1: 64 -> 6(+2) channels over
// 2 M pixels).  A 128x32 MFMA tile wastes 3/4 of its columns; here a wave owns 16 consecutive GEMM rows: lane =
// (row r = lane / 16 of a 4-row group, 4 input channels k4 = lane % 16), so one float4 load per lane fetches 4 rows x
// 64 channels, the 8 weight rows of a tap (float4 per lane, L1-resident) are reused by the 4 row groups, and the
// 16-lane partial sums are combined by one butterfly transpose-reduction at the end.  Exact fp32 (fmaf chains).
__global__ __launch_bounds__(256) void igemm_narrow_kernel(const ConvArgs p) {
    // the launch's weights (<= 16 taps x 8 columns x 64 channels = 32 KB) live in LDS: every wave re-reads them for each of
    // its row groups, and as 8 vector loads per tap they made the kernel texture-path bound
    __shared__ __attribute__((aligned(16))) float wsm[16 * 8 * 64];
    for (int e = threadIdx.x; e < p.ntaps * 8 * 16; e += 256) {
        const int k4 = (e & 15) * 4, n = (e >> 4) & 7, t = e >> 7;
        const float* wt = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.w) + p.tap_w[t]);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < p.Co) v = *reinterpret_cast<const float4*>(wt + (size_t)n * p.w_row_stride + k4);
        *reinterpret_cast<float4*>(&wsm[(t * 8 + n) * 64 + k4]) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, lr = lane >> 4, k4 = (lane & 15) * 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int m_first = wave * 16;
    if (m_first >= p.M) return;
    int iy0[4], ix0[4], pixo[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int m = m_first + g * 4 + lr;
        const int mm = m < p.M ? m : 0;
        const int gx = mm % p.Wg;
        const int t = mm / p.Wg;
        const int gy = t % p.Hg;
        const int b = t / p.Hg;
        iy0[g] = m < p.M ? gy * p.isy : -100000;      // invalid rows fail every bounds test
        ix0[g] = gx * p.isx;
        pixo[g] = (b * p.Hi * p.Wi + gy * p.isy * p.Wi + gx * p.isx) * 64 + k4;      // element offset of (row, k4), tap (0,0)
    }
    float acc[32];                     // [row group g][column n]
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    for (int t = 0; t < p.ntaps; ++t) {
        const int yx = p.tap_yx[t];
        const int dy = (int)(short)(yx & 0xffff), dx = yx >> 16;
        const int adelta = p.tap_a[t] >> 2;                                           // (dy*Wi + dx) * Ci elements
        // branch-free: every load is issued unconditionally from a clamped address, masked afterwards
        float4 wv[8], av[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int iy = iy0[g] + dy, ix = ix0[g] + dx;
            const bool v = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const float4 a = *reinterpret_cast<const float4*>(p.x + (v ? pixo[g] + adelta : k4));
            av[g] = v ? a : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) wv[n] = *reinterpret_cast<const float4*>(&wsm[(t * 8 + n) * 64 + k4]);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                float s = acc[g * 8 + n];
                s = fmaf(av[g].x, wv[n].x, s); s = fmaf(av[g].y, wv[n].y, s);
                s = fmaf(av[g].z, wv[n].z, s); s = fmaf(av[g].w, wv[n].w, s);
                acc[g * 8 + n] = s;
            }
    }
    // butterfly transpose-reduction over the 16 lanes of a row: 32 values -> 2 per lane
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int half = 16 >> s, off = 8 >> s;
        const bool upper = (lane & off) != 0;
#pragma unroll
        for (int i = 0; i < half; ++i) {
            const float send = upper ? acc[i] : acc[i + half];
            const float keep = upper ? acc[i + half] : acc[i];
            acc[i] = keep + __shfl_xor(send, off, 64);
        }
    }
    // lane (lr, j = lane % 16) now holds value indices 2j and 2j+1 -> (g, n) = (idx / 8, idx % 8) of row g*4 + lr
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = (lane & 15) * 2 + u, g = idx >> 3, n = idx & 7;
        const int m = m_first + g * 4 + lr;
        if (m < p.M && n < p.Co) {
            const int gx = m % p.Wg;
            const int t = m / p.Wg;
            const int gy = t % p.Hg;
            const int b = t / p.Hg;
            const size_t pix = (size_t)b * p.Ho * p.Wo + (size_t)(gy * p.osy + p.oy0) * p.Wo + gx * p.osx + p.ox0;
            p.y[pix * p.Co + n] = acc[u] * p.alpha + (p.bias ? p.bias[n] : 0.f);
        }
    }
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
int launch_nt(const ConvArgs& a, hipStream_t st) {
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.Co + BN - 1) / BN;
    const size_t smem = (size_t)2 * (BM + BN) * (BK + 1) * sizeof(float);
    const size_t smem_epi = (size_t)4 * BM * sizeof(int);
    const size_t sm = smem > smem_epi ? smem : smem_epi;
    dim3 grid((unsigned)(ntm * ntn), (unsigned)a.ksplit), block(256);
    wgs_note_kernel("igemm_nt_kernel<%d, %d, %d, %d, %d, %s>", BM, BN, BK, WAVES_M, WAVES_N, a.a_scale ? "true" : "false");
    if (a.a_scale) {
        auto k = igemm_nt_kernel<BM, BN, BK, WAVES_M, WAVES_N, true>;
        if (sm > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        WGS_LAUNCH(k, grid, block, sm, st, a);
    } else {
        auto k = igemm_nt_kernel<BM, BN, BK, WAVES_M, WAVES_N, false>;
        if (sm > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        WGS_LAUNCH(k, grid, block, sm, st, a);
    }
    return 0;
}

__global__ __launch_bounds__(256) void repack_w_t_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         int Co, int T, int Ci) {
    // dst[t][ci][co] = src[co][t][ci]; 32x32 LDS transpose per (t, ci-tile, co-tile)
    __shared__ float tile[32][33];
    const int t = blockIdx.z, ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        tile[r][tx] = (co < Co && ci < Ci) ? src[((size_t)co * T + t) * Ci + ci] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Ci && co < Co) dst[((size_t)t * Ci + ci) * Co + co] = tile[tx][r];
    }
}

}  // namespace

int wgs_conv_wgrad16(const wgs_wgrad_desc* d, hipStream_t st);      // conv_wgrad16.hip
int wgs_conv_wgrad_direct(const wgs_wgrad_desc* d, hipStream_t st);      // conv_wgrad_direct.hip

extern "C" {

int wgs_split_bf16(const float* x, uint16_t* hi, uint16_t* lo, int64_t n, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && hi && lo && n > 0 && n % 4 == 0, "wgs_split_bf16: bad arguments (n %% 4)");
    wgsconv::split_bf16(x, nullptr, 0, hi, lo, 1, (long)n, 4, (hipStream_t)stream);
    WGS_CHECK_LAUNCH("modsplit_kernel");
    return WGS_OK;
}

int wgs_split_f16(const float* x, uint16_t* hi, uint16_t* lo, int64_t n, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && hi && n > 0 && n % 4 == 0, "wgs_split_f16: bad arguments (n %% 4)");
    wgsconv::split_f16(x, nullptr, 0, hi, lo, 1, (long)n, 4, nullptr, nullptr, 1.f, (hipStream_t)stream);
    WGS_CHECK_LAUNCH("modcvt_f16_kernel");
    return WGS_OK;
}

int wgs_repack_w_t(const float* src, float* dst, int Co, int T, int Ci, wgs_stream_t stream) {
    WGS_CHECK_ARG(src && dst && Co > 0 && T > 0 && Ci > 0, "wgs_repack_w_t: bad arguments");
    dim3 grid((Ci + 31) / 32, (Co + 31) / 32, T);
    WGS_LAUNCH(repack_w_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, Co, T, Ci);
    WGS_CHECK_LAUNCH("repack_w_t_kernel");
    return WGS_OK;
}

static int build_conv_args(const wgs_conv_desc* d, ConvArgs& a) {
    WGS_CHECK_ARG(d && (d->x || d->x_f16) && d->w && (d->y || d->rgb_out), "wgs_conv_igemm: null pointer");
    WGS_CHECK_ARG(!d->rgb_out || (d->rgb_s && d->rgb_w && d->ntaps == 9 && d->rgb_ld >= d->Co && !d->addend && d->act == 0 &&
                                  ((d->x_f16 && ((d->Co == 128 && d->Ci <= 128) || d->Co == 256)) ||
                                   (!d->x_f16 && d->precision >= 1 && (d->Co == 32 || d->Co == 64)))),
                  "wgs_conv_igemm: rgb_out needs 9 taps, rgb_s / rgb_w / rgb_ld >= Co, the leaky-relu epilogue and either an x_f16 operand with Co == 128 "
                  "(Ci <= 128) or 256, or a 16-bit precision with Co == 32 or 64 (the few-channel kernel)");
    WGS_CHECK_ARG(!d->x_f16 || (d->precision == 2 && !d->a_scale && !d->ups && d->w_hi && d->Ci % 32 == 0 && d->Co % 128 == 0 && d->ntaps <= 16),
                  "wgs_conv_igemm: x_f16 needs precision 2, no a_scale / ups, pre-split weights, Ci %% 32 == 0, Co %% 128 == 0, <= 16 taps");
    WGS_CHECK_ARG(d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Hg > 0 && d->Wg > 0 && d->Ho > 0 && d->Wo > 0,
                  "wgs_conv_igemm: bad spatial sizes");
    WGS_CHECK_ARG(d->Ci > 0 && d->Ci % 8 == 0, "wgs_conv_igemm: Ci=%d must be a multiple of 8", d->Ci);
    WGS_CHECK_ARG(d->Co > 0, "wgs_conv_igemm: Co=%d", d->Co);
    WGS_CHECK_ARG(d->ntaps > 0 && d->ntaps <= 64, "wgs_conv_igemm: ntaps=%d out of range (1..64)", d->ntaps);
    WGS_CHECK_ARG((long)d->B * d->Hg * d->Wg < (1L << 31) && (long)d->B * d->Ho * d->Wo < (1L << 31),
                  "wgs_conv_igemm: pixel count overflows int32");
    a.x = d->x; a.w = d->w; a.y = d->y; a.a_scale = d->a_scale; a.col_scale = d->col_scale;
    a.bias = d->bias; a.noise = d->noise; a.noise_w = d->noise_w;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Hg = d->Hg; a.Wg = d->Wg;
    a.isy = d->isy; a.isx = d->isx; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co;
    a.osy = d->osy; a.osx = d->osx; a.oy0 = d->oy0; a.ox0 = d->ox0; a.ntaps = d->ntaps;
    a.M = d->B * d->Hg * d->Wg;
    a.HW = d->Hg * d->Wg; a.Mimg = a.HW;
    a.a_ld = d->a_ld > 0 ? d->a_ld : d->Ci;
    a.col_ld = d->col_ld > 0 ? d->col_ld : d->Co;
    a.w_tap_stride = d->w_tap_stride; a.w_row_stride = d->w_row_stride;
    a.act_slope = d->act_slope; a.gain = d->gain;
    a.alpha = d->alpha == 0.f ? 1.f : d->alpha;
    a.addend = d->addend; a.ups = d->ups; a.add_ups = d->add_ups; a.act = d->act;
    WGS_CHECK_ARG(d->ups >= 0 && d->ups <= 3 && d->add_ups >= 0 && d->add_ups <= 3, "wgs_conv_igemm: bad upsample shift");
    for (int t = 0; t < d->ntaps; ++t) { a.dy[t] = d->dy[t]; a.dx[t] = d->dx[t]; a.wt[t] = d->wt[t]; }
    a.ws = d->ws; a.ws_bytes = d->ws ? d->ws_bytes : 0; a.ksplit = 1;
    a.w_hi = d->w_hi; a.w_lo = d->w_lo; a.a_hi = d->x_f16; a.a_lo = nullptr;
    a.rgb_out = d->rgb_out; a.rgb_s = d->rgb_s; a.rgb_w = d->rgb_w; a.rgb_scale = d->rgb_scale; a.rgb_ld = d->rgb_ld;
    a.col_stats = d->col_stats;
    WGS_CHECK_ARG(!d->col_stats || (!d->rgb_out && !d->x_f16 && d->Co % 4 == 0 && d->y),
                  "wgs_conv_igemm: col_stats needs a stored output, Co %% 4 == 0, and neither rgb_out nor x_f16");
    a.pn_eps = d->a_pixelnorm_eps > 0.f ? d->a_pixelnorm_eps : 0.f;
    WGS_CHECK_ARG(!(a.pn_eps > 0.f) || (d->precision >= 1 && !d->a_scale && !d->x_f16 && (d->Ci == 16 || d->Ci == 32)),
                  "wgs_conv_igemm: a_pixelnorm_eps needs a 16-bit precision, no a_scale / x_f16 and Ci in {16, 32}");
    WGS_CHECK_ARG(d->precision >= 0 && d->precision <= 3, "wgs_conv_igemm: precision=%d (0 fp32, 1 bf16x3, 2 f16, 3 f16x2)", d->precision);
    a.sch = d->precision > 0 ? d->precision - 1 : 4;      // conv_scheme.h: 0..2 the 16-bit schemes, 4 exact fp32
    a.a_amax = d->precision >= 2 ? d->a_amax : nullptr;
    a.a_bound = d->a_bound > 0.f ? d->a_bound : 1.f;
    a.a_amax2 = d->precision >= 2 ? d->a_amax2 : nullptr;
    a.y_amax = d->y_amax;
    wgsconv::fill_tap_tables(a);
    return WGS_OK;
}

int wgs_conv_pixelnorm_supported(const wgs_conv_desc* d) {
    if (!d || d->precision < 1 || d->a_scale || d->x_f16 || (d->Ci != 16 && d->Ci != 32) || !d->x || !d->w || !d->y) return 0;
    wgs_conv_desc c = *d;
    if (!(c.a_pixelnorm_eps > 0.f)) c.a_pixelnorm_eps = 1e-8f;
    // (a launch beyond 2 GiB is issued per sample range, wgs_conv_igemm: the decision is the one for a single sample)
    const long xs = (long)c.Hi * c.Wi * c.Ci * 4, ys = (long)c.Ho * c.Wo * c.Co * 4, lim = 0x7fffffffL;
    if ((long)c.B * xs > lim || (long)c.B * ys > lim) {
        if (xs > lim || ys > lim) return 0;
        long nb = lim / (xs > ys ? xs : ys);
        c.B = (int)(nb < 1 ? 1 : (nb < c.B ? nb : c.B));
    }
    ConvArgs a;
    if (build_conv_args(&c, a) != WGS_OK) return 0;
    int wt_max = 0;
    for (int t = 0; t < a.ntaps; ++t) wt_max = a.wt[t] > wt_max ? a.wt[t] : wt_max;
    if (!wgsconv::set_extents(a, wt_max)) return 0;
    return wgsconv::launch_halo16(a, nullptr, true) == 0 ? 1 : 0;
}

int wgs_conv_rgb_supported(const wgs_conv_desc* d) {
    if (!d || d->x_f16 || d->precision < 1 || (d->Co != 32 && d->Co != 64) || !d->x || !d->w) return 0;
    wgs_conv_desc c = *d;
    if (!c.rgb_out) { c.rgb_out = const_cast<float*>(d->x); c.rgb_s = d->x; c.rgb_w = d->x; c.rgb_ld = d->Co; c.rgb_scale = 1.f; }
    if (!c.y) c.y = const_cast<float*>(d->x);
    const long xs = (long)c.Hi * c.Wi * c.Ci * 4, ys = (long)c.Ho * c.Wo * c.Co * 4, lim = 0x7fffffffL;
    if ((long)c.B * xs > lim || (long)c.B * ys > lim) return 0;       // (a launch split over sample ranges would need rgb_out split too)
    ConvArgs a;
    if (build_conv_args(&c, a) != WGS_OK) return 0;
    int wt_max = 0;
    for (int t = 0; t < a.ntaps; ++t) wt_max = a.wt[t] > wt_max ? a.wt[t] : wt_max;
    if (!wgsconv::set_extents(a, wt_max)) return 0;
    return wgsconv::launch_halo16(a, nullptr, true) == 0 ? 1 : 0;
}

int wgs_conv_igemm(const wgs_conv_desc* d, wgs_stream_t stream) {
    // Tensors beyond 2 GiB (ProgGAN's 512^2 / 1024^2 feature maps at batch 32: 2.1 - 4.3 GB): the fast kernels address their
    // operands through buffer descriptors with 31-bit byte offsets and would decline the launch (-> the plain fp32 kernel with
    // 64-bit addressing, 20 - 40 TFLOP/s).  A conv is independent per sample, so such a launch is issued as consecutive launches
    // over sample ranges that fit.
    if (d && !d->x_f16 && d->B > 1 && d->Ci > 0 && d->Ci % 16 == 0 && d->ntaps <= 16 && d->x && d->y) {
        const long xs = (long)d->Hi * d->Wi * d->Ci * 4, ys = (long)d->Ho * d->Wo * d->Co * 4, lim = 0x7fffffffL;
        if (((long)d->B * xs > lim || (long)d->B * ys > lim) && xs <= lim && ys <= lim) {
            long nb = lim / (xs > ys ? xs : ys);
            if (nb < 1) nb = 1;
            for (int b0 = 0; b0 < d->B; b0 += (int)nb) {
                wgs_conv_desc c = *d;
                c.B = d->B - b0 < nb ? d->B - b0 : (int)nb;
                c.x = d->x + (size_t)b0 * (xs / 4);
                c.y = d->y + (size_t)b0 * (ys / 4);
                if (d->a_scale) c.a_scale = d->a_scale + (size_t)b0 * (d->a_ld > 0 ? d->a_ld : d->Ci);
                if (d->col_scale) c.col_scale = d->col_scale + (size_t)b0 * (d->col_ld > 0 ? d->col_ld : d->Co);
                if (d->addend) c.addend = d->addend + (size_t)b0 * (d->Ho >> d->add_ups) * (d->Wo >> d->add_ups) * d->Co;
                const int rc = wgs_conv_igemm(&c, stream);
                if (rc != WGS_OK) return rc;
            }
            return WGS_OK;
        }
    }
    ConvArgs a;
    const int rc = build_conv_args(d, a);
    if (rc != WGS_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (d->x_f16) {
        WGS_CHECK_ARG(wgsconv::launch_bf16x3(a, st) == 0, "wgs_conv_igemm: x_f16 launch not covered by the LDS-DMA kernel (operand extents)");
        WGS_CHECK_LAUNCH("igemm_dma16_kernel<x_f16>");
        return WGS_OK;
    }
    const bool k32 = (d->Ci % 32 == 0);
    if (d->precision >= 1 && wgsconv::launch_bf16x3(a, st) == 0) {
        WGS_CHECK_LAUNCH("igemm_nt16_kernel");
        return WGS_OK;
    }
    WGS_CHECK_ARG(!(a.pn_eps > 0.f), "wgs_conv_igemm: a_pixelnorm_eps: the launch is not one the few-channel kernel takes (wgs_conv_pixelnorm_supported)");
    WGS_CHECK_ARG(!a.rgb_out, "wgs_conv_igemm: rgb_out without an x_f16 operand: the launch is not one the few-channel kernel takes (wgs_conv_rgb_supported)");
    // exact fp32: the slot-interleaved kernel (conv_igemm_f32.hip) where it covers the shape, else the plain kernel below
    a.sch = 4;
    if (!wgs_flags().f32_old && wgsconv::launch_f32(a, st) == 0) {
        WGS_CHECK_LAUNCH("igemm_nt16_kernel<4>");
        return WGS_OK;
    }
    WGS_CHECK_ARG(!a.col_stats, "wgs_conv_igemm: col_stats: the launch falls to a kernel without the shared epilogue (Ci %% 32 != 0, > 16 taps or > 2 GiB operands)");
    // (64 -> <= 8 channels, the image gradient of ResNet conv1: the template's 128 x 32 tiles measure 1.07 ms against this kernel's
    // 1.47 ms at B = 32, so it is the fallback now — WGS_F32_OLD, or operands the template declines)
    if (d->precision == 0 && d->Co <= 8 && d->Ci == 64 && d->ntaps <= 16 && (long)d->B * d->Hi * d->Wi * 64 < (1L << 31) && !d->ups && !d->a_scale && !d->col_scale && !d->noise && !d->addend &&
        d->act == 0 && d->act_slope == 1.f && d->gain == 1.f) {
        const long waves = ((long)a.M + 15) / 16;
        wgs_note_kernel("igemm_narrow_kernel");
        WGS_LAUNCH(igemm_narrow_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
        WGS_CHECK_LAUNCH("igemm_narrow_kernel");
        return WGS_OK;
    }
    if (d->Co > 64) {
        // too few tiles for the 256 CUs (ResNet layer3/4 at 16x16 / 8x8): split K over the caller's workspace
        const int tiles = ((a.M + 127) / 128) * ((a.Co + 127) / 128);
        const int nk = a.ntaps * (a.Ci / (k32 ? 32 : 8));
        a.ksplit = wgsconv::choose_ksplit(a, tiles, nk);
        if (k32) launch_nt<128, 128, 32, 2, 2>(a, st); else launch_nt<128, 128, 8, 2, 2>(a, st);
        if (a.ksplit > 1) wgsconv::launch_splitk_epilogue(a, st);
    } else if (d->Co > 32) {
        if (k32) launch_nt<128, 64, 32, 2, 2>(a, st); else launch_nt<128, 64, 8, 2, 2>(a, st);
    } else {
        if (k32) launch_nt<128, 32, 32, 4, 1>(a, st); else launch_nt<128, 32, 8, 4, 1>(a, st);
    }
    WGS_CHECK_LAUNCH("igemm_nt_kernel");
    return WGS_OK;
}

// 0 = the launches ran (or, dry, would run) as ONE merged kernel; 1 = they do not merge
static int multi_merged(const wgs_conv_desc* descs, int n, hipStream_t st, bool dry) {
    if (n < 2 || n > 4) return 1;
    ConvArgs as[4];
    bool ok = true;
    for (int i = 0; i < n && ok; ++i) ok = descs[i].precision == descs[0].precision && descs[i].a_amax == descs[0].a_amax && build_conv_args(&descs[i], as[i]) == WGS_OK;
    if (!ok) return 1;
    if (descs[0].precision >= 1) return wgsconv::launch_bf16x3_multi(as, n, st, dry);
    return wgs_flags().f32_old ? 1 : wgsconv::launch_f32_multi(as, n, st, dry);
}

int wgs_conv_igemm_multi_merges(const wgs_conv_desc* descs, int n) {
    if (!descs || n <= 0) return 0;
    for (int i = 0; i < n; ++i) if (descs[i].x_f16) return 0;
    return multi_merged(descs, n, nullptr, true) == 0 ? 1 : 0;
}

int wgs_conv_igemm_multi(const wgs_conv_desc* descs, int n, wgs_stream_t stream) {
    WGS_CHECK_ARG(descs && n > 0, "wgs_conv_igemm_multi: bad arguments");
    for (int i = 0; i < n; ++i) WGS_CHECK_ARG(!descs[i].x_f16, "wgs_conv_igemm_multi: x_f16 operands are single-launch only (wgs_conv_igemm)");
    if (multi_merged(descs, n, (hipStream_t)stream, false) == 0) {
        WGS_CHECK_LAUNCH("igemm_nt16_kernel<multi>");
        return WGS_OK;
    }
    for (int i = 0; i < n; ++i) {
        const int rc = wgs_conv_igemm(&descs[i], stream);
        if (rc != WGS_OK) return rc;
    }
    return WGS_OK;
}

int wgs_conv_wgrad(const wgs_wgrad_desc* d, wgs_stream_t stream) {
    WGS_CHECK_ARG(d && d->x && d->dy && d->dw, "wgs_conv_wgrad: null pointer");
    WGS_CHECK_ARG(d->Ci > 0 && d->Ci % 4 == 0 && d->Co > 0 && d->Co % 4 == 0,
                  "wgs_conv_wgrad: Ci=%d, Co=%d must be multiples of 4", d->Ci, d->Co);
    WGS_CHECK_ARG(d->ntaps > 0 && d->ntaps <= 64, "wgs_conv_wgrad: ntaps=%d out of range", d->ntaps);
    WGS_CHECK_ARG((long)d->B * d->Ho * d->Wo < (1L << 31), "wgs_conv_wgrad: pixel count overflows int32");
    WgradArgs a;
    a.x = d->x; a.dy = d->dy; a.dw = d->dw;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co;
    a.isy = d->isy; a.isx = d->isx; a.ntaps = d->ntaps; a.M = d->B * d->Ho * d->Wo;
    a.w_tap_stride = d->w_tap_stride; a.w_row_stride = d->w_row_stride;
    a.x_s2d = d->x_s2d;
    WGS_CHECK_ARG(!d->x_s2d || (d->Ci == 8 && d->Hi % 2 == 0 && d->Wi % 2 == 0 && (d->precision == 0 || (d->Co % 64 == 0 && d->Wo % 8 == 0))),
                  "wgs_conv_wgrad: x_s2d needs Ci == 8 and even Hi / Wi (the stem's weight gradient); in split-bf16 also Co %% 64 == 0, Wo %% 8 == 0");
    for (int t = 0; t < d->ntaps; ++t) { a.dy_[t] = d->dy_t[t]; a.dx_[t] = d->dx_t[t]; a.wt[t] = d->wt[t]; }
    // reciprocal multiplies are exact while dividend * divisor < 2^32 (wgs_div_magic); divisor 1 needs none
    auto magic_of = [](long n_max, int dv) -> unsigned { return (dv >= 2 && n_max * dv < (1L << 32)) ? wgs_div_magic(dv) : 0u; };
    a.magic_wo = magic_of((long)a.M + 64, d->Wo);
    a.magic_ho = magic_of((long)d->B * d->Ho + 64, d->Ho);
    a.magic_ci = magic_of((long)d->ntaps * d->Ci + 256, d->Ci);
    hipStream_t st = (hipStream_t)stream;
    WGS_CHECK_ARG(!d->ws || (d->ws_bytes > 0 && ((uintptr_t)d->ws & 15) == 0), "wgs_conv_wgrad: ws needs ws_bytes > 0 and 16-byte alignment");
    if ((d->precision == 0 || d->precision == 1) && wgs_conv_wgrad_direct(d, st) == 0) {
        WGS_CHECK_LAUNCH("wgrad_direct_kernel");
        return WGS_OK;
    }
    if (d->precision == 1 && wgs_conv_wgrad16(d, st) == 0) {
        WGS_CHECK_LAUNCH("igemm_wgrad16_kernel");
        return WGS_OK;
    }
    if (d->Ci <= 32 && d->ntaps * d->Ci >= 64) {
        // few input channels: flatten (tap, ci) into the GEMM columns — one pass over dy instead of one per tap
        const int ncol = d->ntaps * d->Ci;
        const int tiles = ((d->Co + 63) / 64) * ((ncol + 127) / 128);
        const int nchunks = (a.M + 31) / 32;
        int ks = d->ksplit;
        if (ks <= 0) { ks = (1024 + tiles - 1) / tiles; if (ks > nchunks / 4) ks = nchunks / 4; if (ks < 1) ks = 1; }
        a.ksplit = ks;
        wgs_note_kernel("igemm_wgrad_kernel<64, 128, 2, 2, true>");
        WGS_LAUNCH((igemm_wgrad_kernel<64, 128, 2, 2, true>), dim3((unsigned)tiles, 1, (unsigned)ks), dim3(256), 0, st, a);
        WGS_CHECK_LAUNCH("igemm_wgrad_kernel<flat>");
        return WGS_OK;
    }
    const int BM = d->Co >= 128 ? 128 : 64;
    const int BN = d->Ci >= 128 ? 128 : (d->Ci >= 64 ? 64 : 32);
    const int tiles = ((d->Co + BM - 1) / BM) * ((d->Ci + BN - 1) / BN);
    const int nchunks = (a.M + 31) / 32;
    int ks = d->ksplit;
    if (ks <= 0) {
        ks = (1024 + tiles * d->ntaps - 1) / (tiles * d->ntaps);
        if (ks > nchunks / 4) ks = nchunks / 4;
        if (ks < 1) ks = 1;
    }
    a.ksplit = ks;
    dim3 grid((unsigned)tiles, (unsigned)d->ntaps, (unsigned)ks), block(256);
    wgs_note_kernel("igemm_wgrad_kernel<%d, %d, %d, %d, false>", BM, BN, (BM == 128 && BN == 32) ? 4 : 2, (BN == 32) ? 1 : 2);
    if (BM == 128 && BN == 128) WGS_LAUNCH((igemm_wgrad_kernel<128, 128, 2, 2>), grid, block, 0, st, a);
    else if (BM == 128 && BN == 64) WGS_LAUNCH((igemm_wgrad_kernel<128, 64, 2, 2>), grid, block, 0, st, a);
    else if (BM == 128) WGS_LAUNCH((igemm_wgrad_kernel<128, 32, 4, 1>), grid, block, 0, st, a);
    else if (BN == 128) WGS_LAUNCH((igemm_wgrad_kernel<64, 128, 2, 2>), grid, block, 0, st, a);
    else if (BN == 64) WGS_LAUNCH((igemm_wgrad_kernel<64, 64, 2, 2>), grid, block, 0, st, a);
    else WGS_LAUNCH((igemm_wgrad_kernel<64, 32, 2, 1>), grid, block, 0, st, a);
    WGS_CHECK_LAUNCH("igemm_wgrad_kernel");
    return WGS_OK;
}

}  // extern "C"
