// 16-bit-operand 3x3 stride-1 'same' convolution for FEW channels (Cin, Cout <= 64) on LARGE maps: the 64- / 32- / 16-channel layers
// at 256^2 .. 1024^2 of StyleGAN2-1024 (models/StyleGAN2/model.py:297-307) and ProgGAN (models/ProgGAN/model.py:65-95), forward and
// input-gradient launches alike.
//
// Why its own kernel: with K = 9 * Cin = 144 .. 576 and N = Cout <= 64 a launch has ~70 FLOP per byte of HBM traffic — at the 16-bit
// MFMA rate it is an HBM-bound streaming problem (2.1 GB per launch at 32 -> 32 @1024^2, B = 8: 0.36 ms at 6 TB/s), not a GEMM.  The
// GEMM-tiled kernels ran it as a quarter of a million 128 x 32 tiles whose K loop is 5 - 9 barrier-separated chunk steps, each
// staging one tap's activation rows again: latency-bound at 3 - 4x the HBM floor.  Here:
//
//   * a workgroup (4 waves) owns an 8 x 32 block of output pixels of ONE sample and ALL output channels;
//   * per 32-channel chunk (the whole K for Cin <= 32) it stages the 10 x 34-pixel input patch (block + halo) ONCE — loaded as
//     16-byte pieces, multiplied by the style vector and the power-of-two operand scale, converted to the scheme's 16-bit planes;
//     one barrier; then 9 x (chunk / 16) k-steps of MFMAs straight through, the taps reading the patch with shifted row addresses
//     (as conv_igemm_patch.hip does);
//   * the weights never pass through LDS: a pre-pass (halo_wfrag_kernel, ~150 KB, a few microseconds) re-orders the launch's 16-bit
//     weight planes into the main kernel's B-operand FRAGMENT order in the caller's workspace, so that a fragment is one coalesced
//     1-KB wave load (the same for every workgroup: L2-resident), double-buffered in registers one tap ahead of its MFMAs.  (First
//     version: all nine taps' weight rows of a chunk in LDS next to the patch — 92 KB for 64 columns in the two-plane schemes, ONE
//     4-wave workgroup per CU, every load latency exposed: 0.75 ms split-bf16 / 0.61 ms fp16 x2 at 64 -> 64 @512^2, B = 8.)
//   * LDS holds only the patch (16 - 54 KB): two or three workgroups per CU overlap each other's load latency, MFMAs and epilogue
//     stores (the shared branch-free epilogue of conv_epilogue.h: demodulation, noise, bias, activation, |y| maximum).
//
// HBM traffic = the input once (halo pixels come from L2: neighbouring tiles run back to back on one XCD) + the output once.
#include "wgs_common.h"
#include "conv_args.h"
#include "conv_epilogue.h"
#include "conv_scheme.h"

typedef wgsconv::epi_f32x16 f32x16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

using wgsconv::ConvArgs;

constexpr int TH = 8, TW = 32, BM = TH * TW;             // output block of a workgroup: 256 GEMM rows
constexpr int OOB = (int)0x80000000;

// WIN = side of the tap window: 3 (the 3 x 3 'same' convs: 9 taps at dy, dx in [-1, 1]) or 4 (<= 16 taps at dy, dx in [dy_min, dy_min + 3]:
// a 7 x 7 stride-2 conv over a space-to-depth input, the ResNet stem of the Reconstructor and its input gradient, reconstructor.py)
struct HaloGeom {
    int tiles_x, tiles_per_img, dy_min, dx_min;
    int tapoff[16];       // patch pixel offset of tap t: (dy - dy_min) * PW + (dx - dx_min)
};

template <int SCH, int KC, int CO, int WIN>
struct HaloCfg {
    static constexpr int NT = WIN * WIN;                   // taps (a 4 x 4 window may list fewer: the launcher pads with zero weights)
    static constexpr int PH = TH + WIN - 1, PW = TW + WIN - 1, NPIX = PH * PW;     // input patch: 340 / 385 pixels
    typedef wgsconv::Scheme<SCH> SC;
    static constexpr int NA = SC::NA, NB = SC::NB;
    static constexpr int PROW = KC * 2 + 16;               // bytes per patch pixel / weight row of a chunk (padded: conflict-free b128 reads)
    static constexpr int P_BYTES = NPIX * PROW;            // one patch plane
    static constexpr int SMEM = NA * P_BYTES;
    static constexpr int KS = KC / 16, TN = CO / 32;
    static constexpr int FRAGS_PER_TAP = KS * TN * NB;     // 1-KB B fragments of one (chunk, tap)
    // three workgroups per CU (170 VGPRs) unless the double-buffered weight fragments (2 x FRAGS_PER_TAP x 4 registers) need more
    static constexpr int WGS_PER_CU = (SMEM <= 52 * 1024 && FRAGS_PER_TAP <= 4) ? 3 : 2;
    static constexpr long wfrag_bytes(int nchunks) { return (long)nchunks * NT * FRAGS_PER_TAP * 1024; }
};

// B-operand fragments of the launch in issue order: frag[((c * 9 + t) * KS + ks) * TN + j][plane][lane] = the 8 consecutive k
// (chunk c, k-step ks, half lane / 32) of weight row co = j * 32 + lane % 32 of tap t; rows past Cout are zero.
template <int SCH, int KC, int CO, int WIN>
__global__ __launch_bounds__(256) void halo_wfrag_kernel(const ConvArgs p, unsigned short* __restrict__ dst, int total) {
    typedef HaloCfg<SCH, KC, CO, WIN> CF;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int lane = e & 63;
    int f = e >> 6;
    const int pl = f % CF::NB; f /= CF::NB;
    const int j = f % CF::TN; f /= CF::TN;
    const int ks = f % CF::KS; f /= CF::KS;
    const int t = f % CF::NT, c = f / CF::NT;
    const int co = j * 32 + (lane & 31), ci = c * KC + ks * 16 + (lane >> 5) * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (co < p.Co && t < p.ntaps) {
        const size_t off = (size_t)co * p.w_row_stride + (size_t)p.wt[t] * p.w_tap_stride + ci;
        if (p.w_hi) v = *reinterpret_cast<const uint4*>((pl ? p.w_lo : p.w_hi) + off);
        else {
            // no pre-split planes (trained weights: the Reconstructor's convs): split the fp32 weights here, the same roundings
            const float4 a = *reinterpret_cast<const float4*>(p.w + off), b = *reinterpret_cast<const float4*>(p.w + off + 4);
            const f32x4 fa = {a.x, a.y, a.z, a.w}, fb = {b.x, b.y, b.z, b.w};
            uint2 ha, la, hb, lb;
            wgsconv::Scheme<SCH>::cvt4(fa, ha, la);
            wgsconv::Scheme<SCH>::cvt4(fb, hb, lb);
            v = pl ? make_uint4(la.x, la.y, lb.x, lb.y) : make_uint4(ha.x, ha.y, hb.x, hb.y);
        }
    }
    *reinterpret_cast<uint4*>(dst + (size_t)e * 8) = v;
}

// RGB: ToRGB in the epilogue (wgs_conv_desc.rgb_out; the workgroup holds all Cout = CO channels of its pixels) — its own instantiation
template <int SCH, int KC, int CO, int WIN, bool RGB = false>
__global__ __launch_bounds__(256, (HaloCfg<SCH, KC, CO, WIN>::WGS_PER_CU)) void halo3x3_kernel(const ConvArgs p, const HaloGeom g, const unsigned short* __restrict__ wfrag, int wfrag_bytes) {
    typedef HaloCfg<SCH, KC, CO, WIN> CF;
    constexpr int NT = CF::NT, PW = CF::PW, NPIX = CF::NPIX;
    typedef wgsconv::Scheme<SCH> SC;
    typedef typename SC::frag frag;
    constexpr int NA = SC::NA, NB = SC::NB, PROW = CF::PROW, P_BYTES = CF::P_BYTES, KS = CF::KS;
    constexpr int TM = 2, TN = CO / 32, WM = 64, WN = CO;   // a wave: two image rows of the block x all columns
    constexpr int EP = KC / 4;                               // float4 pieces per patch pixel and chunk
    constexpr int NPL = (NPIX * EP + 255) / 256;             // patch loads per thread and chunk
    static_assert(CF::SMEM >= 4 * BM * 4, "the epilogue's row arrays reuse the staging buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* patch = smem_b;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid;
    {   // XCD-aware order: every XCD gets a contiguous range of tiles (neighbouring tiles share their halo rows in that XCD's L2)
        const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int qn = nb >> 3, rn = nb & 7;
        bid = xcd * qn + min(xcd, rn) + slot;
    }
    const int b = bid / g.tiles_per_img;
    const int trem = bid - b * g.tiles_per_img;
    const int ty0 = (trem / g.tiles_x) * TH, tx0 = (trem % g.tiles_x) * TW;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwf = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wfrag), 0, wfrag_bytes, 0x00020000);

    // ---- staging maps: patch element e = tid + j * 256 -> pixel e / EP, float4 piece q = e % EP
    const int q = tid % EP;
    int p_goff[NPL], p_loff[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int pp = (tid + j * 256) / EP;
        const int pr = pp / PW, pc = pp - pr * PW;
        const int iy = ty0 + g.dy_min + pr, ix = tx0 + g.dx_min + pc;
        // p.ups: the conv runs on the nearest-neighbour up-sampled grid of x (ProgGAN's Upsample + conv, models/ProgGAN/model.py:53-62):
        // grid pixel (iy, ix) reads stored pixel (iy >> ups, ix >> ups); the four copies of a stored pixel come out of L1 / L2
        const bool v = pp < NPIX && (unsigned)iy < (unsigned)(p.Hi << p.ups) && (unsigned)ix < (unsigned)(p.Wi << p.ups);
        p_goff[j] = v ? (((b * p.Hi + (iy >> p.ups)) * p.Wi + (ix >> p.ups)) * p.Ci + q * 4) * 4 : OOB;
        p_loff[j] = pp < NPIX ? pp * PROW + q * 8 : -1;
    }
    float op_mult = 1.f, op_inv = 1.f;
    if (SCH != 0) wgsconv::operand_scale(p.a_amax, p.a_amax2, p.a_bound, op_mult, op_inv);

    u32x4 pr_[NPL];
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
    const int cpt = p.Ci / KC;
    auto load_chunk = [&](int c) {
        const int xb = c * (KC * 4);
#pragma unroll
        for (int j = 0; j < NPL; ++j) pr_[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)((unsigned)p_goff[j] + (unsigned)xb), 0, 0);
        if (p.a_scale) sc = *reinterpret_cast<const float4*>(p.a_scale + (size_t)b * p.a_ld + c * KC + q * 4);
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            float4 v = make_float4(__uint_as_float(pr_[j].x), __uint_as_float(pr_[j].y), __uint_as_float(pr_[j].z), __uint_as_float(pr_[j].w));
            v.x = __fmul_rn(v.x, sc.x); v.y = __fmul_rn(v.y, sc.y); v.z = __fmul_rn(v.z, sc.z); v.w = __fmul_rn(v.w, sc.w);
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));   // keep the rounded product (the roundings of the other 16-bit kernels)
            if (p.pn_eps > 0.f) {
                // PixelNorm of the pixel's Ci = KC channels (its EP consecutive lanes hold them): the arithmetic of pixelnorm_fwd_vec_kernel
                // (stylegan2_ops.hip) — per-lane fma chain, xor tree over the lanes, rsqrt(mean + eps), one rounded product per value
                float s2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, v.w * v.w)));
#pragma unroll
                for (int off = EP / 2; off > 0; off >>= 1) s2 += __shfl_xor(s2, off, 64);
                const float fn = rsqrtf(s2 * (1.f / KC) + p.pn_eps);
                v.x = __fmul_rn(v.x, fn); v.y = __fmul_rn(v.y, fn); v.z = __fmul_rn(v.z, fn); v.w = __fmul_rn(v.w, fn);
                asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            }
            if (SCH != 0) { v.x *= op_mult; v.y *= op_mult; v.z *= op_mult; v.w *= op_mult; }
            const f32x4 f = {v.x, v.y, v.z, v.w};
            uint2 h, l;
            SC::cvt4(f, h, l);
            if (p_loff[j] >= 0) {
                *reinterpret_cast<uint2*>(patch + p_loff[j]) = h;
                if (NA == 2) *reinterpret_cast<uint2*>(patch + P_BYTES + p_loff[j]) = l;
            }
        }
    };
    // B fragments of (chunk c, tap t): FRAGS_PER_TAP coalesced 1-KB wave loads from the fragment-ordered weights
    frag bq[2][KS][TN][NB];
    auto load_b = [&](int set, int c, int t) {
        const int base = (c * NT + t) * CF::FRAGS_PER_TAP * 1024 + lane * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < NB; ++pl) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rwf, base + ((ks * TN + j) * NB + pl) * 1024, 0, 0);
                    bq[set][ks][j][pl] = __builtin_bit_cast(frag, v);
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
    // fragment rows: GEMM row r = ty * 32 + tx; wave wm owns image rows 2 wm, 2 wm + 1 of the block
    int pa0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) pa0[i] = ((2 * wm + i) * PW + l31) * PROW + lh * 16;

    load_chunk(0);
    load_b(0, 0, 0);
    // (at most two chunks — Cin = 64 — and the loop is unrolled: the register set of a tap's weight fragments, (NT c + t) & 1, is then
    // a compile-time index)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (c >= cpt) break;
        if (c > 0) __syncthreads();                  // the previous chunk's fragment reads are done
        store_chunk();
        __syncthreads();
        if (c + 1 < cpt) load_chunk(c + 1);          // in flight behind the MFMAs
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // the next tap's weight fragments (the next chunk's first tap after the last one) are requested before this tap's MFMAs
            if (t + 1 < NT) load_b((NT * c + t + 1) & 1, c, t + 1);
            else if (c + 1 < cpt) load_b((NT * c + t + 1) & 1, c + 1, 0);
            const int to = g.tapoff[t] * PROW;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                frag af[TM][NA];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pl = 0; pl < NA; ++pl) af[i][pl] = *reinterpret_cast<const frag*>(patch + pl * P_BYTES + pa0[i] + to + ks * 32);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = SC::mma(af[i], bq[(NT * c + t) & 1][ks][j], acc[i][j]);
            }
        }
    }
    __syncthreads();

    // ---- epilogue (contract of conv_igemm.hip; rows are the 8 x 32 block's pixels of sample b)
    int* r_pix = reinterpret_cast<int*>(smem_b);
    int* r_b = r_pix + BM;
    float* r_nz = reinterpret_cast<float*>(r_b + BM);
    int* r_add = reinterpret_cast<int*>(r_nz + BM);
    {
        const int ty = tid >> 5, tx = tid & 31;
        const int oy = ty0 + ty, ox = tx0 + tx;
        const int hw = oy * p.Wo + ox;
        r_pix[tid] = b * p.Ho * p.Wo + hw;
        r_b[tid] = b;
        r_nz[tid] = (p.noise && p.noise_w) ? p.noise_w[0] * p.noise[hw] : 0.f;
        r_add[tid] = (b * (p.Ho >> p.add_ups) + (oy >> p.add_ups)) * (p.Wo >> p.add_ups) + (ox >> p.add_ups);
    }
    __syncthreads();
    if constexpr (RGB) {
        static_assert(CF::SMEM >= (4 * BM + 2 * BM * 4) * 4, "the ToRGB partial sums live behind the row arrays");
        wgsconv::conv_epilogue_rgb<BM, TM, TN, WM, WN, 1>(p, acc, smem_b, wm, 0, l31, lh, tid, op_inv);
    } else
        wgsconv::conv_epilogue_apply<BM, TM, TN, WM, WN>(p, acc, smem_b, 0, wm, 0, l31, lh, op_inv);
}

template <int SCH, int KC, int CO, int WIN>
int launch_halo_k(const ConvArgs& a, const HaloGeom& g, int nblocks, hipStream_t st, bool dry) {
    typedef HaloCfg<SCH, KC, CO, WIN> CF;
    const long wfb = CF::wfrag_bytes(a.Ci / KC);
    if (!a.ws || a.ws_bytes < wfb) return 1;            // needs the caller's workspace for the fragment-ordered weights (<= 150 KB)
    if (a.pn_eps > 0.f && a.Ci != KC) return 1;         // a PixelNorm operand: the whole channel vector of a pixel in one chunk
    if (a.rgb_out && (a.Co != CO || WIN != 3)) return 1;    // ToRGB in the epilogue: every column of the tile is an output channel
    if (dry) return 0;
    unsigned short* wf = reinterpret_cast<unsigned short*>(a.ws);
    const int total = (int)(wfb / 16);
    WGS_LAUNCH((halo_wfrag_kernel<SCH, KC, CO, WIN>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, wf, total);
    if constexpr (WIN == 3) {
        if (a.rgb_out) {
            auto kr = halo3x3_kernel<SCH, KC, CO, WIN, true>;
            wgs_note_kernel("halo3x3_kernel<%d, %d, %d, %d, true>", SCH, KC, CO, WIN);
            (void)hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
            WGS_LAUNCH(kr, dim3((unsigned)nblocks), dim3(256), CF::SMEM, st, a, g, (const unsigned short*)wf, (int)wfb);
            return 0;
        }
    }
    auto k = halo3x3_kernel<SCH, KC, CO, WIN>;
    wgs_note_kernel("halo3x3_kernel<%d, %d, %d, %d>", SCH, KC, CO, WIN);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
    WGS_LAUNCH(k, dim3((unsigned)nblocks), dim3(256), CF::SMEM, st, a, g, (const unsigned short*)wf, (int)wfb);
    return 0;
}

template <int SCH>
int launch_halo_s(const ConvArgs& a, const HaloGeom& g, int nblocks, int win, hipStream_t st, bool dry) {
    const bool k16 = a.Ci % 32 != 0;
    if (win == 4) {       // the stem's space-to-depth form: 32 -> 64 (forward) and 64 -> 32 (input gradient) channels
        if (k16) return 1;
        return a.Co > 32 ? launch_halo_k<SCH, 32, 64, 4>(a, g, nblocks, st, dry) : launch_halo_k<SCH, 32, 32, 4>(a, g, nblocks, st, dry);
    }
    if (a.Co > 32) return k16 ? launch_halo_k<SCH, 16, 64, 3>(a, g, nblocks, st, dry) : launch_halo_k<SCH, 32, 64, 3>(a, g, nblocks, st, dry);
    return k16 ? launch_halo_k<SCH, 16, 32, 3>(a, g, nblocks, st, dry) : launch_halo_k<SCH, 32, 32, 3>(a, g, nblocks, st, dry);
}

}  // namespace

namespace wgsconv {

// 0 = launch taken.  Needs pre-split weight planes, the extents (set_extents) and the tap tables.
int launch_halo16(const ConvArgs& a, hipStream_t st, bool dry) {
    if (wgs_flags().no_halo || (a.w_hi && !a.w_lo && a.sch != 1) || a.a_hi || a.ups > 1 || a.isy != 1 || a.isx != 1 || a.osy != 1 || a.osx != 1 ||
        a.oy0 || a.ox0 || a.ntaps < 9 || a.ntaps > 16)
        return 1;
    if ((a.Ci != 16 && a.Ci != 32 && a.Ci != 64) || a.Co % 4 || a.Co > 64 || a.Hg != (a.Hi << a.ups) || a.Wg != (a.Wi << a.ups) || a.Ho != a.Hg || a.Wo != a.Wg || a.Hg % TH || a.Wg % TW) return 1;
    if ((long)a.B * (a.Hg / TH) * (a.Wg / TW) < wgs_flags().halo_min_tiles) return 1;      // small maps: the GEMM-tiled kernels (split K, 128-row tiles)
    HaloGeom g;
    int dy0 = 127, dy1 = -127, dx0 = 127, dx1 = -127;
    for (int t = 0; t < a.ntaps; ++t) {
        dy0 = a.dy[t] < dy0 ? a.dy[t] : dy0; dy1 = a.dy[t] > dy1 ? a.dy[t] : dy1;
        dx0 = a.dx[t] < dx0 ? a.dx[t] : dx0; dx1 = a.dx[t] > dx1 ? a.dx[t] : dx1;
    }
    // 3 x 3 'same' convs (all nine taps around the pixel), or <= 16 taps inside a 4 x 4 window that reaches at most 2 pixels out
    const bool w3 = a.ntaps == 9 && dy0 == -1 && dy1 == 1 && dx0 == -1 && dx1 == 1;
    const bool w4 = !w3 && dy1 - dy0 <= 3 && dx1 - dx0 <= 3 && dy0 >= -3 && dy1 <= 3 && dx0 >= -3 && dx1 <= 3;
    if (!w3 && !w4) return 1;
    const int win = w3 ? 3 : 4, pw = TW + win - 1;
    unsigned seen = 0;
    for (int t = 0; t < 16; ++t) g.tapoff[t] = 0;
    for (int t = 0; t < a.ntaps; ++t) {
        const unsigned bit = 1u << ((a.dy[t] - dy0) * 4 + a.dx[t] - dx0);
        if (seen & bit) return 1;                      // a tap listed twice
        seen |= bit;
        g.tapoff[t] = (a.dy[t] - dy0) * pw + (a.dx[t] - dx0);
    }
    g.dy_min = dy0; g.dx_min = dx0;
    g.tiles_x = a.Wg / TW; g.tiles_per_img = (a.Hg / TH) * g.tiles_x;
    const int nblocks = a.B * g.tiles_per_img;
    ConvArgs b = a;
    b.w_bytes = a.w_bytes / 2;          // extents of the 16-bit weight planes
    if (a.sch == 0) return launch_halo_s<0>(b, g, nblocks, win, st, dry);
    if (a.sch == 1) return launch_halo_s<1>(b, g, nblocks, win, st, dry);
    return launch_halo_s<2>(b, g, nblocks, win, st, dry);
}

}  // namespace wgsconv
