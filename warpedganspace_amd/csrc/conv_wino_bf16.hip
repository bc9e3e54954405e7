// 3x3 stride-1 'same' convolution in split-bf16 (fp32-class, ~2^-16 per product) with the HORIZONTAL taps in the Winograd minimal-filtering
// form F(2, 3) and the vertical taps direct: 4 x 3 = 12 products per pair of output pixels and input channel instead of 18, i.e. TWO
// v_mfma_f32_32x32x16_bf16 per direct product where the direct split-bf16 kernels (conv_igemm_patch.hip, scheme 0) issue three.
// The op it stands for: models/StyleGAN2/model.py:187-228 (ModulatedConv2d.forward with the style on the input and the demodulation on
// the output, as the other conv kernels of this library factor it).
//
//   y[oy, 2t + {0,1}] = A^T sum_ky sum_ci (G g[ky]) (.) (B^T d[oy + ky - 1])       d: 4 input pixels 2t-1 .. 2t+2, g[ky]: the 3 taps of kernel row ky
//
//   V_0 = d0 - d2   V_1 = d1 + d2   V_2 = d2 - d1   V_3 = d1 - d3          (B^T d, on the style-modulated input, fp32)
//   U_0 = g0        U_1 = (g0 + g1 + g2) / 2        U_2 = (g0 - g1 + g2) / 2        U_3 = g2      (G g, fp32, once per weight tensor)
//   m_p = sum_ky sum_ci V_p[oy + ky - 1] U_p[ky]     — four independent GEMMs with K = 3 * Ci
//   y[2t] = m_0 + m_1 + m_2        y[2t + 1] = m_1 - m_2 - m_3
//
// V and U are split into bf16 hi + lo AFTER the fp32 transforms; a product block is hi*hi + hi*lo + lo*hi with fp32 accumulation.
//
// Why not F(2x2, 3x3) (1.33 MFMAs per direct product) at the 16-bit MFMA rate: its 16 positions need 16 fp32 accumulators per 2x2 outputs
// and output channel, so a workgroup's register file (512 KB per CU) holds tiles x channels <= 4096 — either <= 32 tiles (every tile group
// re-streams U: 17 GB of L2 -> register traffic per 512 -> 512 @64^2 launch, 0.67 KB per MFMA against the L2's 0.44 KB per MFMA-time) or
// <= 64 channels (the input transform + split, ~6 vector instructions per V element, then costs 8 issue slots per MFMA).  The fp32 kernel
// of conv_wino_f32.hip lives with both because an fp32 MFMA takes 5.3x longer per k.  The one-dimensional form keeps 4 accumulators per
// pixel pair: 128 tiles (256 pixels) x 128 channels per workgroup, an A fragment is shared by the three kernel rows, U re-streams at
// 0.17 KB per MFMA and the transform + split cost 1.2 - 2.3 vector instructions per MFMA.  DESIGN.md section 3.19.
//
// Workgroup = 8 waves = 8 rows x 32 pixels of one sample (128 pairs) x 128 output channels x the 4 positions.  Wave w owns position
// p = w >> 1 for the channel half w & 1: 4 row blocks (2 rows x 16 pairs) x 2 channel blocks = 128 accumulator registers.
//   * V through LDS: per 16-channel chunk the (8 + 2) x 34-pixel input patch becomes 4 positions x 10 rows x 16 pairs of 16 channels,
//     hi and lo planes, rows of 32 B + 16 B padding (consecutive pairs are consecutive rows: a fragment's 32 rows are conflict-free
//     for ds_read_b128's lane groups); double-buffered, one barrier per chunk.  A V row is one wave's staging task (lane = pair x channel
//     quad); the two halo rows are split by position over the eight waves: every wave runs the same instruction stream.
//   * U never passes through LDS: wino16_weight_kernel writes it in the main kernel's B-fragment order, a wave streams the 4 KB of its
//     (position, channel half) per (chunk, kernel row) straight into registers one kernel row ahead.
//   * epilogue: the four positions of a pixel pair live in four waves: one exchange through LDS per channel half ([pair][position][64
//     channels] fp32, 128 KB), then thread = (pair, channel quad): y0 / y1, demodulation, noise, bias, leaky-relu, two 16-byte stores;
//     optionally ToRGB (wgs_conv_desc.rgb_out) from the finished values — at 128 output channels the whole sum, and then y need not be stored at all;
//     at 256 / 512 the 128-channel block's partial sum (a 16-byte slot per pixel and block; the caller's finishing launch adds the slots).
#include <type_traits>
#include "wgs_common.h"
#include "conv_scheme.h"
#include "conv_epilogue.h"
#include "../../include/wgs.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef wgsconv::sch_bf16x8 frag;

#ifndef WGS_W16SCHED
#define WGS_W16SCHED 0   // instruction-mix hints per unit (development A/B, same box, 512 -> 512 @64^2): 0 = 1 MFMA : 5 VALU 1098-1106 us, 1 = 1 : 3 1093, 2 = unit fence only 1094,
                         // 3 = with DS / VMEM groups 1101, 4 = none 1145 — where an instruction sits inside a unit is not the lever
#endif
#ifndef WGS_W16ORD
#define WGS_W16ORD 1     // 1: channel-block-major workgroup order — an XCD's workgroups share one 128-channel slab of U instead of one input patch.  Alone the
                         // launch is +-1 % either way; in the auto step (other streams competing for the memory system) 24.66 / 24.69 / 24.65 -> 24.59 / 24.59 / 24.55 ms,
                         // same box, three alternations (round 6).  0: pixel tile major
#endif
#ifndef WGS_W16ABL
#define WGS_W16ABL 0     // development ablations (tools/build_abl.sh w16abl): 1 no MFMAs, 2 no staging (loads, transform, LDS stores), 3 no U loads, 4 no epilogue exchange / stores,
                         // 5 no barrier in the chunk loop, 6 staging without its global loads, 7 staging loads only (no transform, no LDS stores)
#endif

constexpr int KC = 16;                   // input channels per chunk = the k of one MFMA
constexpr int TR = 8, NTL = 16;          // tile: 8 output rows x 16 pixel pairs (32 pixels)
constexpr int VR = TR + 2;               // V rows of a tile (one halo row above and below)
constexpr int ENT = VR * NTL;            // (row, pair) entries per position
constexpr int RB = 48;                   // bytes per entry: 16 bf16 + 16 B padding
constexpr int POSB = ENT * RB;           // one position of one plane (7 680 B)
constexpr int PLANE = 4 * POSB;          // hi (or lo) plane of a staged chunk
constexpr int STAGE = 2 * PLANE;         // one staged chunk (61 440 B)
constexpr int EPI = 128 * 4 * 64 * 4;    // epilogue exchange of one channel half: [pair][position][64 channels] fp32
constexpr int SMEM = EPI > 2 * STAGE ? EPI : 2 * STAGE;
constexpr int OOB = (int)0x80000000;

struct W16Args {
    const float* x;
    const unsigned short* U;
    float* y;
    const float* a_scale;
    const float* col_scale;
    const float* bias;
    const float* noise;
    const float* noise_w;
    float* y_amax;
    float* rgb_out; const float* rgb_s; const float* rgb_w; float rgb_scale; int rgb_ld;      // ToRGB in the epilogue (wgs_conv_desc.rgb_out [B,H,W,4 * Co / 128]: one partial sum per 128-channel block)
    int B, H, W, Ci, Co, a_ld, col_ld;
    float alpha, act_slope, gain;
};

struct W16Taps { int w_of[9]; };      // weight slab index of spatial tap (ky, kx)

__device__ __forceinline__ f32x4 buf_load4(const __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ frag buf_loadf(const __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return __builtin_bit_cast(frag, v);
}

// U_p[ky] = G g[ky] of every (co, ci) pair, split into bf16 hi + lo and written as the main kernel's B-operand fragments:
// [Co / 128][Ci / 16][ky][wave = 2 * p + channel half][column block of 32][plane][lane = 32 * (k half) + column][8 consecutive k]
__global__ __launch_bounds__(256) void wino16_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ U, int Ci, int Co,
                                                            long w_row_stride, long w_tap_stride, const W16Taps tp) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Ci * Co) return;
    const int ci = (int)(idx % Ci), co = (int)(idx / Ci);
    const int nchunks = Ci / KC;
    const int nb = co >> 7, n = co & 127, nh = n >> 6, jl = (n >> 5) & 1, col = n & 31;
    const int chunk = ci / KC, kk = ci % KC, lane = (kk >> 3) * 32 + col, e = kk & 7;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        float g[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) g[kx] = w[(size_t)co * w_row_stride + (size_t)tp.w_of[ky * 3 + kx] * w_tap_stride + ci];
        const float u[4] = {g[0], 0.5f * (g[0] + g[1] + g[2]), 0.5f * (g[0] - g[1] + g[2]), g[2]};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const __bf16 h = (__bf16)u[p];
            const __bf16 l = (__bf16)(u[p] - (float)h);
            const size_t blk = ((((size_t)(nb * nchunks + chunk) * 3 + ky) * 8 + (2 * p + nh)) * 2 + jl) * 2;
            U[(blk + 0) * 512 + lane * 8 + e] = __builtin_bit_cast(unsigned short, h);
            U[(blk + 1) * 512 + lane * 8 + e] = __builtin_bit_cast(unsigned short, l);
        }
    }
}

template <bool STY, bool RGB>
__global__ __launch_bounds__(512, 1) void wino16_kernel(const W16Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int ntn = p.Co >> 7, tbx = p.W >> 5, tby = p.H >> 3;
    // XCD-aware order (workgroup i runs on XCD i % 8): every XCD gets a contiguous range of the (pixel tile major, channel block minor)
    // list, so the channel blocks of a pixel tile share their input patch in that XCD's L2
    int bid;
    {
        const int nbk = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, qn = nbk >> 3, rn = nbk & 7;
        bid = xcd * qn + min(xcd, rn) + slot;
    }
#if WGS_W16ORD
    // channel block major: an XCD's resident workgroups share ONE 128-channel slab of U (Ci * 6 KB: L2-resident) instead of one input patch
    const int ntile = gridDim.x / ntn;
    const int nb = bid / ntile, tmi = bid - nb * ntile;
#else
    const int tmi = bid / ntn, nb = bid - tmi * ntn;
#endif
    const int b = tmi / (tbx * tby), rr = tmi - b * (tbx * tby), by = rr / tbx, bx = rr - by * tbx;
    const int nchunks = p.Ci / KC;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)b * p.H * p.W * p.Ci), 0, p.H * p.W * p.Ci * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.U), 0, 48 * p.Ci * p.Co, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(STY ? p.a_scale + (size_t)b * p.a_ld : p.x), 0, STY ? p.Ci * 4 : 0, 0x00020000);

    // ---- staging role: wave w stages V row w + 1 (the tile's own row w: all four positions) and ONE position of a halo row — row -1
    // (waves 0-3) or row 8 (waves 4-7), position w & 3: ten rows over eight waves with the same instruction count in every wave and
    // no divergent code.  lane = (pair, channel quad). ----
    const int st_t = lane >> 2, st_q = lane & 3;
    int a_off[4], q_off[2];
    const int hq = wave >> 2, pq = wave & 3;
    // position pq of a row needs two of the four pixels: V_pq = d[ja] + sgn * d[jb]
    const int ja = pq == 0 ? 0 : (pq == 2 ? 2 : 1), jb = pq == 3 ? 3 : (pq == 2 ? 1 : 2);
    const float sgn = pq == 1 ? 1.f : -1.f;
    {
        const int iy = by * TR + wave, iyh = by * TR + (hq ? TR : -1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ix = bx * 32 + 2 * st_t - 1 + j;
            a_off[j] = (unsigned)ix < (unsigned)p.W ? ((iy * p.W + ix) * p.Ci + st_q * 4) * 4 : OOB;
        }
        const int ixa = bx * 32 + 2 * st_t - 1 + ja, ixb = bx * 32 + 2 * st_t - 1 + jb;
        const bool rok = (unsigned)iyh < (unsigned)p.H;
        q_off[0] = rok && (unsigned)ixa < (unsigned)p.W ? ((iyh * p.W + ixa) * p.Ci + st_q * 4) * 4 : OOB;
        q_off[1] = rok && (unsigned)ixb < (unsigned)p.W ? ((iyh * p.W + ixb) * p.Ci + st_q * 4) * 4 : OOB;
    }
    const int v_dst = ((wave + 1) * NTL + st_t) * RB + st_q * 8;
    const int vq_dst = pq * POSB + ((hq ? VR - 1 : 0) * NTL + st_t) * RB + st_q * 8;
    f32x4 ra[4], rq[2], rsv = {1.f, 1.f, 1.f, 1.f};
    auto load_A = [&](int c) {
        const int cb = min(c, nchunks - 1) * (KC * 4);         // past the end: the last chunk again (stored into a dead buffer)
        if (WGS_W16ABL == 2 || WGS_W16ABL == 6) { ra[0] = ra[1] = ra[2] = ra[3] = rq[0] = rq[1] = (f32x4){1.f, 2.f, 3.f, 4.f}; return; }
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[j] = buf_load4(rx, a_off[j], cb);
        rq[0] = buf_load4(rx, q_off[0], cb);
        rq[1] = buf_load4(rx, q_off[1], cb);
        if (STY) rsv = buf_load4(rs, st_q * 16, cb);
    };
    auto split_store = [&](unsigned char* d, const f32x4 v) {
        uint2 h, l;
        wgsconv::Scheme<0>::cvt4(v, h, l);
        *reinterpret_cast<uint2*>(d) = h;
        *reinterpret_cast<uint2*>(d + PLANE) = l;
    };
    // the halo row's position: V = s * d[ja] + sgn * s * d[jb]
    auto stage_halo = [&](int buf) {
        if (WGS_W16ABL == 2) return;
        if (WGS_W16ABL == 7) { asm volatile("" :: "v"(rq[0]), "v"(rq[1])); return; }
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = STY ? (sgn * rsv[k]) * rq[1][k] : sgn * rq[1][k];
            v[k] = STY ? __builtin_fmaf(rsv[k], rq[0][k], t) : rq[0][k] + t;
        }
        split_store(smem + buf * STAGE + vq_dst, v);
    };
    // positions 2 * half, 2 * half + 1 of the own row: transform (style folded in), split, two 8-byte stores per plane
    auto stage_pair = [&](int buf, int half) {
        if (WGS_W16ABL == 2) return;
        if (WGS_W16ABL == 7) { if (half) asm volatile("" :: "v"(ra[0]), "v"(ra[1]), "v"(ra[2]), "v"(ra[3]), "v"(rsv)); return; }
        f32x4 va, vb;
        if (half == 0) {
            if (STY) { ra[1] = rsv * ra[1]; ra[2] = rsv * ra[2]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                va[k] = STY ? __builtin_fmaf(rsv[k], ra[0][k], -ra[2][k]) : ra[0][k] - ra[2][k];
                vb[k] = ra[1][k] + ra[2][k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                va[k] = ra[2][k] - ra[1][k];
                vb[k] = STY ? __builtin_fmaf(-rsv[k], ra[3][k], ra[1][k]) : ra[1][k] - ra[3][k];
            }
        }
        unsigned char* d = smem + buf * STAGE + (2 * half) * POSB + v_dst;
        split_store(d, va);
        split_store(d + POSB, vb);
    };

    // ---- MFMA role: wave = (position pw, channel half nh) ----
    const int pw = wave >> 1, nh = wave & 1;
    const int a_rd = pw * POSB + l31 * RB + lh * 16;          // + buf * STAGE + (2 i + ky) * NTL * RB (+ PLANE: lo)
    const int u_voff = lane * 16;
    frag bfr[3][2][2];          // [ring slot = kernel row][column block][plane]: requested two kernel rows ahead
    auto load_B = [&](int slot, int c, int ky) {
        if (WGS_W16ABL == 3) {
#pragma unroll
            for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) asm volatile("" : "=v"(bfr[slot][jl][pl]));
            return;
        }
        const int cb = (((nb * nchunks + min(c, nchunks - 1)) * 3 + ky) * 8 + wave) * 4096;
#pragma unroll
        for (int jl = 0; jl < 2; ++jl)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) bfr[slot][jl][pl] = buf_loadf(ru, u_voff, cb + (jl * 2 + pl) * 1024);
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Chunk kt multiplies LDS buffer kt & 1 in 12 units (kernel row ky, row block i) of 6 MFMAs; the B fragments of kernel row ky sit in
    // ring slot ky, and at the head of each group of four units the fragments of the group after the next are requested (an L2 round
    // trip under load is about one group long).  The staging work of chunk kt + 1 and the requests of chunk kt + 2 are spread over the units.
    // Measured and left out (round 6, same box, 512 -> 512 @64^2 at 1.07 ms): a two-deep instead of a three-deep B ring (+-0); the next chunk's first B
    // request in front of the patch requests, loads returning in order (-0.5 %); an L2 warm-up load of the patch two chunks ahead (+3 %); the
    // two waves of a SIMD staging in different thirds of the chunk (+2 %); the instruction-mix hints below (+-1 % between any two forms, no
    // hints +4 %).  What the ablations say: without the patch requests -8 %, without the transform + split behind them a further -20 %,
    // without the U requests -21 %, without the chunk barrier -1.5 %, without MFMAs -46 %: the kernel runs at 1.74 GHz under its power cap
    // (GRBM_GUI_ACTIVE / time), and what removing work buys is the energy of that work — re-arranging the same work buys nothing.
    auto mma_chunk = [&](int kt) {
        const int cur = kt & 1, nxt = cur ^ 1;
        const unsigned char* base = smem + cur * STAGE + a_rd;
        frag af[2][2];          // [unit parity][plane]
        af[0][0] = *reinterpret_cast<const frag*>(base);
        af[0][1] = *reinterpret_cast<const frag*>(base + PLANE);
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int ky = u >> 2, i = u & 3;
            if (u + 1 < 12) {
                const int ky1 = (u + 1) >> 2, i1 = (u + 1) & 3;
                af[(u + 1) & 1][0] = *reinterpret_cast<const frag*>(base + (2 * i1 + ky1) * (NTL * RB));
                af[(u + 1) & 1][1] = *reinterpret_cast<const frag*>(base + (2 * i1 + ky1) * (NTL * RB) + PLANE);
            }
            if (i == 0) { if (ky == 0) load_B(2, kt, 2); else load_B(ky - 1, kt + 1, ky - 1); }
            if (WGS_W16ABL != 1) {
                const frag ah = af[u & 1][0], al = af[u & 1][1];
                // the two column blocks interleaved: dependent MFMAs are two issue slots apart
#pragma unroll
                for (int jl = 0; jl < 2; ++jl) acc[i][jl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bfr[ky][jl][0], acc[i][jl], 0, 0, 0);
#pragma unroll
                for (int jl = 0; jl < 2; ++jl) acc[i][jl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bfr[ky][jl][1], acc[i][jl], 0, 0, 0);
#pragma unroll
                for (int jl = 0; jl < 2; ++jl) acc[i][jl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bfr[ky][jl][0], acc[i][jl], 0, 0, 0);
            } else {
                asm volatile("" :: "v"(af[u & 1][0]), "v"(af[u & 1][1]), "v"(bfr[ky][0][0]), "v"(bfr[ky][0][1]), "v"(bfr[ky][1][0]), "v"(bfr[ky][1][1]));
            }
            if (u == 1) stage_halo(nxt);
            if (u == 2) stage_pair(nxt, 0);
            if (u == 3) { stage_pair(nxt, 1); load_A(kt + 2); }
#if WGS_W16SCHED == 0
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#elif WGS_W16SCHED == 1
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#elif WGS_W16SCHED == 2
            __builtin_amdgcn_sched_barrier(0);
#elif WGS_W16SCHED == 3
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };

    load_A(0);
    load_B(0, 0, 0); load_B(1, 0, 1);
    stage_halo(0); stage_pair(0, 0); stage_pair(0, 1);
    load_A(1);
    __syncthreads();
    for (int kt = 0; kt < nchunks; ++kt) {
        mma_chunk(kt);
        if (WGS_W16ABL != 5) __syncthreads();
    }

    // ---- epilogue: per channel half h the four position waves exchange their accumulators through LDS, then every thread finishes
    // (pair T, channel quad cq): y0 = m0 + m1 + m2, y1 = m1 - m2 - m3, demodulation, noise, bias, leaky-relu, two 16-byte stores ----
    if (WGS_W16ABL == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) asm volatile("" :: "v"(acc[i][jl]));
        return;
    }
    const int cq = tid & 15;
    const float nw = p.noise ? p.noise_w[0] : 0.f;
    const float slope = p.act_slope, gain = p.gain;
    float vmax = 0.f;
    float* yb = p.y ? p.y + (size_t)b * p.H * p.W * p.Co : nullptr;
    // ToRGB (models/StyleGAN2/model.py:270-282) in this epilogue: a thread's partial channel sums of its 4 x 2 pixels over both channel halves,
    // then one sum over the 16 lanes (channel quads) of a pixel pair on the vector ALU (DPP row) and one 16-byte pixel store; y itself is
    // stored only when the caller wants it (a pass that keeps nothing never writes the layer's output)
    float racc[RGB ? 4 : 1][2][3];
    if (RGB) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 2; ++j) racc[k][j][0] = racc[k][j][1] = racc[k][j][2] = 0.f;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (nh == h && WGS_W16ABL != 4) {
            // pair T = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh: the lane part and the column block in one of four base registers (row blocks
            // 0 - 1 / 2 - 3 are 64 KB apart), the rest a multiple of 1 KB below 64 KB — an immediate of the store (ds_write2st64_b32 takes two of
            // them; with one base the compiler spent a vector add per store on the offsets beyond a ds_write's 16 bits)
#pragma unroll
            for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int eo = (i < 2 ? 0 : 65536) + (4 * lh) * 1024 + pw * 256 + (jl * 32 + l31) * 4;
                    asm volatile("" : "+v"(eo));        // (opaque: otherwise the two column blocks' stores are paired as ds_write2_b32, whose 8-bit offsets need an add per pair)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        *reinterpret_cast<float*>(smem + eo + ((i & 1) * 32 + (r & 3) + 8 * (r >> 2)) * 1024) = acc[i][jl][r];
                }
        }
        const int co = (nb << 7) + h * 64 + cq * 4;
        f32x4 cs = {1.f, 1.f, 1.f, 1.f}, bs = {0.f, 0.f, 0.f, 0.f};
        if (p.col_scale) cs = *reinterpret_cast<const f32x4*>(p.col_scale + (size_t)b * p.col_ld + co);
        if (p.bias) bs = *reinterpret_cast<const f32x4*>(p.bias + co);
        cs *= p.alpha;
        float q[3][4];
        if (RGB) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float sr = p.rgb_s[(size_t)b * p.rgb_ld + co + c] * p.rgb_scale;
#pragma unroll
                for (int o = 0; o < 3; ++o) q[o][c] = p.rgb_w[o * p.Co + co + c] * sr;
            }
        }
        __syncthreads();
        if (WGS_W16ABL != 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int T = (tid >> 4) + 32 * k;
                const int oy = by * TR + (T >> 4), ox = bx * 32 + 2 * (T & 15);
                const unsigned char* e = smem + T * 1024 + cq * 16;
                const f32x4 m0 = *reinterpret_cast<const f32x4*>(e), m1 = *reinterpret_cast<const f32x4*>(e + 256),
                            m2 = *reinterpret_cast<const f32x4*>(e + 512), m3 = *reinterpret_cast<const f32x4*>(e + 768);
                float nz0 = 0.f, nz1 = 0.f;
                if (p.noise) { const float2 t2 = *reinterpret_cast<const float2*>(p.noise + oy * p.W + ox); nz0 = nw * t2.x; nz1 = nw * t2.y; }
                f32x4 y0 = m0 + m1 + m2, y1 = m1 - m2 - m3;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v0 = y0[c] * cs[c] + (nz0 + bs[c]), v1 = y1[c] * cs[c] + (nz1 + bs[c]);
                    v0 = fmaxf(v0, v0 * slope) * gain; v1 = fmaxf(v1, v1 * slope) * gain;
                    y0[c] = v0; y1[c] = v1;
                    vmax = fmaxf(vmax, fmaxf(fabsf(v0), fabsf(v1)));
                    if (RGB) {
#pragma unroll
                        for (int o = 0; o < 3; ++o) { racc[k][0][o] = __builtin_fmaf(v0, q[o][c], racc[k][0][o]); racc[k][1][o] = __builtin_fmaf(v1, q[o][c], racc[k][1][o]); }
                    }
                }
                if (!RGB || yb) {
                    float* dst = yb + ((size_t)oy * p.W + ox) * p.Co + co;
                    *reinterpret_cast<f32x4*>(dst) = y0;
                    *reinterpret_cast<f32x4*>(dst + p.Co) = y1;
                }
            }
        }
        if (h == 0) __syncthreads();
    }
    if (RGB) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int T = (tid >> 4) + 32 * k;
            const int oy = by * TR + (T >> 4), ox = bx * 32 + 2 * (T & 15);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float t0 = wgsconv::dpp_sum16(racc[k][j][0]), t1 = wgsconv::dpp_sum16(racc[k][j][1]), t2 = wgsconv::dpp_sum16(racc[k][j][2]);
                // (Co > 128: this channel block's PARTIAL sums, pixel = 4 * Co / 128 floats, block nb at floats 4 nb .. 4 nb + 3; the finishing launch adds them)
                if (cq == 0) *reinterpret_cast<f32x4*>(p.rgb_out + (((size_t)b * p.H * p.W + (size_t)oy * p.W + ox + j) * ntn + nb) * 4) = (f32x4){t0, t1, t2, 0.f};
            }
        }
    }
    if (p.y_amax) {
        vmax = wave_max(vmax);
        if (lane == 0) raise_amax(p.y_amax, vmax);
    }
}

bool w16_taps(const wgs_conv_desc* d, W16Taps& tp) {
    if (d->ntaps != 9) return false;
    for (int s = 0; s < 9; ++s) tp.w_of[s] = -1;
    for (int t = 0; t < 9; ++t) {
        const int dy = d->dy[t], dx = d->dx[t];
        if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
        int& slot = tp.w_of[(dy + 1) * 3 + dx + 1];
        if (slot >= 0 || d->wt[t] < 0) return false;
        slot = d->wt[t];
    }
    return true;
}

bool w16_ok(const wgs_conv_desc* d) {
    if (!d || !d->x || !d->w || d->x_f16 || d->col_stats || d->a_pixelnorm_eps > 0.f) return false;
    // ToRGB in the epilogue: a tile holds 128 channels of its pixels — every channel at Co == 128, else one of Co / 128 partial sums per pixel; y optional
    if (d->rgb_out ? !(d->Co <= 512 && d->rgb_s && d->rgb_w && d->rgb_ld >= d->Co) : !d->y) return false;
    W16Taps tp;
    if (!w16_taps(d, tp)) return false;
    if (!(d->isy == 1 && d->isx == 1 && d->osy == 1 && d->osx == 1 && d->oy0 == 0 && d->ox0 == 0 && d->ups == 0 && d->Hg == d->Hi && d->Wg == d->Wi &&
          d->Ho == d->Hi && d->Wo == d->Wi && d->Hi % TR == 0 && d->Wi % 32 == 0 && d->Ci % 32 == 0 && d->Co % 128 == 0 && d->act == 0 && !d->addend &&
          d->act_slope >= 0.f && d->act_slope <= 1.f && d->B > 0 && (!d->noise || d->noise_w))) return false;
    if ((long)d->Hi * d->Wi * d->Ci * 4 >= 0x7fffffffL || (long)d->Hi * d->Wi * d->Co * 4 >= 0x7fffffffL || (long)48 * d->Ci * d->Co >= 0x7fffffffL) return false;
    if ((d->a_ld > 0 ? d->a_ld : d->Ci) % 4 || (d->col_ld > 0 ? d->col_ld : d->Co) % 4) return false;       // 16-byte style / demodulation loads
    // fewer workgroups than CUs: the direct kernels' split-K forms fill the chip better
    return (long)d->B * (d->Hi / TR) * (d->Wi / 32) * (d->Co / 128) >= wgs_flags().wino16_min_wg;      // (WGS_WINO16_MIN_WG, default 200)
}

}  // namespace

extern "C" {

int wgs_conv_wino16_supported(const wgs_conv_desc* d) { return w16_ok(d) ? 1 : 0; }

int wgs_conv_wino16_weight(const wgs_conv_desc* d, uint16_t* U, wgs_stream_t stream) {
    WGS_CHECK_ARG(w16_ok(d) && U, "wgs_conv_wino16_weight: not a 3x3 stride-1 'same' launch the split-bf16 F(2,3) kernel covers (wgs_conv_wino16_supported)");
    W16Taps tp;
    w16_taps(d, tp);
    const long n = (long)d->Ci * d->Co;
    WGS_LAUNCH(wino16_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d->w, U, d->Ci, d->Co,
               (long)d->w_row_stride, (long)d->w_tap_stride, tp);
    WGS_CHECK_LAUNCH("wino16_weight_kernel");
    return WGS_OK;
}

int wgs_conv_wino16(const wgs_conv_desc* d, const uint16_t* U, wgs_stream_t stream) {
    WGS_CHECK_ARG(w16_ok(d) && U, "wgs_conv_wino16: not a 3x3 stride-1 'same' launch the split-bf16 F(2,3) kernel covers (wgs_conv_wino16_supported)");
    W16Args a;
    a.x = d->x; a.U = U; a.y = d->y; a.a_scale = d->a_scale; a.col_scale = d->col_scale; a.bias = d->bias; a.noise = d->noise; a.noise_w = d->noise_w;
    a.y_amax = d->y_amax;
    a.rgb_out = d->rgb_out; a.rgb_s = d->rgb_s; a.rgb_w = d->rgb_w; a.rgb_scale = d->rgb_scale; a.rgb_ld = d->rgb_ld;
    a.B = d->B; a.H = d->Hi; a.W = d->Wi; a.Ci = d->Ci; a.Co = d->Co;
    a.a_ld = d->a_ld > 0 ? d->a_ld : d->Ci; a.col_ld = d->col_ld > 0 ? d->col_ld : d->Co;
    a.alpha = d->alpha != 0.f ? d->alpha : 1.f; a.act_slope = d->act_slope; a.gain = d->gain;
    const unsigned grid = (unsigned)((long)d->B * (d->Hi / TR) * (d->Wi / 32) * (d->Co / 128));
    hipStream_t st = (hipStream_t)stream;
#define WGS_W16_LAUNCH(STY, RGB)                                                                                     \
    {                                                                                                                \
        auto k = wino16_kernel<STY, RGB>;                                                                            \
        wgs_note_kernel("wino16_kernel<%s, %s>", STY ? "true" : "false", RGB ? "true" : "false");                    \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);                  \
        WGS_LAUNCH(k, dim3(grid), dim3(512), SMEM, st, a);                                                           \
    }
    if (d->rgb_out) { if (d->a_scale) WGS_W16_LAUNCH(true, true) else WGS_W16_LAUNCH(false, true) }
    else { if (d->a_scale) WGS_W16_LAUNCH(true, false) else WGS_W16_LAUNCH(false, false) }
#undef WGS_W16_LAUNCH
    WGS_CHECK_LAUNCH("wino16_kernel");
    return WGS_OK;
}

}  // extern "C"
