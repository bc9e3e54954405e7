// Implicit-GEMM convolution on the bf16 matrix cores with fp32-class accuracy ("split-bf16 x3").
//
// Same GEMM view, descriptor, prologue (style scale) and epilogue (demod / noise / bias / addend / activation) as
// conv_igemm.hip, but every fp32 operand value v is split while it is staged into LDS,
//     hi = bf16_rn(v),  lo = bf16_rn(v - hi)            (v = hi + lo up to 2^-17 relative)
// and each 32x32x16 product block is evaluated as  hi*hi + hi*lo + lo*hi  with three
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate; the dropped lo*lo term is <= 2^-16 relative).  The bf16 MFMA
// issues in 8 cycles/CU against 64 for v_mfma_f32_32x32x2_f32 x 8 k-steps, so three of them cost 3/16 of the
// exact fp32 path: a 5.3x higher matrix-core ceiling (833 TFLOP/s fp32-equivalent) at ~1e-5 relative error.
//
// LDS image per stage: A_hi, A_lo [BM][32] bf16 and B_hi, B_lo [BN][32] bf16, rows padded to 80 bytes so that the
// 16-lane groups of ds_read_b128 (one lane = one row, 8 consecutive k) hit 16 distinct 4-bank slots.
// 2 stages x 40 KiB = 80 KiB per workgroup -> two workgroups per CU.
//
// The same kernels also run the fp16 operand schemes (conv_scheme.h: one fp16 plane per operand and 1 MFMA per product
// block, or two weight planes and 2 MFMAs) — template parameter SCH; everything below that says "hi / lo" then has
// NA / NB planes per operand.
#include "wgs_common.h"
#include "conv_args.h"
#include "conv_epilogue.h"
#include "conv_scheme.h"

typedef wgsconv::epi_f32x16 f32x16;

namespace {

#include "conv_nt_kernel.inc"

// Second pass of a split-K launch: y[pix(m)][n] = epilogue(sum_s ws[s][m][n]) — the same epilogue as above
// (alpha, demod, noise, bias, addend, activation).  One thread per 4 output channels.
// A thread finishes SK_ROWS consecutive GEMM rows of its 4 channels (a wave's lanes = adjacent channel quads of the same rows: coalesced
// either way).  One row per thread without column statistics — the form every generator launch takes (eight rows per thread, tried for all
// launches, doubled this kernel's time: 11.5 -> 21.4 us per launch, +0.45 ms per step: it is latency-bound and wants the threads) — four
// with them (wgs_conv_desc.col_stats: 8 atomics per thread and four rows).
template <int SK_ROWS>
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(const ConvArgs p) {
    const int c4 = p.Co / 4;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const int groups = (p.M + SK_ROWS - 1) / SK_ROWS;
    if (e >= (long)groups * c4) return;
    const int m_begin = (int)(e / c4) * SK_ROWS, n = (int)(e % c4) * 4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float am = 0.f;
    for (int m = m_begin; m < min(p.M, m_begin + SK_ROWS); ++m) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < p.ksplit; ++s) {
            const float4 t = *reinterpret_cast<const float4*>(p.ws + ((size_t)s * p.M + m) * p.Co + n);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        const int b = m / p.Mimg, pq = m - b * p.Mimg;
        if (pq >= p.HW) continue;            // padding row of a sample (see launch_bf16x3)
        const int gy = pq / p.Wg, gx = pq - gy * p.Wg;
        const int oy = gy * p.osy + p.oy0, ox = gx * p.osx + p.ox0;
        const int hw = oy * p.Wo + ox;
        const float nz = (p.noise && p.noise_w) ? p.noise_w[0] * p.noise[hw] : 0.f;
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float u = o[k] * p.alpha;
            if (p.col_scale) u *= p.col_scale[(size_t)b * p.col_ld + n + k];
            u += nz + (p.bias ? p.bias[n + k] : 0.f);
            if (p.addend) u += p.addend[((size_t)(b * (p.Ho >> p.add_ups) + (oy >> p.add_ups)) * (p.Wo >> p.add_ups) + (ox >> p.add_ups)) * p.Co + n + k];
            o[k] = (p.act == 1) ? tanhf(u) : (u > 0.f ? u : u * p.act_slope) * p.gain;
            s1[k] += o[k]; s2[k] = fmaf(o[k], o[k], s2[k]);
        }
        *reinterpret_cast<float4*>(p.y + ((size_t)b * p.Ho * p.Wo + hw) * p.Co + n) = make_float4(o[0], o[1], o[2], o[3]);
        am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
    }
    if (p.y_amax) raise_amax(p.y_amax, am);     // magnitude bound for the next layer's fp16 operand scale
    if (p.col_stats) {
        double* wr = p.col_stats + (size_t)(blockIdx.x % (unsigned)wgs_bn_nrep(p.Co)) * 2 * p.Co;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsafeAtomicAdd(wr + n + k, (double)s1[k]);
            unsafeAtomicAdd(wr + p.Co + n + k, (double)s2[k]);
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
void launch(ConvArgs& a, hipStream_t st) {
    if (a.sch == 0) launch_s<0, BM, BN, WAVES_M, WAVES_N>(a, st);
    else if (a.sch == 1) launch_s<1, BM, BN, WAVES_M, WAVES_N>(a, st);
    else launch_s<2, BM, BN, WAVES_M, WAVES_N>(a, st);
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
void launch_big(ConvArgs& a, hipStream_t st, int nblocks = 0) {
    if (a.sch == 0) launch_big_s<0, BM, BN, WAVES_M, WAVES_N>(a, st, nblocks);
    else if (a.sch == 1) launch_big_s<1, BM, BN, WAVES_M, WAVES_N>(a, st, nblocks);
    else launch_big_s<2, BM, BN, WAVES_M, WAVES_N>(a, st, nblocks);
}

}  // namespace

namespace wgsconv {

void launch_splitk_epilogue(const ConvArgs& a, hipStream_t st) {
    if (a.col_stats) {
        const long work = (long)((a.M + 3) / 4) * (a.Co / 4);
        WGS_LAUNCH(conv_splitk_epilogue_kernel<4>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, a);
    } else {
        const long work = (long)a.M * (a.Co / 4);
        WGS_LAUNCH(conv_splitk_epilogue_kernel<1>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, a);
    }
}

// operand extents for the buffer descriptors; every stream must be addressable with a 31-bit byte offset
bool set_extents(ConvArgs& a, int wt_max) {
    // a producer-written fp16 plane (a_hi, wgs_conv_desc.x_f16) is addressed as such: 2 bytes per element
    const long xb = (long)a.B * a.Hi * a.Wi * a.Ci * (a.a_hi ? 2 : 4);
    const long wb = ((long)wt_max * a.w_tap_stride + (long)(a.Co - 1) * a.w_row_stride + a.Ci) * 4;
    const long sb = a.a_scale ? ((long)(a.B - 1) * a.a_ld + a.Ci) * 4 : 0;
    const long lim = 0x7fffffffL;
    if (xb > lim || wb > lim || sb > lim) return false;
    a.x_bytes = (int)xb; a.w_bytes = (int)wb; a.s_bytes = (int)sb;
    return true;
}

// LDS-DMA path (conv_igemm_dma.hip) for a launch that takes the 8-wave tiles: needs pre-split weights and a workspace for
// the split activation planes (4 bytes per input element).  Runs the modulate+split pre-pass and the DMA kernel.
static bool try_dma(ConvArgs& a, int bn, int nblocks, hipStream_t st) {
    const long elems = (long)a.B * a.Hi * a.Wi * a.Ci;
    if (!a.w_hi || (!a.w_lo && a.sch != 1) || !a.ws || a.ws_bytes < elems * 4 || WGS_ABL == 15) return false;
    // the pre-pass reads + writes 8 bytes per input element whatever Cout is; measured (B=32): Cout=512 +15 % net,
    // Cout=256 / 128 -2 % net (the DMA kernel alone is 13-21 % faster) — take it only where it pays
    if (a.Co < 512 && !wgs_flags().dma_always) return false;
    unsigned short* hi = reinterpret_cast<unsigned short*>(a.ws);
    unsigned short* lo = hi + elems;
    if (a.sch == 0) split_bf16(a.x, a.a_scale, a.a_ld, hi, lo, a.B, (long)a.Hi * a.Wi * a.Ci, a.Ci, st);
    else split_f16(a.x, a.a_scale, a.a_ld, hi, nullptr, a.B, (long)a.Hi * a.Wi * a.Ci, a.Ci, a.a_amax, a.a_amax2, a.a_bound, st);
    a.a_hi = hi; a.a_lo = lo;
    a.x_bytes /= 2; a.w_bytes /= 2;            // extents of the bf16 planes
    launch_dma_bf16x3(a, bn, nblocks, st);
    return true;
}

#define WGS_NT_16BIT 1
#define WGS_NT_BIG_TILES 1
#define WGS_NT_LAUNCH_NAME launch_bf16x3
#define WGS_NT_MULTI_NAME launch_bf16x3_multi
#include "conv_nt_launch.inc"

}  // namespace wgsconv
