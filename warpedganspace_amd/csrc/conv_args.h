// Shared launch arguments of the implicit-GEMM convolution kernels (conv_igemm.hip: exact fp32 MFMA;
// conv_igemm_bf16.hip: split-bf16 MFMA).  Semantics: include/wgs.h, wgs_conv_desc.
#pragma once
#include "wgs_common.h"

namespace wgsconv {

// One output "phase" of a launch: the GEMM-row geometry and tap tables that differ between the sub-pixel phases of a
// stride-2 transposed conv.  The split-bf16 kernel always reads phase-dependent values from ph[] (ph[0] for an
// ordinary conv), so that the four phases of an up-conv can run as ONE launch (wgs_conv_igemm_multi).
struct PhaseArgs {
    int Wg, oy0, ox0, ntaps, HW, Mimg, M, tiles, cnt8;   // merged launches: (m,n) tiles of this phase, and ceil(tiles/8) per XCD
    int tap_yx[16], tap_a[16], tap_w[16];
};

struct ConvArgs {
    const float* x;
    const float* w;
    float* y;
    const float* a_scale;
    const float* col_scale;
    const float* bias;
    const float* noise;
    const float* noise_w;
    const float* addend;
    int B, Hi, Wi, Ci, Hg, Wg, isy, isx, Ho, Wo, Co, osy, osx, oy0, ox0, ntaps, M, a_ld, col_ld, ups, add_ups, act;
    long w_tap_stride, w_row_stride;
    float act_slope, gain, alpha;
    int HW, Mimg;       // Hg*Wg, and GEMM rows per sample (>= HW; M = B*Mimg).  Mimg == HW except in launch_bf16x3
    const unsigned short* w_hi;     // pre-split 16-bit planes of w (same packed layout; bf16 for sch 0, fp16 for sch 1 / 2), or null
    const unsigned short* w_lo;
    int sch;                        // operand scheme (conv_scheme.h): precision - 1
    const float* a_amax;            // fp16 schemes: device scalar bounding |x| (dynamic power-of-two operand scale), or null
    float a_bound;                  // ... times this factor (bound of |a_scale|, blur gain ...)
    const float* a_amax2;           // ... times this optional second device scalar (bound of |a_scale| in forward launches)
    float* y_amax;                  // optional device scalar raised (atomic max) to max |y| of this launch: the next layer's a_amax
    float* rgb_out; const float* rgb_s; const float* rgb_w; float rgb_scale; int rgb_ld;      // ToRGB in the epilogue (wgs_conv_desc.rgb_out)
    float pn_eps;                   // > 0: the operand is PixelNorm(x) (wgs_conv_desc.a_pixelnorm_eps; conv_halo16.hip only)
    double* col_stats;              // per-channel sum y / sum y^2 of the output into BatchNorm scratch (wgs_conv_desc.col_stats), or null
    const unsigned short* a_hi;     // split (and style-modulated) activation planes: set by launch_bf16x3 (LDS-DMA path)
    const unsigned short* a_lo;
    float* ws;          // split-K workspace or null
    long ws_bytes;
    int ksplit;         // K splits of this launch (1 = none); set by launch_bf16x3
    int x_bytes, w_bytes, s_bytes;      // operand extents (bytes), filled in by launch_bf16x3 for its buffer descriptors
    signed char dy[64], dx[64];
    short wt[64];
    // 32-bit per-tap tables (dword entries so that the uniform per-iteration lookups are scalar loads — gfx9 has no
    // sub-dword scalar loads, and a vector load of a byte table entry makes the kernel drain its whole vmcnt queue):
    int tap_yx[64];     // (dy & 0xffff) | (dx << 16)
    int tap_a[64];      // byte offset of the tap inside x for the non-upsampling case: ((dy*Wi + dx) * Ci) * 4
    int tap_w[64];      // byte offset of the tap's weight slab: wt * w_tap_stride * 4
    int nphase;
    PhaseArgs ph[4];
};

inline void fill_tap_tables(ConvArgs& a) {
    for (int t = 0; t < a.ntaps; ++t) {
        a.tap_yx[t] = ((int)a.dy[t] & 0xffff) | ((int)a.dx[t] << 16);
        a.tap_a[t] = ((int)a.dy[t] * a.Wi + (int)a.dx[t]) * a.Ci * 4;
        a.tap_w[t] = (int)((long)a.wt[t] * a.w_tap_stride * 4);
    }
}


// Split-K policy shared by both kernels: `tiles` 128x128 output tiles, `nk` K-chunks.  Returns the number of splits
// (1 = none): fill ~256 CUs, at least 6 chunks per split, at most 16, bounded by the caller's workspace.
inline int choose_ksplit(const ConvArgs& a, int tiles, int nk) {
    if (!a.ws || tiles > 128 || a.Co % 4 != 0) return 1;
    int ks = 256 / tiles;
    if (ks > nk / 6) ks = nk / 6;
    const long per = (long)a.M * a.Co * 4;
    if ((long)ks * per > a.ws_bytes) ks = (int)(a.ws_bytes / per);
    if (ks > 16) ks = 16;
    if (ks < 2) return 1;
    const int kper = (nk + ks - 1) / ks;
    return (nk + kper - 1) / kper;      // no empty splits
}

// second pass of a split-K launch: reduce ws[split][M][Co] and apply the epilogue (conv_igemm_bf16.hip)
void launch_splitk_epilogue(const ConvArgs& a, hipStream_t st);

// 16-bit-operand kernels (a.sch selects split-bf16 x3 / fp16 / fp16 x2, conv_scheme.h); returns 0 when it handled the launch,
// 1 when the shape is not supported (caller falls back to the exact fp32 kernel).
int launch_bf16x3(const ConvArgs& a, hipStream_t st);
// the same for n <= 4 launches that differ only in (Hg, Wg, oy0, ox0, taps); 0 = handled as one merged launch
int launch_bf16x3_multi(const ConvArgs* a, int n, hipStream_t st, bool dry = false);   // dry: decide only, launch nothing

// The same kernel template in exact fp32 (scheme 4, conv_igemm_f32.hip): a.sch must be 4; 0 = handled, 1 = shape not covered
// (Ci % 32, > 16 taps, operands beyond 2 GiB: the caller keeps the plain fp32 kernel of conv_igemm.hip)
int launch_f32(const ConvArgs& a, hipStream_t st);
int launch_f32_multi(const ConvArgs* a, int n, hipStream_t st, bool dry = false);
// operand extents (x_bytes / w_bytes / s_bytes) for the buffer descriptors; false when a stream exceeds a 31-bit byte offset
bool set_extents(ConvArgs& a, int wt_max);

// LDS-DMA form of the 8-wave kernels (conv_igemm_dma.hip)
void split_bf16(const float* x, const float* s, int s_ld, unsigned short* hi, unsigned short* lo, long nsamples, long per_sample,
                int C, hipStream_t st);
// the same for the fp16 schemes: hi = f16(x * s * mult), lo = f16(residual) (lo may be null); mult from (a_amax, a_bound)
void split_f16(const float* x, const float* s, int s_ld, unsigned short* hi, unsigned short* lo, long nsamples, long per_sample,
               int C, const float* a_amax, const float* a_amax2, float a_bound, hipStream_t st);
void launch_dma_bf16x3(const ConvArgs& a, int bn, int nblocks, hipStream_t st);

// few-channel 3x3 stride-1 convs on large maps (conv_halo16.hip): 0 = launch taken.  Needs x_bytes / w_bytes and the tap tables.
int launch_halo16(const ConvArgs& a, hipStream_t st, bool dry = false);     // dry: decide only, launch nothing

// patch form for stride-1 convs (conv_igemm_patch.hip); 0 = launch taken.  Needs x_bytes / w_bytes (fp32 extents) and the
// 64-entry tap tables filled.
int launch_patch_bf16x3(const ConvArgs& a, hipStream_t st);
// ... its 128 x 128 tile for a producer-written fp16 plane in plain fp16, every operand by LDS-DMA (conv_patch_dma.hip); 0 = launch taken.
// a.w_bytes: extent of the 16-bit weight plane.
int launch_patch_dma(const ConvArgs& a, hipStream_t st);

}  // namespace wgsconv
