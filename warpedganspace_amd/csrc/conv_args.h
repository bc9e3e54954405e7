// Shared launch arguments of the implicit-GEMM convolution kernels (conv_igemm.hip: exact fp32 MFMA;
// conv_igemm_bf16.hip: split-bf16 MFMA).  Semantics: include/wgs.h, wgs_conv_desc.
#pragma once
#include "wgs_common.h"

namespace wgsconv {

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
};

inline void fill_tap_tables(ConvArgs& a) {
    for (int t = 0; t < a.ntaps; ++t) {
        a.tap_yx[t] = ((int)a.dy[t] & 0xffff) | ((int)a.dx[t] << 16);
        a.tap_a[t] = ((int)a.dy[t] * a.Wi + (int)a.dx[t]) * a.Ci * 4;
        a.tap_w[t] = (int)((long)a.wt[t] * a.w_tap_stride * 4);
    }
}


// split-bf16 (3 x v_mfma_f32_32x32x16_bf16 per product block) variant; returns 0 when it handled the launch,
// 1 when the shape is not supported (caller falls back to the exact fp32 kernel).
int launch_bf16x3(const ConvArgs& a, hipStream_t st);

}  // namespace wgsconv
