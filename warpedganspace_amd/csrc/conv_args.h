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
    signed char dy[64], dx[64];
    short wt[64];
};


// split-bf16 (3 x v_mfma_f32_32x32x16_bf16 per product block) variant; returns 0 when it handled the launch,
// 1 when the shape is not supported (caller falls back to the exact fp32 kernel).
int launch_bf16x3(const ConvArgs& a, hipStream_t st);

}  // namespace wgsconv
