// 4x4 FIR (StyleGAN2 blur, up = down = 1) on NHWC with a sliding row window.
//
//   out[b, oy, ox, c] = sum_{ky,kx} kf[ky][kx] * in[b, oy + ky - py0, ox + kx - px0, c]        (zero outside the input)
//
// where kf is the FLIPPED kernel (upfirdn2d correlates with the flipped kernel, upfirdn2d.py:176-178).  One thread
// owns one (sample, 16-row strip, column, 4 channels) and walks down its strip keeping the last four input rows
// (4x5 float4: two adjacent output columns per thread) in registers: ~3 loads per output instead of 16, every load a
// fully coalesced row segment.
// EPI adds the StyleGAN2 noise + bias + leaky-relu*sqrt(2) epilogue (model.py:303-310, fused_act.py:91-99).
// F16 (without EPI): the result is stored as the fp16 operand plane of the consuming conv, hi = f16_rn(out * 2^k) with the
// power of two of conv_scheme.h's operand_scale(a_amax, a_bound) — exactly what that conv's own staging (or the LDS-DMA
// pre-pass) would make of the fp32 tensor, which is then never written: half the store bytes, no pre-pass.
// XF16 (with F16, round 6): the INPUT is an fp16 plane too, f16_rn(in * 2^k1) with k1 from operand_scale(a_amax, 1) (sg2_act_bwd's dy as the
// plane it writes for a gradient conv): read as 8 bytes per four channels, filtered in fp32, rescaled by the exact power of two 2^(k - k1).  Fed
// the same values, bit for bit the plane the fp32-input form writes (a power-of-two scaling commutes with every rounding of the filter).
#pragma once
#include "wgs_common.h"
#include "conv_scheme.h"

namespace wgsfir {

constexpr int TY = 16;

template <bool EPI, bool F16 = false, bool XF16 = false>
__global__ __launch_bounds__(256) void fir4_kernel(const float* __restrict__ x, const float* __restrict__ kern,
                                                   float* __restrict__ y, int B, int Hin, int Win, int Ho, int Wo, int C,
                                                   int py0, int px0, const float* __restrict__ noise,
                                                   const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                   float* __restrict__ y_amax, const float* __restrict__ a_amax, float a_bound) {
    static_assert(!(EPI && F16), "the fp16 plane is the operand of a gradient conv: no epilogue");
    static_assert(!XF16 || F16, "an fp16 input plane is filtered into an fp16 output plane");
    float op_mult = 1.f, op_inv = 1.f;
    if (F16) wgsconv::operand_scale(a_amax, nullptr, a_bound, op_mult, op_inv);
    if (XF16) {       // the input carries 2^k1: what is left to apply is 2^(k - k1)
        float in_mult, in_inv;
        wgsconv::operand_scale(a_amax, nullptr, 1.f, in_mult, in_inv);
        op_mult *= in_inv;
    }
    float kf[16];
    float vmax = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) kf[i] = kern[15 - i];
    const int c4n = C >> 2;
    const int strips = (Ho + TY - 1) / TY;
    const int wpairs = (Wo + 1) >> 1;                 // a thread owns TWO adjacent output columns: 5 loads per row for 2 outputs
    const long total = (long)B * strips * wpairs * c4n;
    const long e0 = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = e0 < total;         // threads past the end shadow the last element (no store): the wave stays converged
    const long e = live ? e0 : total - 1; // for the amax reduction below
    long r = e;
    const int c = (int)(r % c4n) * 4; r /= c4n;
    const int ox = (int)(r % wpairs) * 2; r /= wpairs;
    const int oy0 = (int)(r % strips) * TY;
    const int b = (int)(r / strips);
    const bool two = ox + 1 < Wo;
    const float nw = (EPI && noise) ? noise_w[0] : 0.f;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI) bv = *reinterpret_cast<const float4*>(bias + c);
    const float* xb = x + (size_t)b * Hin * Win * C + c;
    const unsigned short* xh = reinterpret_cast<const unsigned short*>(x) + (size_t)b * Hin * Win * C + c;
    float* yb = y + (size_t)b * Ho * Wo * C + c;
    const int ix0 = ox - px0;
    float4 win[4][5];
#pragma unroll
    for (int rr = 0; rr < TY + 3; ++rr) {
        const int iy = oy0 - py0 + rr;
        const bool rowok = iy >= 0 && iy < Hin;
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const int ix = ix0 + kx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rowok && ix >= 0 && ix < Win) {
                if (XF16) {
                    const uint2 h = *reinterpret_cast<const uint2*>(xh + ((size_t)iy * Win + ix) * C);
                    const wgsconv::sch_f16x4 hv = __builtin_bit_cast(wgsconv::sch_f16x4, h);
                    v = make_float4((float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]);
                } else v = *reinterpret_cast<const float4*>(xb + ((size_t)iy * Win + ix) * C);
            }
            win[rr & 3][kx] = v;
        }
        if (rr >= 3) {
            const int oy = oy0 + rr - 3;
            if (oy < Ho) {
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    if (px == 1 && !two) break;
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 4; ++kx) {
                            const float wv = kf[ky * 4 + kx];
                            const float4 v = win[(rr - 3 + ky) & 3][kx + px];
                            acc.x = fmaf(v.x, wv, acc.x); acc.y = fmaf(v.y, wv, acc.y);
                            acc.z = fmaf(v.z, wv, acc.z); acc.w = fmaf(v.w, wv, acc.w);
                        }
                    if (EPI) {
                        const float nz = noise ? nw * noise[oy * Wo + ox + px] : 0.f;
                        acc.x += nz + bv.x; acc.y += nz + bv.y; acc.z += nz + bv.z; acc.w += nz + bv.w;
                        acc.x = (acc.x > 0.f ? acc.x : 0.2f * acc.x) * 1.4142135623730951f;
                        acc.y = (acc.y > 0.f ? acc.y : 0.2f * acc.y) * 1.4142135623730951f;
                        acc.z = (acc.z > 0.f ? acc.z : 0.2f * acc.z) * 1.4142135623730951f;
                        acc.w = (acc.w > 0.f ? acc.w : 0.2f * acc.w) * 1.4142135623730951f;
                    }
                    if (EPI) vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
                    if (F16) {
                        const wgsconv::sch_f32x4 f = {acc.x * op_mult, acc.y * op_mult, acc.z * op_mult, acc.w * op_mult};
                        uint2 h, l;
                        wgsconv::Scheme<1>::cvt4(f, h, l);
                        unsigned short* yh = reinterpret_cast<unsigned short*>(y) + (size_t)b * Ho * Wo * C + c;
                        if (live) *reinterpret_cast<uint2*>(yh + ((size_t)oy * Wo + ox + px) * C) = h;
                    } else if (live) *reinterpret_cast<float4*>(yb + ((size_t)oy * Wo + ox + px) * C) = acc;
                }
            }
        }
    }
    // magnitude bound of the activation for the next layer's fp16 operand scale: one atomic per wave
    if (EPI && y_amax) {
        vmax = wave_max(vmax);
        if ((threadIdx.x & 63) == 0) raise_amax(y_amax, vmax);
    }
}

template <bool EPI, bool F16 = false, bool XF16 = false>
inline void launch_fir4(const float* x, const float* kern, float* y, int B, int Hin, int Win, int Ho, int Wo, int C, int py0,
                        int px0, const float* noise, const float* noise_w, const float* bias, hipStream_t st, float* y_amax = nullptr,
                        const float* a_amax = nullptr, float a_bound = 1.f) {
    const int strips = (Ho + TY - 1) / TY;
    const long total = (long)B * strips * ((Wo + 1) / 2) * (C / 4);
    WGS_LAUNCH((fir4_kernel<EPI, F16, XF16>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, kern, y, B, Hin, Win, Ho, Wo,
                       C, py0, px0, noise, noise_w, bias, y_amax, a_amax, a_bound);
}

}  // namespace wgsfir
