// Operand schemes of the 16-bit-input implicit-GEMM kernels (wgs_conv_desc.precision 1..3).  A product block of the GEMM is a
// sum of NP v_mfma_f32_32x32x16_{bf16,f16} instructions over the (plane of A) x (plane of B) pairs listed below; the
// accumulator, the style / demodulation factors and the whole epilogue stay fp32 in every scheme.
//
//   SCH 0  split-bf16 x3 : A = {hi, lo}, B = {hi, lo} bf16;  lo*hi + hi*lo + hi*hi         ~2^-16 per product, 3 MFMAs
//   SCH 1  fp16          : A = {hi},     B = {hi}     fp16;  hi*hi                          ~2^-11 per operand, 1 MFMA
//   SCH 2  fp16 x2       : A = {hi},     B = {hi, lo} fp16;  hi*lo + hi*hi (weights ~2^-22) ~2^-11 on the activation only, 2 MFMAs
//   SCH 3  fp16 x2 (A)   : A = {hi, lo}, B = {hi}     fp16;  lo*hi + hi*hi (activations ~2^-22) ~2^-11 on the weights only, 2 MFMAs
//
// fp16 has 5 exponent bits: the activation operand is multiplied by a power of two chosen from a device-resident bound of
// its magnitude (wgs_conv_desc.a_amax) before it is rounded, and the accumulator is multiplied back in the epilogue, so
// that gradients of any magnitude keep their 11 significant bits (see operand_scale below).
#pragma once
#include <hip/hip_runtime.h>

namespace wgsconv {

typedef float sch_f32x4 __attribute__((ext_vector_type(4)));
typedef float sch_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 sch_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sch_bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 sch_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sch_f16x4 __attribute__((ext_vector_type(4)));

template <int SCH> struct Scheme;

template <> struct Scheme<0> {
    static constexpr int EB = 2;      // bytes per staged operand element
    static constexpr int NA = 2, NB = 2, NP = 3;
    typedef sch_bf16x8 frag;
    // 4 floats -> packed 16-bit hi (and lo) words
    static __device__ __forceinline__ void cvt4(const sch_f32x4 f, uint2& hi, uint2& lo) {
        const sch_bf16x4 h = __builtin_convertvector(f, sch_bf16x4);
        const sch_f32x4 r = f - __builtin_convertvector(h, sch_f32x4);
        const sch_bf16x4 l = __builtin_convertvector(r, sch_bf16x4);
        hi = __builtin_bit_cast(uint2, h);
        lo = __builtin_bit_cast(uint2, l);
    }
    // a[0] = hi, a[1] = lo; same for b
    static __device__ __forceinline__ sch_f32x16 mma(const frag* a, const frag* b, sch_f32x16 c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
        return c;
    }
};

template <> struct Scheme<1> {
    static constexpr int EB = 2;      // bytes per staged operand element
    static constexpr int NA = 1, NB = 1, NP = 1;
    typedef sch_f16x8 frag;
    static __device__ __forceinline__ void cvt4(const sch_f32x4 f, uint2& hi, uint2& lo) {
        const sch_f16x4 h = __builtin_convertvector(f, sch_f16x4);      // round to nearest even
        hi = __builtin_bit_cast(uint2, h);
        lo = hi;
    }
    static __device__ __forceinline__ sch_f32x16 mma(const frag* a, const frag* b, sch_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c, 0, 0, 0);
    }
};

template <> struct Scheme<2> {
    static constexpr int EB = 2;      // bytes per staged operand element
    static constexpr int NA = 1, NB = 2, NP = 2;
    typedef sch_f16x8 frag;
    static __device__ __forceinline__ void cvt4(const sch_f32x4 f, uint2& hi, uint2& lo) {
        const sch_f16x4 h = __builtin_convertvector(f, sch_f16x4);
        const sch_f32x4 r = f - __builtin_convertvector(h, sch_f32x4);
        const sch_f16x4 l = __builtin_convertvector(r, sch_f16x4);
        hi = __builtin_bit_cast(uint2, h);
        lo = __builtin_bit_cast(uint2, l);
    }
    static __device__ __forceinline__ sch_f32x16 mma(const frag* a, const frag* b, sch_f32x16 c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c, 0, 0, 0);
        return c;
    }
};

// fp16 x2 with the ACTIVATION split instead of the weights: A = {hi, lo}, B = {hi}; lo*hi + hi*hi.  Same two MFMAs and the same
// error class as SCH 2 (one operand to ~2^-22, the other rounded to 11 bits); used where the weight planes' LDS-DMA traffic,
// not the MFMA rate, bounds the kernel (conv_upfused.hip): the second weight plane would double that traffic.
template <> struct Scheme<3> {
    static constexpr int EB = 2;      // bytes per staged operand element
    static constexpr int NA = 2, NB = 1, NP = 2;
    typedef sch_f16x8 frag;
    static __device__ __forceinline__ void cvt4(const sch_f32x4 f, uint2& hi, uint2& lo) { Scheme<2>::cvt4(f, hi, lo); }
    static __device__ __forceinline__ sch_f32x16 mma(const frag* a, const frag* b, sch_f32x16 c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c, 0, 0, 0);
        return c;
    }
};

// SCH 4: exact fp32 — the reference's arithmetic on the matrix cores.  One fp32 plane per operand; a fragment is FOUR consecutive
// k values of a row (one ds_read_b128), consumed by four v_mfma_f32_32x32x2_f32: the instruction's two k slots are fed by
// the two half-waves, which read adjacent 16-byte groups of the same 32-byte k block (lanes 0-31: k = 8j + e, lanes 32-63:
// k = 8j + 4 + e for MFMA e) — A and B use the same pairing, and a contraction does not care in which order its k are summed.
template <> struct Scheme<4> {
    static constexpr int EB = 4;
    static constexpr int NA = 1, NB = 1, NP = 4;
    typedef sch_f32x4 frag;
    static __device__ __forceinline__ void cvt4(const sch_f32x4 f, uint2& hi, uint2& lo) { hi = make_uint2(0, 0); lo = hi; }   // unused
    static __device__ __forceinline__ sch_f32x16 mma(const frag* a, const frag* b, sch_f32x16 c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][0], b[0][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][1], b[0][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][2], b[0][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][3], b[0][3], c, 0, 0, 0);
        return c;
    }
};

// LDS row of a staged 32-deep K chunk: 32 elements + 16 bytes of padding (80 B for the 16-bit planes, 144 B for fp32: the 8
// rows that a ds_read_b128 / ds_write_b128 serves per cycle then start 4 banks apart — conflict-free in both layouts)
template <int SCH> struct SchemeRow { static constexpr int BYTES = 32 * Scheme<SCH>::EB + 16; };

// Power-of-two operand scale for the fp16 schemes.  `amax` bounds |A| (before the per-sample a_scale factor, which the
// caller's bound must include).  Returns mult = 2^k with amax * mult in [2^11, 2^12) — four binades below the fp16 maximum
// (2^16), 26 above its smallest normal number — and inv = 2^-k for the accumulator.  amax == 0 / NaN / inf: no scaling.
// amax2 (optional second device scalar) multiplies the bound: forward launches pass max|x| of the producing layer and
// max|style| separately.
// mult = 2^k with am * 2^k in [2^11, 2^12), inv = 2^-k; am == 0 / NaN / inf / denormal: no scaling
__device__ __forceinline__ void scale_of_bound(float am, float& mult, float& inv) {
    mult = 1.f; inv = 1.f;
    const int e = (__float_as_int(am) >> 23) & 0xff;       // am in [2^(e-127), 2^(e-126))
    if (!(am > 0.f) || e == 0 || e == 255) return;
    int k = 127 + 12 - (e - 126);                         // biased exponent of 2^(12 - (e - 126))
    k = k < 1 ? 1 : (k > 253 ? 253 : k);
    mult = __int_as_float(k << 23);
    inv = __int_as_float((254 - k) << 23);
}

__device__ __forceinline__ void operand_scale(const float* amax_ptr, const float* amax2_ptr, float bound, float& mult, float& inv) {
    mult = 1.f; inv = 1.f;
    if (!amax_ptr) return;
    scale_of_bound(amax_ptr[0] * bound * (amax2_ptr ? amax2_ptr[0] : 1.f), mult, inv);
}

}  // namespace wgsconv
