// Implicit-GEMM convolution in EXACT fp32 on the matrix cores (v_mfma_f32_32x32x2_f32) — the reference's arithmetic — through the
// register-staged, slot-interleaved kernel template of conv_nt_kernel.inc (scheme 4, conv_scheme.h).
//
// Why a second fp32 kernel: igemm_nt_kernel (conv_igemm.hip) runs a chunk as three phases — issue the next chunk's loads,
// 64 MFMAs, wait + style multiply + 32 scalar LDS stores — and with two workgroups per CU the two waves that share a SIMD
// fall into step (both multiply, then both stage), so the matrix pipe idles a third of the time (104 of 157 TFLOP/s).  Here a
// chunk is 4*TM slots of 4*TN MFMAs (256 cycles each per 32x32 block); the vector loads of a later chunk and the LDS stores of
// the staged one are issued INSIDE the slots, in the shadow of the 64-cycle MFMAs, operand fragments are one ds_read_b128 per
// four MFMAs (rows padded to 144 B: conflict-free), and out-of-image taps are hardware range checks on buffer loads.
// Covers what the 16-bit template covers: plain / strided / transposed-phase / up-sampled gathers, split-K, style and
// demodulation, merged sub-pixel phases.  Results equal igemm_nt_kernel's up to the summation order inside a 32-deep chunk.
#include "wgs_common.h"
#include "conv_args.h"
#include "conv_epilogue.h"
#include "conv_scheme.h"

typedef wgsconv::epi_f32x16 f32x16;

namespace {

#include "conv_nt_kernel.inc"

template <int BM, int BN, int WAVES_M, int WAVES_N>
void launch(ConvArgs& a, hipStream_t st) { launch_s<4, BM, BN, WAVES_M, WAVES_N>(a, st); }

template <int BM, int BN, int WAVES_M, int WAVES_N>
void launch_big(ConvArgs& a, hipStream_t st, int nblocks = 0) { launch_big_s<4, BM, BN, WAVES_M, WAVES_N>(a, st, nblocks); }

}  // namespace

namespace wgsconv {

#define WGS_NT_16BIT 0
// 8-wave 256 x 256 tiles where Cout allows (144 vs 137 TFLOP/s on 512 -> 512 @64^2; the 256 x 128 tile of Cout = 128 loses to two
// 128 x 128 workgroups per CU: 129 vs 134) and the merged sub-pixel phases of the up-convs; WGS_F32_SMALL: 4-wave tiles only
#define WGS_NT_BIG_TILES (!wgs_flags().f32_small)
#define WGS_NT_LAUNCH_NAME launch_f32
#define WGS_NT_MULTI_NAME launch_f32_multi
#include "conv_nt_launch.inc"

}  // namespace wgsconv
