// Shared epilogue of the split-bf16 implicit-GEMM kernels (same contract as conv_igemm.hip): per-row output addressing
// and the noise term are staged once in LDS, then every accumulator gets alpha, demodulation, noise, bias, addend and the
// activation and is stored as a 128-B channel run per lane group.
#pragma once
#include "conv_args.h"

namespace wgsconv {

typedef float epi_f32x16 __attribute__((ext_vector_type(16)));

// second half: the staged row arrays (r_pix / r_b / r_nz / r_add in LDS, filled by the caller and published with a
// barrier) -> every accumulator gets alpha, demodulation, noise, bias, addend, activation and is stored
template <int BM, int TM, int TN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue_apply(const ConvArgs& p, epi_f32x16 (&acc)[TM][TN], const unsigned char* smem_b,
                                                    int n0, int wm, int wn, int l31, int lh, float alpha_mul = 1.f) {
    const float alpha = p.alpha * alpha_mul;      // alpha_mul: inverse of the fp16 schemes' power-of-two operand scale (exact)
    const int* r_pix = reinterpret_cast<const int*>(smem_b);
    const int* r_b = r_pix + BM;
    const float* r_nz = reinterpret_cast<const float*>(r_b + BM);
    const int* r_add = reinterpret_cast<const int*>(r_nz + BM);
    // demodulation factors: a tile usually covers one or two samples -> two registers per column
    const int b_lo = r_b[0], b_hi2 = r_b[BM - 1];
    const bool cs_fast = p.col_scale && (b_hi2 - b_lo <= 1);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + l31;
        const bool nok = n < p.Co;
        const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
        float cs0 = 1.f, cs1 = 1.f;
        if (cs_fast && nok) {
            cs0 = p.col_scale[(size_t)b_lo * p.col_ld + n];
            cs1 = p.col_scale[(size_t)b_hi2 * p.col_ld + n];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int pix = r_pix[row];
                if (pix >= 0 && nok) {
                    float v = acc[i][j][r] * alpha;
                    if (p.col_scale) v *= cs_fast ? (r_b[row] == b_lo ? cs0 : cs1) : p.col_scale[(size_t)r_b[row] * p.col_ld + n];
                    v += r_nz[row] + bias;
                    if (p.addend) v += p.addend[(size_t)r_add[row] * p.Co + n];
                    v = (p.act == 1) ? tanhf(v) : (v > 0.f ? v : v * p.act_slope) * p.gain;
                    p.y[(size_t)pix * p.Co + n] = v;
                }
            }
        }
    }
}

template <int BM, int TM, int TN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, const PhaseArgs& P, epi_f32x16 (&acc)[TM][TN], unsigned char* smem_b,
                                              int m0, int n0, int wm, int wn, int tid, int l31, int lh, float alpha_mul = 1.f) {
    int* r_pix = reinterpret_cast<int*>(smem_b);
    int* r_b = r_pix + BM;
    float* r_nz = reinterpret_cast<float*>(r_b + BM);
    int* r_add = reinterpret_cast<int*>(r_nz + BM);
    if (tid < BM) {
        const int m = m0 + tid;
        int pix = -1, bb = 0, ap = 0;
        float nz = 0.f;
        const int bq = m / P.Mimg, pq = m - bq * P.Mimg;
        // rows past the end keep the sample index of the last valid row, so that r_b[BM-1] bounds the tile's samples
        bb = m < P.M ? bq : (P.M - 1) / P.Mimg;
        if (m < P.M && pq < P.HW) {
            const int gy = pq / P.Wg, gx = pq - gy * P.Wg;
            const int oy = gy * p.osy + P.oy0, ox = gx * p.osx + P.ox0;
            const int hw = oy * p.Wo + ox;
            pix = bb * p.Ho * p.Wo + hw;
            if (p.noise && p.noise_w) nz = p.noise_w[0] * p.noise[hw];
            ap = (bb * (p.Ho >> p.add_ups) + (oy >> p.add_ups)) * (p.Wo >> p.add_ups) + (ox >> p.add_ups);
        }
        r_pix[tid] = pix; r_b[tid] = bb; r_nz[tid] = nz; r_add[tid] = ap;
    }
    __syncthreads();
    conv_epilogue_apply<BM, TM, TN, WM, WN>(p, acc, smem_b, n0, wm, wn, l31, lh, alpha_mul);
}

// workgroup -> (phase, m-tile, n-tile); false = padding workgroup of a merged launch (exits)
__device__ __forceinline__ bool conv_tile_of_block(const ConvArgs& p, int ntn, int BN, int& phase, int& tm, int& n0) {
    phase = 0;
    if (p.nphase == 1) {
        // XCD-aware tile order: hardware sends workgroup b to XCD b % 8.  Give every XCD a contiguous range of the
        // (m-tile major, n-tile minor) tile list, so the n-tiles of one m-tile and its neighbouring image rows run on
        // the same XCD at the same time and share their activation rows in that XCD's L2.
        const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int qn = nb >> 3, rn = nb & 7;
        const int bid = xcd * qn + min(xcd, rn) + slot;
        tm = bid / ntn;
        n0 = (bid % ntn) * BN;
        return true;
    }
    // merged phases: every XCD gets an equal slice of EVERY phase's tile list (the phases differ in work per tile —
    // 4, 2, 2 and 1 taps — so slicing the concatenated list would leave some XCDs with only the heavy phase);
    // each phase's tile count is padded to a multiple of 8 and the padding workgroups exit.
    const int xcd = blockIdx.x & 7;
    int slot = blockIdx.x >> 3, t = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < p.nphase && t < 0) {
            const int c = p.ph[i].cnt8;
            if (slot < c) { t = xcd * c + slot; phase = i; } else slot -= c;
        }
    }
    if (t < 0 || t >= p.ph[phase].tiles) return false;
    tm = t / ntn;
    n0 = (t % ntn) * BN;
    return true;
}

}  // namespace wgsconv
