// Shared epilogue of the split-bf16 implicit-GEMM kernels (same contract as conv_igemm.hip): per-row output addressing
// and the noise term are staged once in LDS, then every accumulator gets alpha, demodulation, noise, bias, addend and the
// activation and is stored as a 128-B channel run per lane group.
#pragma once
#include "conv_args.h"

namespace wgsconv {

typedef float epi_f32x16 __attribute__((ext_vector_type(16)));
template <bool V> struct EpiTag { static constexpr bool value = V; };

// Column statistics of a launch's output for the train-mode BatchNorm behind it (wgs_conv_desc.col_stats): a lane's partial sums of
// column n over its rows (<= 64 values, fp32) -> the two half-waves hold the same columns: combined with one cross-half shuffle -> one
// fp64 atomic per column and wave into replica blockIdx.x % nrep of the scratch [nrep][2][Co] (the layout of recon_ops.hip's chan_reduce).
__device__ __forceinline__ void col_stats_flush(double* ws, int Co, int n, bool nok, float s1, float s2, int lh) {
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lh == 0 && nok) {
        double* wr = ws + (size_t)(blockIdx.x % (unsigned)wgs_bn_nrep(Co)) * 2 * Co;
        unsafeAtomicAdd(wr + n, (double)s1);
        unsafeAtomicAdd(wr + Co + n, (double)s2);
    }
}

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
    // Fast path (every StyleGAN2 / ProgGAN layer launch of the 8-wave and patch tiles): leaky-relu epilogue, no addend, the
    // wave's columns all inside Co, one sample per tile.  Branch-free: rows past the end get an out-of-range offset that the
    // buffer store drops; per element = 2 mul + add + max-form leaky-relu + gain + one address add + one store.  (The general
    // path below costs ~4x the instructions per element, and for the short-K layers — Cin = 128: 288 MFMAs per wave —
    // the epilogue was a quarter of the kernel's instruction stream.)
    {
        const long ybytes = (long)p.B * p.Ho * p.Wo * p.Co * 4;
        const bool all_cols = n0 + wn * WN + TN * 32 <= p.Co;
        if (p.act == 0 && !p.addend && all_cols && (!p.col_scale || b_lo == b_hi2) && p.act_slope >= 0.f && p.act_slope <= 1.f &&
            ybytes < 0x7fffffffL) {
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)ybytes, 0x00020000);
            float cs[TN], bs[TN];
            int noff[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WN + j * 32 + l31;
                cs[j] = p.col_scale ? p.col_scale[(size_t)b_lo * p.col_ld + n] : 1.f;
                bs[j] = p.bias ? p.bias[n] : 0.f;
                noff[j] = n * 4;
            }
            const int rowbytes = p.Co * 4;
            const float slope = p.act_slope, gain = p.gain;
            float vmax = 0.f;
            // Two copies of the loop (with / without column statistics), and the running maximum pinned where it is computed: left to itself the
            // compiler SINKS the max / sum updates into the `if (p.y_amax)` / `if (stats)` blocks behind the loop, i.e. it keeps all TM * TN * 16
            // finished outputs alive to the end — 44 - 152 bytes of scratch per lane in every 256-row kernel (+16 % HBM writes in the counters,
            // rounds 3 - 5), reloaded one by one behind s_waitcnt vmcnt(0), which also waits for the tile's output stores.
            auto run = [&](auto stats_tag, auto amax_tag) {
                constexpr bool STATS = decltype(stats_tag)::value, AMAX = decltype(amax_tag)::value;
                float st1[TN], st2[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) { st1[j] = 0.f; st2[j] = 0.f; }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int pix = r_pix[row];
                        const float nz = r_nz[row];
                        const int ro = pix >= 0 ? pix * rowbytes : (int)0x80000000;
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            float v = acc[i][j][r] * alpha;
                            v *= cs[j];
                            v += nz + bs[j];
                            v = fmaxf(v, v * slope) * gain;           // == (v > 0 ? v : v * slope) * gain for slope in [0, 1]
                            if (AMAX) {
                                vmax = fmaxf(vmax, pix >= 0 ? fabsf(v) : 0.f);
                                asm volatile("" : "+v"(vmax));
                            }
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, (int)((unsigned)ro + (unsigned)noff[j]), 0, 0);
                            if (STATS) {
                                const float u = pix >= 0 ? v : 0.f;
                                st1[j] += u; st2[j] = fmaf(u, u, st2[j]);
                                asm volatile("" : "+v"(st1[j]), "+v"(st2[j]));
                            }
                        }
                    }
                }
                if (STATS) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) col_stats_flush(p.col_stats, p.Co, n0 + wn * WN + j * 32 + l31, true, st1[j], st2[j], lh);
                }
            };
            if (p.col_stats) { if (p.y_amax) run(EpiTag<true>{}, EpiTag<true>{}); else run(EpiTag<true>{}, EpiTag<false>{}); }
            else if (p.y_amax) run(EpiTag<false>{}, EpiTag<true>{});
            else run(EpiTag<false>{}, EpiTag<false>{});
            if (p.y_amax) {      // magnitude bound for the next layer's fp16 operand scale: one atomic per wave
                vmax = wave_max(vmax);
                if (l31 == 0 && lh == 0) raise_amax(p.y_amax, vmax);
            }
            return;
        }
    }
    float vmax_s = 0.f;
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
        float st1 = 0.f, st2 = 0.f;
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
                    vmax_s = fmaxf(vmax_s, fabsf(v));
                    p.y[(size_t)pix * p.Co + n] = v;
                    st1 += v; st2 = fmaf(v, v, st2);
                }
            }
        }
        if (p.col_stats) col_stats_flush(p.col_stats, p.Co, n, nok, st1, st2, lh);
    }
    if (p.y_amax) raise_amax(p.y_amax, vmax_s);
}

// Sum over the 16 lanes of a DPP row, on the vector ALU (quad permutes, then the half-row and row mirrors): every lane ends with the total.
// (First version of the epilogue below: __shfl_xor butterflies = 15 ds_bpermute per accumulator row; with three workgroups per CU the LDS
// pipe then carried 5 760 permutes per tile round and the kernel went from 780 to 1 064 us.)
__device__ __forceinline__ float dpp_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));    // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));    // row_mirror
    return v;
}

// The fast path above plus ToRGB (wgs_conv_desc.rgb_out): the workgroup's tile holds ALL output channels of its pixels (n0 = 0, BN = Co), so
// the three channel sums of a pixel are finished here — per accumulator row: 3 fused multiply-adds per column block, a butterfly over the 32
// 16-lane DPP rows of the half-wave (dpp_sum16), one LDS slot per DPP row; then the 2 WAVES_N partials are summed and stored as one 16-byte pixel.  y itself is
// stored only when the caller wants it (p.y): in a pass that keeps nothing, the layer's output never reaches HBM.
template <int BM, int TM, int TN, int WM, int WN, int WAVES_N>
__device__ __forceinline__ void conv_epilogue_rgb(const ConvArgs& p, epi_f32x16 (&acc)[TM][TN], unsigned char* smem_b, int wm, int wn, int l31, int lh,
                                                  int tid, float alpha_mul) {
    const float alpha = p.alpha * alpha_mul;
    const int* r_pix = reinterpret_cast<const int*>(smem_b);
    const int* r_b = r_pix + BM;
    const float* r_nz = reinterpret_cast<const float*>(r_b + BM);
    float* part = reinterpret_cast<float*>(smem_b) + 4 * BM;          // [2 WAVES_N][BM][4], behind the four row arrays
    const int b = r_b[0];
    const long ybytes = p.y ? (long)p.B * p.Ho * p.Wo * p.Co * 4 : 0;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y ? p.y : p.rgb_out, 0, (int)ybytes, 0x00020000);   // (y == NULL: stores dropped)
    float cs[TN], bs[TN], q0[TN], q1[TN], q2[TN];
    int noff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = wn * WN + j * 32 + l31;
        cs[j] = p.col_scale ? p.col_scale[(size_t)b * p.col_ld + n] : 1.f;
        bs[j] = p.bias ? p.bias[n] : 0.f;
        noff[j] = n * 4;
        const float sr = p.rgb_s[(size_t)b * p.rgb_ld + n] * p.rgb_scale;
        q0[j] = p.rgb_w[n] * sr; q1[j] = p.rgb_w[p.Co + n] * sr; q2[j] = p.rgb_w[2 * p.Co + n] * sr;
    }
    const int rowbytes = p.Co * 4;
    const float slope = p.act_slope, gain = p.gain;
    float vmax = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int pix = r_pix[row];
            const float nz = r_nz[row];
            const int ro = pix >= 0 ? pix * rowbytes : (int)0x80000000;
            float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v = acc[i][j][r] * alpha;
                v *= cs[j];
                v += nz + bs[j];
                v = fmaxf(v, v * slope) * gain;
                vmax = fmaxf(vmax, pix >= 0 ? fabsf(v) : 0.f);
                asm volatile("" : "+v"(vmax));          // (computed here, not sunk into `if (p.y_amax)` with every output kept alive: see conv_epilogue_apply)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, (int)((unsigned)ro + (unsigned)noff[j]), 0, 0);
                t0 = fmaf(v, q0[j], t0); t1 = fmaf(v, q1[j], t1); t2 = fmaf(v, q2[j], t2);
            }
            // the 32 lanes of a half-wave hold the row's 32 columns of each block: two DPP rows of 16, summed separately
            t0 = dpp_sum16(t0); t1 = dpp_sum16(t1); t2 = dpp_sum16(t2);
            if ((l31 & 15) == 0) *reinterpret_cast<float4*>(part + ((size_t)(wn * 2 + (l31 >> 4)) * BM + row) * 4) = make_float4(t0, t1, t2, 0.f);
        }
    }
    if (p.y_amax) {
        vmax = wave_max(vmax);
        if (l31 == 0 && lh == 0) raise_amax(p.y_amax, vmax);
    }
    __syncthreads();
    if (tid < BM) {
        float4 t = *reinterpret_cast<const float4*>(part + (size_t)tid * 4);
#pragma unroll
        for (int w2 = 1; w2 < 2 * WAVES_N; ++w2) {
            const float4 o = *reinterpret_cast<const float4*>(part + ((size_t)w2 * BM + tid) * 4);
            t.x += o.x; t.y += o.y; t.z += o.z;
        }
        const int pix = r_pix[tid];
        if (pix >= 0) *reinterpret_cast<float4*>(p.rgb_out + (size_t)pix * 4) = t;
    }
}

// first half of the epilogue: the staged row arrays of a tile of BM GEMM rows starting at m0 (published with a barrier)
template <int BM>
__device__ __forceinline__ void conv_epilogue_rows(const ConvArgs& p, const PhaseArgs& P, unsigned char* smem_b, int m0, int tid) {
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
}

template <int BM, int TM, int TN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, const PhaseArgs& P, epi_f32x16 (&acc)[TM][TN], unsigned char* smem_b,
                                              int m0, int n0, int wm, int wn, int tid, int l31, int lh, float alpha_mul = 1.f) {
    conv_epilogue_rows<BM>(p, P, smem_b, m0, tid);
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
