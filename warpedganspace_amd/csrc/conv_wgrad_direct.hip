// Weight gradient of a stride-1 3x3 'same' conv with MFMA operand fragments loaded STRAIGHT FROM GLOBAL MEMORY (round 5):
//     dW[co][t][ci] += sum over pixels m of dy[m][co] * x[m + shift_t][ci]        (torchvision BasicBlock convs of lib/reconstructor.py:52-79,
//                                                                                   restated at oracle/wgs_oracle.py:296-319)
// The staged kernels of conv_wgrad16.hip transpose both operands through LDS with one wave per SIMD: conversion, LDS stores (4-way bank
// conflicts, profiles/r4_wgrad16_pmc.txt), barrier and MFMAs take turns on every SIMD and the matrix pipe sits idle two thirds of the time
// (100-145 TFLOP/s).  Here no operand passes through LDS and the main loop has no barrier:
//
//   * GEMM view: rows = Cout, columns = Cin, K = pixels.  An MFMA A / B fragment wants, per lane, ONE channel and 8 consecutive k — and k
//     is the pixel index, so the 8 values of a lane are 8 pixel rows apart in memory while the 32 lanes of a half-wave read 32 adjacent
//     channels of the same pixel: every load instruction is two fully used 128 / 256-byte segments.  The transposition the old kernels do
//     in registers + LDS is just WHICH register a loaded value lands in.
//   * a wave owns 64 co x 32 ci x the THREE taps of one kernel row (6 accumulator blocks = 96 registers; two workgroups per CU, so that one
//     wave's conversion work runs under the other's MFMAs).  Lane r of a half-wave loads the dy channel PAIR (2r, 2r+1) of a pixel as one
//     8-byte load: the .x halves form the fragment of row block 0 (even channels), the .y halves that of block 1 (odd channels).  The
//     three taps read the same input row shifted by one pixel: 10 pixels per octet serve all three (fragment of tap q = pixels q .. q+7).
//     (TN = 2: 64 ci per wave with channel-pair loads on the x side too — 192 accumulator registers, one workgroup per CU; hipcc parks
//     half of them in AGPRs and moves them back and forth: kept as a template argument, not dispatched.)
//   * split-bf16 x3 (SCH 0: hi / lo of both operands, three v_mfma_f32_32x32x16_bf16 per product, as conv_wgrad16.hip) or exact fp32
//     (SCH 4: v_mfma_f32_32x32x2_f32, eight per product and octet pair, no conversion at all).
//   * the 4 waves of a workgroup take alternating k-steps of the workgroup's pixel range and add their accumulators through LDS once, at the
//     end, in a fixed order; workgroups of different pixel ranges write PARTIAL tiles to a workspace and a second launch adds them in split
//     order into dw: no atomics, bit-reproducible.
//
// Zero padding: a pixel octet never straddles image rows (Wo % 8 == 0); rows above / below the image and the pixel left / right of an
// image row get an out-of-range buffer offset (the load returns 0).
#include "wgs_common.h"
#include "conv_scheme.h"
#include "../../include/wgs.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int OOB = (int)0x80000000;

struct WDArgs {
    const float* x; const float* dy; float* part; float* dw;
    int Co, Ci, Ho, Wo, M, nsteps, ksplit, ntn, tiles;
    int x_bytes, dy_bytes;
    unsigned magic_w, magic_h;
    long w_tap_stride, w_row_stride;
    long part_stride;            // floats per pixel-range split in the workspace (= tiles * 9 * 64 * 64)
    int ntaps;                   // 9
    signed char dy_t[9];
    short wt[9];
};

#ifndef WGS_WDABL
#define WGS_WDABL 0          // development ablations: 1 no MFMA, 2 no conversion (garbage operands), 3 no loads
#endif

// partial-tile layout in the workspace: [split][tile][kernel row][tap q][i][jn][r][lane]  (a wave store = 256 contiguous bytes)
template <int TN>
__device__ __forceinline__ long part_index(int tile, int krow, int q, int i, int jn, int r, int lane) {
    return (((((long)tile * 3 + krow) * 3 + q) * 2 + i) * TN + jn) * 1024 + r * 64 + lane;
}

template <int SCH, int TN>
__global__ __launch_bounds__(256, TN == 1 ? 2 : 1) void wgrad_direct_kernel(const WDArgs p) {
    __shared__ float red[6 * TN * 16 * 64];                              // one wave's accumulators: [block][r 16][lane 64] floats (24 / 48 KB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, g = lane >> 5;
    // work item = (pixel-range split, kernel row, tile), tiles fastest: the items of one split read the same dy / x pixels.  Workgroups are
    // dealt to the 8 XCDs round-robin (each XCD has its own L2): XCD x takes a CONTIGUOUS range of items, so that a split's items share an L2
    int item;
    {
        const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int qn = nb >> 3, rn = nb & 7;
        item = xcd * qn + min(xcd, rn) + slot;
    }
    const int tile = item % p.tiles, krow = (item / p.tiles) % 3, split = item / (3 * p.tiles);
    const int co0 = (tile / p.ntn) * 64, ci0 = (tile % p.ntn) * (32 * TN);
    const int dyt = p.dy_t[krow * 3];
    const int per = (p.nsteps + p.ksplit - 1) / p.ksplit;
    const int s_begin = split * per, s_end = min(p.nsteps, s_begin + per);

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);

    f32x16 acc[3][2][TN];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][i][jn][r] = 0.f;

    // per-lane constant parts of the byte offsets: dy[(m + j) * Co + co0 + 2 l31 (+1)], x[(m + dyt * Wo - 1 + e) * Ci + ci0 + 2 l31 (+1)]
    const int a_lane = ((g * 8) * p.Co + co0 + 2 * l31) * 4;
    const int b_lane = ((g * 8 + dyt * p.Wo - 1) * p.Ci + ci0 + TN * l31) * 4;
    const int a_row = p.Co * 4, b_row = p.Ci * 4;

    struct Raw { f32x2 a[8]; f32x2 b[10]; };
    auto issue = [&](int s, Raw& w) {
        const int m = s * 16 + g * 8;                                  // first pixel of this half-wave's octet
        const int t = wgs_div_fast(m, p.magic_w);                      // image row index b * Ho + oy
        const int ox = m - t * p.Wo;
        const int oy = t - wgs_div_fast(t, p.magic_h) * p.Ho;
        const bool live = s < s_end;                                   // past the range: every offset out of range, the step adds zeros
        const bool rowok = live & ((unsigned)(oy + dyt) < (unsigned)p.Ho);      // (& not &&: selects, no branch inside the loop body)
        const int ab = live ? s * 16 * a_row + a_lane : OOB;
        const int bb = rowok ? s * 16 * b_row + b_lane : OOB;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (WGS_WDABL == 3) { asm volatile("" : "=v"(w.a[j])); continue; }
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rdy, ab + j * a_row, 0, 0);
            w.a[j] = f32x2{__uint_as_float(v.x), __uint_as_float(v.y)};
        }
#pragma unroll
        for (int e = 0; e < 10; ++e) {
            if (WGS_WDABL == 3) { asm volatile("" : "=v"(w.b[e])); continue; }
            int off = bb + e * b_row;
            if (e == 0) off = ox == 0 ? OOB : off;                     // the pixel left of the image row
            if (e == 9) off = ox + 8 == p.Wo ? OOB : off;              // ... and right of it
            if (TN == 2) {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rx, off, 0, 0);
                w.b[e] = f32x2{__uint_as_float(v.x), __uint_as_float(v.y)};
            } else {
                w.b[e] = f32x2{__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0)), 0.f};
            }
        }
    };

    auto consume = [&](const Raw& w) {
        if (SCH == 0) {
            // ---- split-bf16: hi = bf16_rn(v), lo = bf16_rn(v - hi); fragment = 8 consecutive pixels of one channel
            bf16x8 ah[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const f32x2 v = {i ? w.a[j].y : w.a[j].x, i ? w.a[j + 1].y : w.a[j + 1].x};
                    const bf16x2 h = __builtin_convertvector(v, bf16x2);
                    const bf16x2 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x2), bf16x2);
                    ah[i][j] = h.x; ah[i][j + 1] = h.y; al[i][j] = l.x; al[i][j + 1] = l.y;
                }
            __bf16 bh[TN][10], bl[TN][10];
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int e = 0; e < 10; e += 2) {
                    const f32x2 v = {jn ? w.b[e].y : w.b[e].x, jn ? w.b[e + 1].y : w.b[e + 1].x};
                    const bf16x2 h = __builtin_convertvector(v, bf16x2);
                    const bf16x2 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x2), bf16x2);
                    bh[jn][e] = h.x; bh[jn][e + 1] = h.y; bl[jn][e] = l.x; bl[jn][e + 1] = l.y;
                }
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    bf16x8 fh, fl;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { fh[j] = bh[jn][q + j]; fl[j] = bl[jn][q + j]; }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (WGS_WDABL == 1) { asm volatile("" :: "v"(fh), "v"(fl), "v"(ah[i]), "v"(al[i])); continue; }
                        f32x16 c = acc[q][i][jn];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], fh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], fl, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], fh, c, 0, 0, 0);
                        acc[q][i][jn] = c;
                    }
                }
        } else {
            // ---- exact fp32: MFMA e of a product contracts pixel e of the lower half-wave's octet with pixel e of the upper one's
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        f32x16 c = acc[q][i][jn];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float av = i ? w.a[e].y : w.a[e].x;
                            const float bv = jn ? w.b[q + e].y : w.b[q + e].x;
                            if (WGS_WDABL == 1) { asm volatile("" :: "v"(av), "v"(bv)); continue; }
                            c = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c, 0, 0, 0);
                        }
                        acc[q][i][jn] = c;
                    }
        }
    };

    // ---- main loop: this wave's k-steps s_begin + wave, + 4, ...; the loads of the next TWO steps (36 instructions, ~13 KB per wave) are in
    // flight while the current one multiplies: one step ahead left the loop latency-bound (Little: 8 waves x 6.5 KB per CU at ~2 us under
    // load = 6.6 TB/s chip-wide, measured 5.5).  Straight-line body (three steps per trip, no branch around the loads: hipcc's wait counts
    // stay exact — with the loads under an `if` it waited for all but 6 of the NEXT step's loads in front of every conversion block);
    // steps past the range load zeros.
    {
        Raw w0, w1, w2;
        int s = s_begin + wave;
        issue(s, w0);
        issue(s + 4, w1);
        __builtin_amdgcn_sched_barrier(0);       // (without the fences hipcc sinks a step's loads down to their first use: no prefetch left)
        const int trips = s < s_end ? (s_end - s + 11) / 12 : 0;
#pragma clang loop unroll(disable)
        for (int it = 0; it < trips; ++it, s += 12) {
            issue(s + 8, w2);
            __builtin_amdgcn_sched_barrier(0);
            consume(w0);
            __builtin_amdgcn_sched_barrier(0);
            issue(s + 12, w0);
            __builtin_amdgcn_sched_barrier(0);
            consume(w1);
            __builtin_amdgcn_sched_barrier(0);
            issue(s + 16, w1);
            __builtin_amdgcn_sched_barrier(0);
            consume(w2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- the four waves' accumulators, summed through LDS in the fixed order ((w3 + w2) + w1) + w0; wave 0 stores the partial tile
#pragma unroll 1
    for (int round = 3; round >= 1; --round) {
        if (wave == round) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                        for (int r = 0; r < 16; ++r) red[(((q * 2 + i) * TN + jn) * 16 + r) * 64 + lane] = acc[q][i][jn][r];
        }
        __syncthreads();
        if (wave == round - 1) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[q][i][jn][r] = red[(((q * 2 + i) * TN + jn) * 16 + r) * 64 + lane] + acc[q][i][jn][r];
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* part = p.part + (long)split * p.part_stride;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[part_index<TN>(tile, krow, q, i, jn, r, lane)] = acc[q][i][jn][r];
    }
}

// dw[co][t][ci] += sum over splits of the partial tiles, in split order (one thread per element of the partial layout)
template <int TN>
__global__ __launch_bounds__(256) void wgrad_direct_reduce_kernel(const WDArgs p, long n) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float s = 0.f;
    for (int k = 0; k < p.ksplit; ++k) s += p.part[(long)k * p.part_stride + e];
    const int lane = (int)(e & 63), r = (int)(e >> 6) & 15;
    int u = (int)(e >> 10);
    const int jn = u % TN; u /= TN;
    const int i = u & 1; u >>= 1;
    const int q = u % 3; u /= 3;
    const int krow = u % 3;
    const int tile = u / 3;
    const int co = (tile / p.ntn) * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) + i;
    const int ci = (tile % p.ntn) * (32 * TN) + TN * (lane & 31) + jn;
    float* d = p.dw + (long)co * p.w_row_stride + (long)p.wt[krow * 3 + q] * p.w_tap_stride + ci;
    *d += s;
}

}  // namespace

// Takes the launch (returns 0) when the shape is covered: stride-1 3x3 'same' conv, taps as kernel rows (dx = -1, 0, +1), Co % 64 == 0,
// Ci % 64 == 0, Wo % 8 == 0, pixels % 16 == 0, tensors < 2 GB, and a workspace that holds at least one split of partial tiles.
// precision 1: split-bf16 x3; 0: exact fp32.  Returns 1 when the shape is left to the staged kernels.
int wgs_conv_wgrad_direct(const wgs_wgrad_desc* d, hipStream_t st) {
    if (wgs_flags().wgrad_staged || !d->ws || d->x_s2d) return 1;
    if (d->ntaps != 9 || d->isx != 1 || d->isy != 1 || d->Hi != d->Ho || d->Wi != d->Wo) return 1;
    if (d->Co % 64 || d->Ci % 64 || d->Wo % 8) return 1;
    const long M = (long)d->B * d->Ho * d->Wo;
    if (M % 16 || M * d->Co * 4 >= 0x7fffffffL || M * d->Ci * 4 >= 0x7fffffffL) return 1;
    for (int t = 0; t < 9; t += 3)
        if (d->dy_t[t + 1] != d->dy_t[t] || d->dy_t[t + 2] != d->dy_t[t] || d->dx_t[t] != -1 || d->dx_t[t + 1] != 0 || d->dx_t[t + 2] != 1 ||
            d->dy_t[t] < -1 || d->dy_t[t] > 1) return 1;
    WDArgs a;
    a.x = d->x; a.dy = d->dy; a.dw = d->dw; a.part = (float*)d->ws;
    a.Co = d->Co; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.M = (int)M; a.nsteps = (int)(M / 16);
    constexpr int TN = 1;
    a.ntn = d->Ci / (32 * TN);
    a.x_bytes = (int)(M * d->Ci * 4); a.dy_bytes = (int)(M * d->Co * 4);
    a.magic_w = wgs_div_magic(d->Wo);
    a.magic_h = d->Ho >= 2 ? wgs_div_magic(d->Ho) : 0u;
    if (d->Ho < 2 || (M + 64) * d->Wo >= (1L << 32) || ((long)d->B * d->Ho + 64) * d->Ho >= (1L << 32)) return 1;
    a.w_tap_stride = d->w_tap_stride; a.w_row_stride = d->w_row_stride;
    a.ntaps = 9;
    for (int t = 0; t < 9; ++t) { a.dy_t[t] = d->dy_t[t]; a.wt[t] = d->wt[t]; }
    const int tiles = (d->Co / 64) * a.ntn;
    a.tiles = tiles;
    a.part_stride = (long)tiles * 9 * 2048 * TN;
    const long cap = d->ws_bytes / (a.part_stride * 4);
    if (cap < 1) return 1;
    // pixel-range splits (tools/bench_wgrad_direct.py ksweep, round 5): one round of workgroups — about 384 of the 512 resident slots (two per
    // CU) — is best on every layer shape of the Reconstructor at 256^2 inputs (more splits only grow the partial-tile traffic of the second
    // launch); launches with a lot of work per wave (1024^2 inputs) take twice or four times as many while the grid stays <= 1024
    int ks = d->ksplit;
    if (ks <= 0) {
        ks = 384 / (tiles * 3);
        if (ks < 1) ks = 1;
        while (a.nsteps / (ks * 4) > 48 && (long)tiles * 3 * ks * 2 <= 1024) ks *= 2;
        if (ks > a.nsteps / 16) ks = a.nsteps / 16;
    }
    if (ks > cap) ks = (int)cap;
    if (ks > a.nsteps) ks = a.nsteps;
    if (ks < 1) ks = 1;
    a.ksplit = ks;
    dim3 grid((unsigned)(tiles * 3 * ks));
    if (d->precision == 1) {
        auto k = wgrad_direct_kernel<0, TN>;
        wgs_note_kernel("wgrad_direct_kernel<0>");
        WGS_LAUNCH(k, grid, dim3(256), 0, st, a);
    } else {
        auto k = wgrad_direct_kernel<4, TN>;
        wgs_note_kernel("wgrad_direct_kernel<4>");
        WGS_LAUNCH(k, grid, dim3(256), 0, st, a);
    }
    const long n = a.part_stride;
    WGS_LAUNCH(wgrad_direct_reduce_kernel<TN>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, n);
    return 0;
}
