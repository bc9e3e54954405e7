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

using wgsconv::ConvArgs;
using wgsconv::PhaseArgs;

#ifndef WGS_ABL
#define WGS_ABL 0   // development ablations: 1 no split arithmetic, 2 no LDS stores, 3 no MFMA, 4 no global loads,
                    // 7 no style loads, 8 no weight loads, 9 no activation loads, 10 no LDS operand reads,
                    // 11 no 256-row tiles, 13 no split-K, 15 no LDS-DMA path, 16 no patch form
#endif

constexpr int BK = 32;          // fp32 values per K-chunk
constexpr int ROWB = 80;        // bytes per LDS row: 32 bf16 = 64 B + 16 B pad

typedef float f32x4 __attribute__((ext_vector_type(4)));

// split 4 floats into packed 16-bit hi (2 words) and lo (2 words) of the scheme; the casts lower to v_cvt_pk_* (RNE)
template <int SCH>
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
    if (WGS_ABL == 1) { hi = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)); lo = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w)); return; }
    const f32x4 f = {v.x, v.y, v.z, v.w};
    wgsconv::Scheme<SCH>::cvt4(f, hi, lo);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 buf_load4(const __amdgpu_buffer_rsrc_t r, int voff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

constexpr int OOB = (int)0x80000000;     // a byte offset beyond any buffer this kernel accepts (< 2 GiB): the load returns 0

// ASCALE: 0 no style, 1 one style vector per tile row (tiles that span several samples), 2 one per tile (every tile
// lies inside one sample: Hg*Wg is a multiple of BM) — the common case, and three fewer vector loads per thread/chunk.
template <int SCH, int BM, int BN, int WAVES_M, int WAVES_N, int ASCALE, bool UPS>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N == 4) ? 2 : 1) void igemm_nt16_kernel(const ConvArgs p) {
    typedef wgsconv::Scheme<SCH> SC;
    typedef typename SC::frag frag;
    constexpr int NA = SC::NA, NB = SC::NB;
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int CPR = BK / 4;       // float4 chunks per tile row (8)
    constexpr int RPP = NT / CPR;     // rows filled per pass (32 or 64)
    constexpr int PA = BM / RPP, PB = BN / RPP;
    constexpr int PS = ASCALE == 1 ? PA : 1;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int STAGE = NA * A_BYTES + NB * B_BYTES;   // A_hi (| A_lo) | B_hi (| B_lo)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ntn = (p.Co + BN - 1) / BN;
    // XCD-aware tile order: hardware sends workgroup b to XCD b % 8.  Give every XCD a contiguous range of the
    // (m-tile major, n-tile minor) tile list, so the n-tiles of one m-tile and its neighbouring image rows run on the
    // same XCD at the same time and share their activation rows in that XCD's L2 instead of each fetching them from HBM.
    int phase, tm, n0;
    if (!wgsconv::conv_tile_of_block(p, ntn, BN, phase, tm, n0)) return;
    const PhaseArgs& P = p.ph[phase];
    const int m0 = tm * BM;
    const int q = tid % CPR, r0 = tid / CPR;

    // All three operand streams go through buffer descriptors: 32-bit byte offsets (one VGPR per address, uniform
    // parts folded on the scalar unit) and hardware range checking — an out-of-image tap, a row past M or a column
    // past Co gets the offset OOB and reads as zero, so the loads are unconditional and need no masks afterwards.
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ASCALE ? p.a_scale : p.x), 0, ASCALE ? p.s_bytes : 0, 0x00020000);

    int a_iy0[PA], a_ix0[PA], a_off[PA], s_off[PS];
    if (ASCALE == 2) s_off[0] = ((m0 / P.Mimg) * p.a_ld + q * 4) * 4;
#pragma unroll
    for (int pa = 0; pa < PA; ++pa) {
        const int m = m0 + r0 + pa * RPP;
        // GEMM row m -> (sample b, pixel pix of the Hg x Wg grid); a sample owns Mimg >= Hg*Wg consecutive rows
        // (Mimg is rounded up to the tile height when that keeps every tile inside one sample, see launch_bf16x3)
        const int bq = m / P.Mimg, pq = m - bq * P.Mimg;
        const bool ok = m < P.M && pq < P.HW;
        const int b = ok ? bq : 0, pix = ok ? pq : 0;
        const int gy = pix / P.Wg, gx = pix - gy * P.Wg;
        a_iy0[pa] = ok ? gy * p.isy : -100000;
        a_ix0[pa] = gx * p.isx;
        // UPS: pixel index of the image origin; otherwise byte offset of (b, iy0, ix0, q*4) — taps add a uniform delta
        a_off[pa] = UPS ? b * p.Hi * p.Wi : ((b * p.Hi * p.Wi + gy * p.isy * p.Wi + gx * p.isx) * p.Ci + q * 4) * 4;
        if (ASCALE == 1) s_off[pa] = (b * p.a_ld + q * 4) * 4;
    }
    int b_off[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int n = n0 + r0 + pb * RPP;
        b_off[pb] = n < p.Co ? (int)((long)n * p.w_row_stride + q * 4) * 4 : OOB;
    }

    // Two activation staging register sets: chunk kt+2 is requested from HBM/L2 while chunk kt is multiplied and chunk
    // kt+1 (requested one iteration earlier) is split and written to LDS.  Weights and style vectors (L2/L1-resident,
    // shared by every tile) are fetched one iteration ahead in a single set, and are issued BEFORE the far activation
    // prefetch: vmcnt retires in order, so the wait in store_tile() leaves the chunk-(kt+2) loads in flight.
    struct Stage { float4 ra[PA]; };
    constexpr bool DEEP = (TM * TN <= 4);      // 64 accumulator registers: room for two activation staging sets
    Stage s0, s1;
    float4 rs[PS], rb[PB];
    const int cpt = p.Ci / BK;
    // split-K: workgroup (tile, blockIdx.y) contracts chunks [kbeg, kbeg + nk) of the ntaps*cpt chunk sequence
    const int nk_all = P.ntaps * cpt;
    const int kper = (nk_all + p.ksplit - 1) / p.ksplit;
    const int kbeg = (int)blockIdx.y * kper;
    const int nk = min(kper, nk_all - kbeg);
    const int Hup = p.Hi << p.ups, Wup = p.Wi << p.ups;
    // fp16 schemes: dynamic power-of-two operand scale (conv_scheme.h); undone on the accumulators
    float op_mult = 1.f, op_inv = 1.f;
    if (SCH != 0) wgsconv::operand_scale(p.a_amax, p.a_amax2, p.a_bound, op_mult, op_inv);

    // K order: channel chunk OUTER, tap INNER — consecutive iterations re-read the same pixels' channel chunk shifted
    // by one tap, so a tile's activation working set per chunk (~17 KB) stays in L1/L2 across the taps, and the
    // workgroups of an XCD walk the chunks roughly in step, sharing each weight chunk in that XCD's L2.  (Starting
    // every workgroup at a different chunk — WGS_ABL 5 — spreads L2 channels but makes the weights thrash: measured slower.)
    const int kofs = WGS_ABL == 5 ? (int)(blockIdx.x >> 3) % cpt : 0;
    // The main loop issues its loads unconditionally (straight-line code lets the compiler count vmcnt exactly and
    // keep the far prefetch in flight across the LDS store); past the last chunk the uniform offset becomes OOB.
    int tA = kbeg % P.ntaps, cA = (kbeg / P.ntaps + kofs) % cpt;      // (tap, chunk) cursors of the activation ...
    int tB = tA, cB = cA;                                              // ... and weight/style streams
    int nA = 0, nB = 0;                             // chunks requested so far
    // Per-chunk uniform state of the two streams (scalar registers), then one vector load per "piece":
    // pieces 0..PA-1 activation rows, PA..PA+PB-1 weight rows, PA+PB.. style vectors.
    int u_dy = 0, u_dx = 0, u_cbyteA = 0, u_delta = 0, u_cbyteB = 0, u_wdelta = 0;
    auto begin_tile = [&]() {
        const int yx = P.tap_yx[tA];
        u_dy = (int)(short)(yx & 0xffff); u_dx = yx >> 16;
        u_cbyteA = nA < nk ? cA * (BK * 4) : OOB;
        u_delta = UPS ? 0 : (int)((unsigned)P.tap_a[tA] + (unsigned)u_cbyteA);
        ++nA;
        if (++tA == P.ntaps) { tA = 0; if (++cA == cpt) cA = 0; }
    };
    auto begin_scale = [&]() {
        u_cbyteB = nB < nk ? cB * (BK * 4) : OOB;
        u_wdelta = (int)((unsigned)P.tap_w[tB] + (unsigned)u_cbyteB);
        ++nB;
        if (++tB == P.ntaps) { tB = 0; if (++cB == cpt) cB = 0; }
    };
    auto load_piece = [&](Stage& S, int idx) {
        if (idx < PA) {
            const int iy = a_iy0[idx] + u_dy, ix = a_ix0[idx] + u_dx;
            const bool v = (unsigned)iy < (unsigned)Hup && (unsigned)ix < (unsigned)Wup;
            int off;
            if (UPS) off = (int)((unsigned)(((a_off[idx] + (iy >> p.ups) * p.Wi + (ix >> p.ups)) * p.Ci + q * 4) * 4) + (unsigned)u_cbyteA);
            else off = (int)((unsigned)a_off[idx] + (unsigned)u_delta);
            if (WGS_ABL != 4 && WGS_ABL != 9) S.ra[idx] = buf_load4(rx, v ? off : OOB);
        } else if (idx < PA + PB) {
            if (WGS_ABL != 4 && WGS_ABL != 8) rb[idx - PA] = buf_load4(rw, (int)((unsigned)b_off[idx - PA] + (unsigned)u_wdelta));
        } else if (ASCALE && idx < PA + PB + PS) {
            if (WGS_ABL != 4 && WGS_ABL != 7) rs[idx - PA - PB] = buf_load4(rsc, (int)((unsigned)s_off[idx - PA - PB] + (unsigned)u_cbyteB));
        }
    };
    constexpr int NLOADS = PA + PB + (ASCALE ? PS : 0);
    auto load_tile = [&](Stage& S) {     // whole-chunk forms (prologue)
        begin_tile();
#pragma unroll
        for (int idx = 0; idx < PA; ++idx) load_piece(S, idx);
    };
    auto load_scale = [&]() {
        begin_scale();
#pragma unroll
        for (int idx = PA; idx < PA + PB + PS; ++idx) load_piece(s0, idx);
    };
    // one staged float4 (activation pieces 0..PA-1, then weight pieces PA..PA+PB-1): style multiply, hi/lo split, LDS
    auto store_piece = [&](int buf, const Stage& S, int idx) {
        unsigned char* base = smem_b + buf * STAGE;
        float4 v;
        int off;
        if (idx < PA) {
            v = S.ra[idx];
            // rounded fp32 product (no fma contraction into the split's residual): the LDS-DMA path's pre-pass does the same,
            // so the two forms stage identical bf16 pairs
            if (ASCALE) {
                const float4 sc = rs[ASCALE == 1 ? idx : 0];
                v.x = __fmul_rn(v.x, sc.x); v.y = __fmul_rn(v.y, sc.y); v.z = __fmul_rn(v.z, sc.z); v.w = __fmul_rn(v.w, sc.w);
                asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));   // keep the rounded product (no fma into the residual)
            }
            if (SCH != 0 && p.a_amax) { v.x *= op_mult; v.y *= op_mult; v.z *= op_mult; v.w *= op_mult; }
            off = (r0 + idx * RPP) * ROWB + q * 8;
        } else {
            v = rb[idx - PA];
            off = NA * A_BYTES + (r0 + (idx - PA) * RPP) * ROWB + q * 8;
        }
        uint2 hi, lo;
        split4<SCH>(v, hi, lo);
        if (WGS_ABL == 2) { asm volatile("" :: "v"(hi.x), "v"(hi.y), "v"(lo.x), "v"(lo.y)); return; }
        *reinterpret_cast<uint2*>(base + off) = hi;
        if ((idx < PA ? NA : NB) == 2) *reinterpret_cast<uint2*>(base + off + (idx < PA ? A_BYTES : B_BYTES)) = lo;
    };
    auto store_tile = [&](int buf, const Stage& S) {
#pragma unroll
        for (int idx = 0; idx < PA + PB; ++idx) store_piece(buf, S, idx);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    // One chunk of the main loop.  Multiplies LDS buffer `cur`; interleaved with the MFMAs of each (k-step, row-tile)
    // slot it (1) issues the vector loads of a later chunk — weight/style pieces first, then the activation pieces
    // into LD — and (2) splits + stores the already-landed staged chunk ST into LDS buffer `st`, so that neither the
    // address/VMEM-issue work nor the VALU/LDS-store work forms a phase of its own in which the matrix cores idle.
    // Loads go in the first half of the slots, stores in the second half (the weight/style registers are a single set,
    // filled and consumed within the chunk).  DEEP (4-wave tiles, two activation sets): ST was requested one chunk
    // earlier and LD is the other set; one-set 8-wave tiles: LD == ST.
    // The operand fragments of slot s+1 are read from LDS before the MFMAs of slot s are issued.
    constexpr int SLOTS = (BK / 16) * TM;
    constexpr int LSLOTS = SLOTS / 2;                                // slots that carry loads: [0, LSLOTS)
    constexpr int SFIRST = SLOTS / 2;                                // slots that carry stores: [SFIRST, SLOTS)
    constexpr int LPS = (NLOADS + LSLOTS - 1) / LSLOTS;              // loads per slot
    constexpr int PPS = (PA + PB + (SLOTS - SFIRST) - 1) / (SLOTS - SFIRST);    // store pieces per slot
    // load order: weights, styles, then activations (see the vmcnt note at the staging registers)
    auto load_order = [&](int n) { return n < PB + (ASCALE ? PS : 0) ? PA + n : n - PB - (ASCALE ? PS : 0); };
    auto mma_tile = [&](int cur, int st, Stage& LD, const Stage& ST) {
        const unsigned char* base = smem_b + cur * STAGE;
        const unsigned char* a_hi = base + (wm * WM + l31) * ROWB + lh * 16;
        const unsigned char* b_hi = base + NA * A_BYTES + (wn * WN + l31) * ROWB + lh * 16;
        auto read_a = [&](int slot, frag* f) {
            const int ks = slot / TM, i = slot % TM;
#pragma unroll
            for (int pl = 0; pl < NA; ++pl) {
                if (WGS_ABL == 10) { asm volatile("" : "=v"(f[pl])); continue; }
                f[pl] = *reinterpret_cast<const frag*>(a_hi + pl * A_BYTES + i * 32 * ROWB + ks * 32);
            }
        };
        auto read_b = [&](int ks, frag (*f)[NB]) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < NB; ++pl) {
                    if (WGS_ABL == 10) { asm volatile("" : "=v"(f[j][pl])); continue; }
                    f[j][pl] = *reinterpret_cast<const frag*>(b_hi + pl * B_BYTES + j * 32 * ROWB + ks * 32);
                }
        };
        frag bf[2][TN][NB], af[2][NA];
        read_b(0, bf[0]);
        read_a(0, af[0]);
        begin_scale();
        begin_tile();
#pragma unroll
        for (int slot = 0; slot < SLOTS; ++slot) {
            const int ks = slot / TM, i = slot % TM;
            if (slot + 1 < SLOTS) {
                read_a(slot + 1, af[(slot + 1) & 1]);
                if ((slot + 1) % TM == 0) read_b(ks + 1, bf[(ks + 1) & 1]);
            }
            if (slot < LSLOTS) {
#pragma unroll
                for (int u = 0; u < LPS; ++u) {
                    const int n = slot * LPS + u;
                    if (n < NLOADS) load_piece(LD, load_order(n));
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (WGS_ABL == 3) { asm volatile("" :: "v"(af[slot & 1][0]), "v"(bf[ks & 1][j][0])); continue; }
                acc[i][j] = SC::mma(af[slot & 1], bf[ks & 1][j], acc[i][j]);
            }
            if (slot >= SFIRST && st >= 0) {
#pragma unroll
                for (int u = 0; u < PPS; ++u) {
                    const int idx = (slot - SFIRST) * PPS + u;
                    if (idx < PA + PB) store_piece(st, ST, idx);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (DEEP) {
        load_tile(s0);
        load_scale();
        store_tile(0, s0);
        load_tile(s1);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            // even step: LDS[0] = chunk kt; s1 = activations of chunk kt+1 (in flight); request weights/styles of
            // chunk kt+1 and activations of chunk kt+2 -> s0; store chunk kt+1 into LDS[1]
            mma_tile(0, 1, s0, s1);
            __syncthreads();
            if (kt + 1 >= nk) break;
            mma_tile(1, 0, s1, s0);
            __syncthreads();
        }
    } else {
        load_tile(s0);
        load_scale();
        store_tile(0, s0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            mma_tile(cur, cur ^ 1, s0, s0);
            __syncthreads();
        }
    }

    if (p.ksplit > 1) {
        // split-K: raw partial tile -> ws[split][m][n]; conv_splitk_epilogue_kernel reduces and finishes
        float* part = p.ws + (size_t)blockIdx.y * P.M * p.Co;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + l31;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < P.M && n < p.Co) part[(size_t)m * p.Co + n] = acc[i][j][r] * op_inv;
                }
        }
        return;
    }
    wgsconv::conv_epilogue<BM, TM, TN, WM, WN>(p, P, acc, smem_b, m0, n0, wm, wn, tid, l31, lh, op_inv);
}

// Second pass of a split-K launch: y[pix(m)][n] = epilogue(sum_s ws[s][m][n]) — the same epilogue as above
// (alpha, demod, noise, bias, addend, activation).  One thread per 4 output channels.
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(const ConvArgs p) {
    const int c4 = p.Co / 4;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)p.M * c4) return;
    const int m = (int)(e / c4), n = (int)(e % c4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < p.ksplit; ++s) {
        const float4 t = *reinterpret_cast<const float4*>(p.ws + ((size_t)s * p.M + m) * p.Co + n);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const int b = m / p.Mimg, pq = m - b * p.Mimg;
    if (pq >= p.HW) return;            // padding row of a sample (see launch_bf16x3)
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
    }
    *reinterpret_cast<float4*>(p.y + ((size_t)b * p.Ho * p.Wo + hw) * p.Co + n) = make_float4(o[0], o[1], o[2], o[3]);
    if (p.y_amax) {     // magnitude bound for the next layer's fp16 operand scale
        const float am = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
        raise_amax(p.y_amax, am);
    }
}

// fill ph[0] of a single-phase launch from the main fields (after the row padding decision)
void single_phase(ConvArgs& a) {
    a.nphase = 1;
    PhaseArgs& P = a.ph[0];
    P.Wg = a.Wg; P.oy0 = a.oy0; P.ox0 = a.ox0; P.ntaps = a.ntaps; P.HW = a.HW; P.Mimg = a.Mimg; P.M = a.M; P.tiles = 0; P.cnt8 = 0;
    for (int t = 0; t < a.ntaps; ++t) { P.tap_yx[t] = a.tap_yx[t]; P.tap_a[t] = a.tap_a[t]; P.tap_w[t] = a.tap_w[t]; }
}

template <int SCH, int BM, int BN, int WAVES_M, int WAVES_N>
void launch_s(ConvArgs& a, hipStream_t st) {
    single_phase(a);
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.Co + BN - 1) / BN;
    const size_t sm = (size_t)2 * (wgsconv::Scheme<SCH>::NA * BM + wgsconv::Scheme<SCH>::NB * BN) * ROWB;
    dim3 grid((unsigned)(ntm * ntn), (unsigned)a.ksplit), block(64 * WAVES_M * WAVES_N);
    const int mode = !a.a_scale ? 0 : (a.Mimg % BM == 0 ? 2 : 1);
#define WGS_BF16_LAUNCH(AS, UP)                                                                             \
    {                                                                                                       \
        auto k = igemm_nt16_kernel<SCH, BM, BN, WAVES_M, WAVES_N, AS, UP>;                                  \
        wgs_note_kernel("igemm_nt16_kernel<%d, %d, %d, %d, %d, %d, %s>", SCH, BM, BN, WAVES_M, WAVES_N, AS, UP ? "true" : "false"); \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);     \
        WGS_LAUNCH(k, grid, block, sm, st, a);                                                      \
    }
    if (a.ups) {
        if (mode == 0) WGS_BF16_LAUNCH(0, true) else if (mode == 1) WGS_BF16_LAUNCH(1, true) else WGS_BF16_LAUNCH(2, true)
    } else {
        if (mode == 0) WGS_BF16_LAUNCH(0, false) else if (mode == 1) WGS_BF16_LAUNCH(1, false) else WGS_BF16_LAUNCH(2, false)
    }
#undef WGS_BF16_LAUNCH
}
template <int BM, int BN, int WAVES_M, int WAVES_N>
void launch(ConvArgs& a, hipStream_t st) {
    if (a.sch == 0) launch_s<0, BM, BN, WAVES_M, WAVES_N>(a, st);
    else if (a.sch == 1) launch_s<1, BM, BN, WAVES_M, WAVES_N>(a, st);
    else launch_s<2, BM, BN, WAVES_M, WAVES_N>(a, st);
}

// 8-wave 256-row tiles (one workgroup per CU): only the non-upsampling forms are instantiated
template <int SCH, int BM, int BN, int WAVES_M, int WAVES_N>
void launch_big_s(ConvArgs& a, hipStream_t st, int nblocks) {
    if (!nblocks) single_phase(a);
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.Co + BN - 1) / BN;
    const size_t sm = (size_t)2 * (wgsconv::Scheme<SCH>::NA * BM + wgsconv::Scheme<SCH>::NB * BN) * ROWB;
    dim3 grid((unsigned)(nblocks ? nblocks : ntm * ntn)), block(64 * WAVES_M * WAVES_N);
    if (!a.a_scale) {
        auto k = igemm_nt16_kernel<SCH, BM, BN, WAVES_M, WAVES_N, 0, false>;
        wgs_note_kernel("igemm_nt16_kernel<%d, %d, %d, %d, %d, 0, false>", SCH, BM, BN, WAVES_M, WAVES_N);
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        WGS_LAUNCH(k, grid, block, sm, st, a);
    } else {
        auto k = igemm_nt16_kernel<SCH, BM, BN, WAVES_M, WAVES_N, 2, false>;
        wgs_note_kernel("igemm_nt16_kernel<%d, %d, %d, %d, %d, 2, false>", SCH, BM, BN, WAVES_M, WAVES_N);
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        WGS_LAUNCH(k, grid, block, sm, st, a);
    }
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
    const long work = (long)a.M * (a.Co / 4);
    WGS_LAUNCH(conv_splitk_epilogue_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, a);
}

// operand extents for the buffer descriptors; every stream must be addressable with a 31-bit byte offset
static bool set_extents(ConvArgs& a, int wt_max) {
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

// The four sub-pixel phases of an up-conv (or any launches that differ only in grid geometry and taps) as ONE launch
// of the 8-wave kernel: 4x the workgroups per launch (short K loops: 1, 2, 2 and 4 taps) and one tail instead of four.
int launch_bf16x3_multi(const ConvArgs* as, int n, hipStream_t st) {
    if (n < 2 || n > 4) return 1;
    if (as[0].w_hi && wgs_flags().phase_patch) return 1;      // experiment: phases one by one through the patch form
    ConvArgs a = as[0];
    if (a.Ci % 32 != 0 || a.ups || a.Co % 128 != 0) return 1;
    int wt_max = 0, ntm_all = 0;
    for (int i = 0; i < n; ++i) {
        const ConvArgs& b = as[i];
        if (b.ntaps > 16) return 1;
        if (b.x != a.x || b.w != a.w || b.y != a.y || b.a_scale != a.a_scale || b.col_scale != a.col_scale || b.bias != a.bias ||
            b.noise != a.noise || b.noise_w != a.noise_w || b.addend != a.addend || b.B != a.B || b.Hi != a.Hi || b.Wi != a.Wi ||
            b.Ci != a.Ci || b.Ho != a.Ho || b.Wo != a.Wo || b.Co != a.Co || b.isy != a.isy || b.isx != a.isx || b.osy != a.osy ||
            b.osx != a.osx || b.ups != a.ups || b.add_ups != a.add_ups || b.act != a.act || b.alpha != a.alpha ||
            b.act_slope != a.act_slope || b.gain != a.gain || b.a_ld != a.a_ld || b.col_ld != a.col_ld ||
            b.w_tap_stride != a.w_tap_stride || b.w_row_stride != a.w_row_stride)
            return 1;
        for (int t = 0; t < b.ntaps; ++t) wt_max = b.wt[t] > wt_max ? b.wt[t] : wt_max;
        ConvArgs c = b;
        fill_tap_tables(c);
        PhaseArgs& P = a.ph[i];
        const int hw = c.Hg * c.Wg;
        const int mp = (hw + 255) / 256 * 256;
        if (a.a_scale && mp * 100L > hw * 113L) return 1;          // padding a sample's rows to 256 would cost > 13 %
        P.Wg = c.Wg; P.oy0 = c.oy0; P.ox0 = c.ox0; P.ntaps = c.ntaps; P.HW = hw;
        P.Mimg = a.a_scale ? mp : hw; P.M = c.B * P.Mimg;
        for (int t = 0; t < c.ntaps; ++t) { P.tap_yx[t] = c.tap_yx[t]; P.tap_a[t] = c.tap_a[t]; P.tap_w[t] = c.tap_w[t]; }
        ntm_all += (P.M + 255) / 256;
    }
    const int bn = a.Co % 256 == 0 ? 256 : 128, ntn = a.Co / bn;
    int blocks8 = 0;
    for (int i = 0; i < n; ++i) {
        a.ph[i].tiles = ((a.ph[i].M + 255) / 256) * ntn;
        a.ph[i].cnt8 = (a.ph[i].tiles + 7) / 8;
        blocks8 += a.ph[i].cnt8;
    }
    if (!set_extents(a, wt_max)) return 1;
    a.nphase = n; a.ksplit = 1;
    if (ntm_all * ntn < 200) return 1;
    if (try_dma(a, bn, blocks8 * 8, st)) return 0;
    if (bn == 256) launch_big<256, 256, 2, 4>(a, st, blocks8 * 8);
    else launch_big<256, 128, 4, 2>(a, st, blocks8 * 8);
    return 0;
}

int launch_bf16x3(const ConvArgs& a0, hipStream_t st) {
    if (a0.Ci % 32 != 0 || a0.ntaps > 16) return 1;
    ConvArgs a = a0;
    int wt_max = 0;
    for (int t = 0; t < a.ntaps; ++t) wt_max = a.wt[t] > wt_max ? a.wt[t] : wt_max;
    if (!set_extents(a, wt_max)) return 1;
    fill_tap_tables(a);
    if (a.a_hi) {
        // the caller's producer wrote the fp16 operand plane itself (wgs_conv_desc.x_f16): LDS-DMA kernel, no pre-pass
        const int ntm = (a.M + 255) / 256;
        const int bn = a.Co % 256 == 0 && ntm * (a.Co / 256) >= 200 ? 256 : 128;
        single_phase(a);
        a.w_bytes /= 2;                            // extent of the 16-bit weight planes (x_bytes is the fp16 plane's already)
        launch_dma_bf16x3(a, bn, ntm * (a.Co / bn), st);
        return 0;
    }
    // stride-1 3x3 convs with pre-split weights: the patch form stages the activation halo patch once per channel chunk
    if (WGS_ABL != 16 && !wgs_flags().no_patch && launch_patch_bf16x3(a, st) == 0) return 0;
    // Styled launches want every tile inside one sample (one style vector per tile, and the only form the 8-wave
    // tiles support).  When Hg*Wg is not a multiple of the tile height (the sub-pixel phases of the up-convs: 65x65,
    // 129x129 ...) each sample's row range is padded up to it, if that costs < 13 % extra rows.
    auto padded = [&](int bm) { const int mp = (a.HW + bm - 1) / bm * bm; return (!a.a_scale || mp * 100L <= a.HW * 113L) ? mp : 0; };
    auto use_rows = [&](int mimg) { a.Mimg = mimg; a.M = a.B * mimg; };
    const int mp256 = a.a_scale ? padded(256) : a.HW, mp128 = a.a_scale ? padded(128) : a.HW;
    const int ntm256 = mp256 ? (a.B * mp256 + 255) / 256 : 0;
    const bool big_ok = !a.ups && mp256 && WGS_ABL != 11;
    if (big_ok && a.Co % 128 == 0) {
        const int bn = a.Co % 256 == 0 && ntm256 * (a.Co / 256) >= 200 ? 256 : 128;
        if (ntm256 * (a.Co / bn) >= 200) {
            use_rows(mp256);
            single_phase(a);
            if (try_dma(a, bn, ntm256 * (a.Co / bn), st)) return 0;
            if (bn == 256) launch_big<256, 256, 2, 4>(a, st); else launch_big<256, 128, 4, 2>(a, st);
            return 0;
        }
    }
    if (mp128) use_rows(mp128);
    if (a.Co > 64) {
        // too few 128x128 tiles for the 256 CUs: split K (needs the caller's workspace and 4-channel rows)
        const int tiles = ((a.M + 127) / 128) * ((a.Co + 127) / 128);
        a.ksplit = WGS_ABL == 13 ? 1 : choose_ksplit(a, tiles, a.ntaps * (a.Ci / 32));
        launch<128, 128, 2, 2>(a, st);
        if (a.ksplit > 1) launch_splitk_epilogue(a, st);
    }
    else if (a.Co > 32) launch<128, 64, 2, 2>(a, st);
    else launch<128, 32, 4, 1>(a, st);
    return 0;
}

}  // namespace wgsconv
