// Split-bf16 implicit-GEMM convolution, LDS-DMA form: both operands arrive ALREADY split into bf16 hi / lo planes and are
// copied HBM/L2 -> LDS by `buffer_load_dwordx4 ... lds` — no staging registers, no split arithmetic and no ds_write in the
// main loop, which is then only {issue DMA of chunk k+1, ds_read fragments, 3 MFMAs per product block}.
//
//   * weights: split once for the frozen generator (wgs_split_bf16; wgs_conv_desc.w_hi / w_lo, same packed layout as w)
//   * activations: split (and style-modulated: the style multiply must precede the split) by modsplit_kernel below into the
//     caller's workspace — one extra read + write of the activation tensor per launch, paid for by the conv kernel
//     (measured: register-staged split + ds_write cost 24-33 % of the conv time on the 8-wave tiles).
//
// LDS image per stage: A_hi | A_lo [BM][32 bf16 = 64 B], B_hi | B_lo [BN][64 B], rows UNPADDED because an LDS-DMA writes
// wave-uniform base + lane*16: one instruction fills 16 rows x 64 B.  Bank conflicts of the per-lane ds_read_b128 (lane =
// row) are avoided by an XOR swizzle of the four 16-B chunks of a row with (row >> 2) & 3, applied on the SOURCE address of
// the DMA (lane -> row = lane/4, slot = lane%4 fetches logical chunk slot ^ swz) and on the ds_read address.
//
// Template parameter SCH: the same kernel for the fp16 schemes (conv_scheme.h) — one activation plane, one or two weight planes.
//
// Pipeline: a ring of NS stages (as many as fit in LDS: 4 for the fp16 forms, 3 / 2 for the two-plane forms).  EVERY vector
// memory operation of the main loop is an LDS-DMA, so — unlike the kernels that mix DMAs with VGPR loads (conv_igemm_patch.hip)
// — a COUNTED wait is legal: at the top of step k a wave waits until all but its newest NS - 2 chunks have landed
// (s_waitcnt vmcnt((NS - 2) * NPIECE)), the barrier then publishes chunk k of all waves and retires stage (k - 1) % NS, whose
// refill with chunk k + NS - 1 is issued between the first MFMA slots.  A chunk has NS - 1 steps of flight instead of one.
#ifndef WGS_DABL
#define WGS_DABL 0   // development ablations (tools/build_abl.sh dabl 1 2): 1 = two stages, everything drained at every
                     // barrier; 2 = ring without the ping-pong schedule
#endif
//
// Schedule of the single-plane fp16 form (8 waves = two per SIMD): PING-PONG.  A wave alternates a memory phase (all 12
// operand fragments of a chunk LDS -> registers, the DMAs of a later chunk) with a compute phase (the chunk's 16 MFMAs from
// registers), one barrier after each; waves 4-7 run one phase behind waves 0-3, so on every SIMD one wave multiplies while
// the other fetches, and the matrix pipe goes from one wave's MFMA block straight into the other's.  (With all eight waves in
// phase, every barrier was followed by eight waves waiting on their first fragment reads.)
#include "wgs_common.h"
#include "conv_args.h"
#include "conv_epilogue.h"
#include "conv_scheme.h"

typedef wgsconv::epi_f32x16 f32x16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

using wgsconv::ConvArgs;
using wgsconv::PhaseArgs;

constexpr int BK = 32;                 // K values per chunk
constexpr int ROW = 64;                // bytes per LDS row and plane (32 bf16)
constexpr int OOB = (int)0x80000000;   // byte offset beyond any buffer this kernel accepts: the DMA writes zeros

typedef __attribute__((address_space(3))) unsigned char lds_byte;

constexpr int dma_stages(int stage_bytes) { return 160 * 1024 / stage_bytes >= 4 ? 4 : (160 * 1024 / stage_bytes >= 3 ? 3 : 2); }

// RGB: ToRGB in the epilogue (wgs_conv_desc.rgb_out; the tile's BN columns are all of Cout) — its own instantiation
template <int SCH, int BM, int BN, int WAVES_M, int WAVES_N, bool RGB = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 1) void igemm_dma16_kernel(const ConvArgs p) {
    typedef wgsconv::Scheme<SCH> SC;
    typedef typename SC::frag frag;
    constexpr int NA = SC::NA, NB = SC::NB;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
    constexpr int A_BYTES = BM * ROW, B_BYTES = BN * ROW;
    constexpr int STAGE = NA * A_BYTES + NB * B_BYTES;
    constexpr int NS = WGS_DABL == 1 ? 2 : dma_stages(STAGE);
    constexpr int AI = BM / 16 / NW, BI = BN / 16 / NW;     // 16-row DMA instructions per wave and plane
    static_assert(AI >= 1 && BI >= 1 && AI * 16 * NW == BM && BI * 16 * NW == BN, "tile / wave count mismatch");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ntn = (p.Co + BN - 1) / BN;
    int phase, tm, n0;
    if (!wgsconv::conv_tile_of_block(p, ntn, BN, phase, tm, n0)) return;
    const PhaseArgs& P = p.ph[phase];
    const int m0 = tm * BM;

    const __amdgpu_buffer_rsrc_t rah = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.a_hi), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ral = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.a_lo), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbh = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w_hi), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbl = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w_lo), 0, p.w_bytes, 0x00020000);

    // DMA source side: this lane feeds LDS row (16*instr + lane/4), slot lane%4 of every instruction of its wave
    const int lrow = lane >> 2, slot = lane & 3;
    int a_iy0[AI], a_ix0[AI], a_off[AI], b_off[BI];
#pragma unroll
    for (int j = 0; j < AI; ++j) {
        const int row = (wave * AI + j) * 16 + lrow;
        const int m = m0 + row;
        const int bq = m / P.Mimg, pq = m - bq * P.Mimg;
        const bool ok = m < P.M && pq < P.HW;
        const int b = ok ? bq : 0, pix = ok ? pq : 0;
        const int gy = pix / P.Wg, gx = pix - gy * P.Wg;
        a_iy0[j] = ok ? gy * p.isy : -100000;
        a_ix0[j] = gx * p.isx;
        const int lc = slot ^ ((row >> 2) & 3);
        a_off[j] = ((b * p.Hi * p.Wi + gy * p.isy * p.Wi + gx * p.isx) * p.Ci + lc * 8) * 2;      // bytes in a bf16 plane
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int row = (wave * BI + j) * 16 + lrow;
        const int n = n0 + row;
        const int lc = slot ^ ((row >> 2) & 3);
        b_off[j] = n < p.Co ? (int)((long)n * p.w_row_stride + lc * 8) * 2 : OOB;
    }

    const int cpt = p.Ci / BK;
    const int nk = P.ntaps * cpt;
    // K order: channel chunk outer, tap inner (see conv_igemm_bf16.hip)
    int tC = 0, cC = 0, nC = 0;
    // DMA of the next chunk, as 2*(AI+BI) pieces (activation hi/lo per 16-row group, then weight hi/lo) so that the
    // issue can be spread between the MFMA slots; past the last chunk everything is OOB (zeros, no memory traffic)
    constexpr int NPIECE = NA * AI + NB * BI;
    int u_dy = 0, u_dx = 0;
    unsigned u_adelta = 0, u_bdelta = 0;
    auto begin_chunk = [&]() {
        const int yx = P.tap_yx[tC];
        u_dy = (int)(short)(yx & 0xffff); u_dx = yx >> 16;
        const int cbyte = nC < nk ? cC * (BK * 2) : OOB;
        // tap_a / tap_w are byte offsets for fp32 elements; unsigned adds: the sums may carry the OOB marker
        u_adelta = (unsigned)(P.tap_a[tC] >> 1) + (unsigned)cbyte;
        u_bdelta = (unsigned)(P.tap_w[tC] >> 1) + (unsigned)cbyte;
        ++nC;
        if (++tC == P.ntaps) { tC = 0; if (++cC == cpt) cC = 0; }
    };
    auto issue_piece = [&](int buf, int idx) {
        lds_byte* st = (lds_byte*)(smem_b + buf * STAGE);
        if (idx < NA * AI) {
            const int j = idx / NA, pl = idx % NA;
            const int iy = a_iy0[j] + u_dy, ix = a_ix0[j] + u_dx;
            const bool v = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const int off = v ? (int)((unsigned)a_off[j] + u_adelta) : OOB;
            lds_byte* d = st + (wave * AI + j) * 16 * ROW + pl * A_BYTES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(pl ? ral : rah, d, 16, off, 0, 0, 0);
        } else {
            const int j = (idx - NA * AI) / NB, pl = (idx - NA * AI) % NB;
            const int off = (int)((unsigned)b_off[j] + u_bdelta);
            lds_byte* d = st + NA * A_BYTES + (wave * BI + j) * 16 * ROW + pl * B_BYTES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(pl ? rbl : rbh, d, 16, off, 0, 0, 0);
        }
    };
    auto issue = [&](int buf) {
        begin_chunk();
#pragma unroll
        for (int idx = 0; idx < NPIECE; ++idx) issue_piece(buf, idx);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 2) & 3;
    const int kc0 = ((0 + lh) ^ swz) * 16, kc1 = ((2 + lh) ^ swz) * 16;       // byte offset of this lane's 8 k-values, k-step 0 / 1
    const int a_rd = (wm * WM + l31) * ROW, b_rd = NA * A_BYTES + (wn * WN + l31) * ROW;
    float op_mult = 1.f, op_inv = 1.f;       // fp16 schemes: the pre-pass scaled the activations by op_mult (conv_scheme.h)
    if (SCH != 0) wgsconv::operand_scale(p.a_amax, p.a_amax2, p.a_bound, op_mult, op_inv);

    // Multiply stage `cur` while the DMA of the next chunk is issued into stage `nxt` between the first MFMA slots; the
    // operand fragments of slot s+1 are read from LDS before the MFMAs of slot s.
    constexpr int SLOTS = (BK / 16) * TM;
    constexpr int LSLOTS = SLOTS / 2;
    constexpr int LPS = (NPIECE + LSLOTS - 1) / LSLOTS;
    auto mma_tile = [&](int cur, int nxt) {
        const unsigned char* base = smem_b + cur * STAGE;
        auto read_a = [&](int slot, frag* f) {
            const int i = slot % TM, kc = (slot / TM) ? kc1 : kc0;
#pragma unroll
            for (int pl = 0; pl < NA; ++pl) f[pl] = *reinterpret_cast<const frag*>(base + a_rd + pl * A_BYTES + i * 32 * ROW + kc);
        };
        auto read_b = [&](int ks, frag (*f)[NB]) {
            const int kc = ks ? kc1 : kc0;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < NB; ++pl) f[j][pl] = *reinterpret_cast<const frag*>(base + b_rd + pl * B_BYTES + j * 32 * ROW + kc);
        };
        frag bf[2][TN][NB], af[2][NA];
        read_b(0, bf[0]);
        read_a(0, af[0]);
        begin_chunk();
#pragma unroll
        for (int slot = 0; slot < SLOTS; ++slot) {
            const int ks = slot / TM, i = slot % TM;
            if (slot + 1 < SLOTS) {
                read_a(slot + 1, af[(slot + 1) & 1]);
                if ((slot + 1) % TM == 0) read_b(ks + 1, bf[(ks + 1) & 1]);
            }
            if (slot < LSLOTS) {
#pragma unroll
                for (int u = 0; u < LPS; ++u)
                    if (slot * LPS + u < NPIECE) issue_piece(nxt, slot * LPS + u);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = SC::mma(af[slot & 1], bf[ks & 1][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

#pragma unroll
    for (int st = 0; st < NS - 1; ++st) issue(st);      // chunks 0 .. NS-2 (past the end: zeros)
    int cur = 0, nxt = NS - 1;
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");    // (the bare s_barrier does not order memory operations for the compiler)
        __builtin_amdgcn_sched_barrier(0);
    };
    constexpr bool PP = SCH == 1 && NW == 8 && NS >= 3 && WGS_DABL == 0;
    if (PP) {
        const int grp = wave >> 2;
        frag af[2][TM][NA], bf[2][TN][NB];
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * NPIECE) : "memory");   // own pieces of chunk 0
        bar();                                   // chunk 0 of all waves
        if (grp == 1) bar();                     // waves 4-7: one phase behind
        for (int kt = 0; kt < nk; ++kt) {
            // memory phase: fragments of chunk kt, DMAs of chunk kt+NS-1 into the stage chunk kt-1 was read from (by waves 0-3 two
            // phases ago, by waves 4-7 one phase ago: both behind a barrier)
            const unsigned char* base = smem_b + cur * STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int kc = ks ? kc1 : kc0;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pl = 0; pl < NB; ++pl) bf[ks][j][pl] = *reinterpret_cast<const frag*>(base + b_rd + pl * B_BYTES + j * 32 * ROW + kc);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pl = 0; pl < NA; ++pl) af[ks][i][pl] = *reinterpret_cast<const frag*>(base + a_rd + pl * A_BYTES + i * 32 * ROW + kc);
            }
            begin_chunk();
#pragma unroll
            for (int idx = 0; idx < NPIECE; ++idx) issue_piece(nxt, idx);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * NPIECE) : "memory");   // own pieces of chunk kt+1 have landed
            bar();
            // compute phase
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = SC::mma(af[ks][i], bf[ks][j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            bar();
            cur = cur + 1 == NS ? 0 : cur + 1;
            nxt = nxt + 1 == NS ? 0 : nxt + 1;
        }
        if (grp == 0) bar();
    } else {
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * NPIECE) : "memory");   // this wave's pieces of chunk kt have landed
        bar();                            // ... and everybody's; stage `nxt` (chunk kt-1) is no longer read
        mma_tile(cur, nxt);               // chunk kt+NS-1 is issued into `nxt` while chunk kt is multiplied
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        cur = cur + 1 == NS ? 0 : cur + 1;
        nxt = nxt + 1 == NS ? 0 : nxt + 1;
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the zero-fill DMAs past the last chunk: the epilogue re-uses the LDS
    __syncthreads();
    if constexpr (RGB) {
        wgsconv::conv_epilogue_rows<BM>(p, P, smem_b, m0, tid);
        wgsconv::conv_epilogue_rgb<BM, TM, TN, WM, WN, WAVES_N>(p, acc, smem_b, wm, wn, l31, lh, tid, op_inv);
        return;
    }
    wgsconv::conv_epilogue<BM, TM, TN, WM, WN>(p, P, acc, smem_b, m0, n0, wm, wn, tid, l31, lh, op_inv);
}

// fp16 schemes: hi = f16_rn(v * mult), lo = f16_rn(residual) (lo may be null) of v = x * style; mult: conv_scheme.h
__global__ __launch_bounds__(256) void modcvt_f16_kernel(const float* __restrict__ x, const float* __restrict__ s, int s_ld,
                                                         unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                                         long per_sample4, int c4n, long total4, const float* a_amax, const float* a_amax2,
                                                         float a_bound) {
    float mult, inv;
    wgsconv::operand_scale(a_amax, a_amax2, a_bound, mult, inv);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(x)[e];
        if (s) {
            const long b = e / per_sample4;
            const int c = (int)(e % c4n) * 4;
            const float4 sc = *reinterpret_cast<const float4*>(s + b * s_ld + c);
            v.x = __fmul_rn(v.x, sc.x); v.y = __fmul_rn(v.y, sc.y); v.z = __fmul_rn(v.z, sc.z); v.w = __fmul_rn(v.w, sc.w);
        }
        const f32x4 f = {v.x * mult, v.y * mult, v.z * mult, v.w * mult};
        uint2 h, l;
        if (lo) { wgsconv::Scheme<2>::cvt4(f, h, l); reinterpret_cast<uint2*>(lo)[e] = l; }
        else wgsconv::Scheme<1>::cvt4(f, h, l);
        reinterpret_cast<uint2*>(hi)[e] = h;
    }
}

// hi = bf16_rn(v), lo = bf16_rn(v - hi) of v = x * style (style per sample and channel, or none)
__global__ __launch_bounds__(256) void modsplit_kernel(const float* __restrict__ x, const float* __restrict__ s, int s_ld,
                                                       unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                                       long per_sample4, int c4n, long total4) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(x)[e];
        if (s) {
            const long b = e / per_sample4;
            const int c = (int)(e % c4n) * 4;
            const float4 sc = *reinterpret_cast<const float4*>(s + b * s_ld + c);
            v.x = __fmul_rn(v.x, sc.x); v.y = __fmul_rn(v.y, sc.y); v.z = __fmul_rn(v.z, sc.z); v.w = __fmul_rn(v.w, sc.w);
                asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));   // keep the rounded product (no fma into the residual)
        }
        const f32x4 f = {v.x, v.y, v.z, v.w};
        const bf16x4 h = __builtin_convertvector(f, bf16x4);
        const f32x4 r = f - __builtin_convertvector(h, f32x4);
        const bf16x4 l = __builtin_convertvector(r, bf16x4);
        reinterpret_cast<uint2*>(hi)[e] = __builtin_bit_cast(uint2, h);
        reinterpret_cast<uint2*>(lo)[e] = __builtin_bit_cast(uint2, l);
    }
}

template <int SCH, int BM, int BN, int WAVES_M, int WAVES_N>
void launch_dma_s(const ConvArgs& a, hipStream_t st, int nblocks) {
    const size_t stage = (size_t)(wgsconv::Scheme<SCH>::NA * BM + wgsconv::Scheme<SCH>::NB * BN) * ROW;
    const size_t sm = (WGS_DABL == 1 ? 2 : dma_stages((int)stage)) * stage;
    auto k = igemm_dma16_kernel<SCH, BM, BN, WAVES_M, WAVES_N>;
    wgs_note_kernel("igemm_dma16_kernel<%d, %d, %d, %d, %d>", SCH, BM, BN, WAVES_M, WAVES_N);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    WGS_LAUNCH(k, dim3((unsigned)nblocks), dim3(64 * WAVES_M * WAVES_N), sm, st, a);
}
template <int BM, int BN, int WAVES_M, int WAVES_N>
void launch_dma(const ConvArgs& a, hipStream_t st, int nblocks) {
    if (a.sch == 0) launch_dma_s<0, BM, BN, WAVES_M, WAVES_N>(a, st, nblocks);
    else if (a.sch == 1) launch_dma_s<1, BM, BN, WAVES_M, WAVES_N>(a, st, nblocks);
    else launch_dma_s<2, BM, BN, WAVES_M, WAVES_N>(a, st, nblocks);
}

}  // namespace

namespace wgsconv {

void split_bf16(const float* x, const float* s, int s_ld, unsigned short* hi, unsigned short* lo, long nsamples, long per_sample,
                int C, hipStream_t st) {
    const long total4 = nsamples * per_sample / 4;
    long grid = (total4 + 255) / 256;
    if (grid > 16384) grid = 16384;
    WGS_LAUNCH(modsplit_kernel, dim3((unsigned)grid), dim3(256), 0, st, x, s, s_ld, hi, lo, per_sample / 4, C / 4, total4);
}

void split_f16(const float* x, const float* s, int s_ld, unsigned short* hi, unsigned short* lo, long nsamples, long per_sample,
               int C, const float* a_amax, const float* a_amax2, float a_bound, hipStream_t st) {
    const long total4 = nsamples * per_sample / 4;
    long grid = (total4 + 255) / 256;
    if (grid > 16384) grid = 16384;
    WGS_LAUNCH(modcvt_f16_kernel, dim3((unsigned)grid), dim3(256), 0, st, x, s, s_ld, hi, lo, per_sample / 4, C / 4, total4, a_amax, a_amax2, a_bound);
}

// a: fully prepared arguments (phases filled, a_hi/a_lo/w_hi/w_lo and extents set); bn = 256 or 128
void launch_dma_bf16x3(const ConvArgs& a, int bn, int nblocks, hipStream_t st) {
    if (a.rgb_out && a.sch == 1 && bn == 256 && a.Co == 256) {       // ToRGB in the epilogue: one 256-column tile holds all of Cout
        constexpr int BM = 256, BN = 256;
        const size_t stage = (size_t)(BM + BN) * ROW;
        const size_t sm = dma_stages((int)stage) * stage;
        auto k = igemm_dma16_kernel<1, BM, BN, 2, 4, true>;
        wgs_note_kernel("igemm_dma16_kernel<1, 256, 256, 2, 4, true>");
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        WGS_LAUNCH(k, dim3((unsigned)nblocks), dim3(512), sm, st, a);
        return;
    }
    if (bn == 256) launch_dma<256, 256, 2, 4>(a, st, nblocks);
    else launch_dma<256, 128, 4, 2>(a, st, nblocks);
}

}  // namespace wgsconv
