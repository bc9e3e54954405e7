// Plain-fp16 3x3 stride-1 convolution of a producer-written fp16 activation plane (wgs_conv_desc.x_f16), PATCH form with EVERY
// operand copied global -> LDS by DMA: the 128-pixel x 128-column tile of conv_igemm_patch.hip (igemm_patch_kernel<1, 128, 128, 2, 2,
// 1, 0, true>: StyleGAN2's 128 -> 128 layers at 256^2), re-cut so that its main loop issues nothing but LDS-DMA loads.
//
// Why: that kernel loads the next chunk's patch into REGISTERS (to write it to a padded LDS image) next to the weight DMAs, and
// on gfx950 register loads and LDS-DMA loads do not retire in order relative to each other — so it can only drain everything
// (s_waitcnt vmcnt(0)) at the end of every one-tap step, i.e. a step lasts one L2 round trip (~0.8 us) for 8 MFMAs (0.1 us) and
// only the three resident workgroups hide it.  Round-5 ablations of that kernel at 128 -> 128 @256^2 B = 32 (0.758 ms): without
// the weight DMAs 0.580, without the patch loads 0.551, without the MFMAs 0.506, without the barriers 0.742 ms: latency, not work.
//
// Here the patch is a 64-byte-row LDS image as well (the fp16 plane needs no conversion), XOR-swizzled like the weight stages:
//   LDS pixel p, 16-byte slot s  holds  channels 8 * (s ^ ((p >> 2) & 3)) .. + 7  of patch pixel p       (p = pr * 18 + pc)
// 16 consecutive pixels x one piece cover all 64 banks once for ANY start pixel (p mod 16 -> (p % 4, piece ^ (p >> 2 & 3)) is a
// bijection), so a tap's fragment read (16 pixels of one tile row per 16 lanes, start pixel shifted by the tap) stays
// conflict-free; the price is five integer instructions per fragment address and tap (computed one step ahead, under the MFMAs).
// With DMAs only, vmcnt counts prove arrival (as in conv_igemm_dma.hip), so: weight stages in a ring of THREE (the DMA of step
// s + 2 is issued in step s), the patch double-buffered (chunk c + 1 is issued in the first step of chunk c), one counted wait and
// one barrier per step.  LDS of the 128-row tile: 2 x 12 KB patch + 3 x 8 KB weights = 48 KB -> three workgroups per CU as before.
//
// Measured (128 -> 128 @256^2, B = 32, same box): register-staged kernel 0.813-0.832 ms, this one with 128-row tiles 0.752-0.774, with
// 256-row tiles 0.720-0.737 (the default; two workgroups of four waves per CU).  Ablations of the 128-row form (0.745 ms): no weight DMAs
// 0.611, no patch DMAs 0.629, no MFMAs 0.511, no fragment reads 0.539, no barriers 0.745, no epilogue 0.605 — the stages now add up instead
// of hiding behind one another's latency; what is left is the LDS traffic of 64 x 64 wave tiles (1 KB of fragment reads per MFMA, the
// reason the 256-row tile with its 128 x 64 wave tiles is faster) and the epilogue's share of a K = 1152 tile.
// Also tried on the 256-row tile, both bit-identical and both without effect (0.725-0.733 vs 0.720-0.737 ms): a ring of four weight stages;
// fragment reads running half a step ahead of the MFMAs (the k-step-0 registers refilled for step s + 1 behind a mid-step barrier, under
// the MFMAs of k-step 1, and vice versa) — neither the DMAs' nor the LDS's latency is what is left.
#include "wgs_common.h"
#include "conv_args.h"
#include "conv_epilogue.h"
#include "conv_scheme.h"

typedef wgsconv::epi_f32x16 f32x16;

namespace {

using wgsconv::ConvArgs;

#ifndef WGS_QABL
#define WGS_QABL 0   // development ablations (tools/build_abl.sh qabl): 1 no weight DMAs, 2 no patch DMAs, 4 no MFMAs, 5 no fragment reads, 6 no barriers,
                     // 7 no epilogue (the accumulators sunk into one conditional store)
#endif

#ifndef WGS_PD_PRIO
#define WGS_PD_PRIO 0        // 1: raise the wave's priority for the MFMAs of a step — measured 0.707-0.718 vs 0.705-0.715 ms without: nothing
#endif

#ifndef WGS_PD_NST256
#define WGS_PD_NST256 3      // weight stages of the 256-row tile (4 = 80 KB, two workgroups fill the CU's 160 KB exactly: measured 0.721 vs 0.722 ms,
                             // i.e. with two steps of lead the DMAs' latency is covered)
#endif

constexpr int WN = 64, TN = 2;                      // wave tiles are 64 columns wide: BN / 64 wave columns x 2 wave rows
constexpr int ROW = 64;                             // bytes per LDS row (32 fp16 channels), patch and weights
constexpr int BI = 2;                               // weight DMA instructions per wave and step: BN / 16 rows-of-16 over BN / 32 waves
constexpr int BK = 32;
constexpr int OOB = (int)0x80000000;

// BM = 128: 8 x 16 pixels, wave tiles 64 x 64, 180 patch pixels in 12 DMA instructions, 48 KB -> three workgroups per CU.
// BM = 256: 8 x 32 pixels, wave tiles 128 x 64 (six fragment reads per eight MFMAs instead of eight; the weights of a step serve twice
// the pixels), 340 patch pixels in 24 instructions, 72 KB and ~210 VGPRs -> two workgroups per CU.
template <int BM, int BN> struct Tile {
    static constexpr int WAVES_N = BN / WN, NW = 2 * WAVES_N, NT = 64 * NW;      // 128 columns: 4 waves; 256 columns: 8 waves (one workgroup per CU)
    static constexpr int WST = BN * ROW;                // one weight stage (one tap of one chunk)
    // Tile = TH rows x TW pixels of one sample.  ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32): with
    // TW = 32 the 32 rows of an A fragment are 32 CONSECUTIVE patch pixels and every group holds 16 distinct pixels mod 16 -> conflict-free
    // under the swizzle for any tap shift.  With TW = 16 lanes 16-31 sit one patch row (18 pixels) further: two 2-way conflicts per group,
    // i.e. every A read takes twice its LDS cycles (kept for the 128-row tile: 32 x 4 tiles would need a 16 KB patch image, two workgroups per CU).
    // (Measured: 16 x 16 -> 8 x 32 tiles at 256 rows: 0.886 -> 0.875-0.905 of the register-staged kernel's time, i.e. nothing — the LDS is not
    // what the tile waits for.)
    static constexpr int TW = BM == 128 ? 16 : 32, TWL = BM == 128 ? 4 : 5;
    static constexpr int PW = TW + 2;
    static constexpr int TH = BM / TW, PH = TH + 2, NPIX = PW * PH;
    static constexpr int PPIX = (NPIX + 16 * NW - 1) / (16 * NW) * (16 * NW);
    static constexpr int PB_BYTES = PPIX * ROW;         // one patch buffer
    static constexpr int PI = PPIX / 16 / NW;           // patch DMA instructions per wave and chunk
    static constexpr int WM = BM / 2, TM = WM / 32;
    static constexpr int NST = BM == 128 ? 3 : WGS_PD_NST256;       // weight stages: the DMA of step s + NST - 1 is issued in step s
    static constexpr int SMEM = 2 * PB_BYTES + NST * WST;
};

typedef __attribute__((address_space(3))) unsigned char lds_byte;

template <int V> struct Tag { static constexpr int value = V; };

struct PdGeom {
    int tiles_x, tiles_per_img, ntaps;
    int tappix[16];      // patch pixel offset of tap t: (dy + 1) * 18 + (dx + 1)
    int tapw[16];        // byte offset of tap t's slab inside a weight row of the 16-bit plane
};

template <int BM, int BN, bool RGB>
__global__ __launch_bounds__(64 * 2 * (BN / 64), BN == 256 ? 1 : (BM == 128 ? 3 : 2)) void patch_dma_kernel(const ConvArgs p, const PdGeom g) {
    typedef Tile<BM, BN> T;
    constexpr int WAVES_N = T::WAVES_N, NW = T::NW, WST = T::WST;
    static_assert(BN / 16 / NW == BI, "two weight DMA instructions per wave and step");
    constexpr int TW = T::TW, TWL = T::TWL, PW = T::PW, TH = T::TH, NPIX = T::NPIX, PPIX = T::PPIX, PB_BYTES = T::PB_BYTES, PI = T::PI, WM = T::WM, TM = T::TM, NST = T::NST;
    typedef wgsconv::Scheme<1> SC;
    typedef SC::frag frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* patch = smem_b;
    unsigned char* wring = smem_b + 2 * PB_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ntn = p.Co / BN;
    int bid;      // XCD-aware order over the (tile-major, column-tile minor) list, as in the other kernels
    {
        const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int qn = nb >> 3, rn = nb & 7;
        bid = xcd * qn + min(xcd, rn) + slot;
    }
    const int tile = bid / ntn, n0 = (bid % ntn) * BN;
    const int b = tile / g.tiles_per_img;
    const int trem = tile - b * g.tiles_per_img;
    const int tyq = trem / g.tiles_x;
    const int ty0 = tyq * TH, tx0 = (trem - tyq * g.tiles_x) * TW;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.a_hi), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbh = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w_hi), 0, p.w_bytes, 0x00020000);

    // ---- per-lane DMA sources.  Instruction k of a buffer fills LDS bytes [k * 1024, (k + 1) * 1024): lane -> pixel 16 k + lane / 4, slot lane % 4
    const int lrow = lane >> 2, slot = lane & 3;
    int p_goff[PI];
#pragma unroll
    for (int j = 0; j < PI; ++j) {
        const int pp = (wave * PI + j) * 16 + lrow;
        const int piece = slot ^ ((pp >> 2) & 3);
        const int pr = pp / PW, pc = pp - pr * PW;
        const int iy = ty0 - 1 + pr, ix = tx0 - 1 + pc;
        const bool v = (pp < NPIX) & ((unsigned)iy < (unsigned)p.Hi) & ((unsigned)ix < (unsigned)p.Wi);
        p_goff[j] = v ? ((b * p.Hi + iy) * p.Wi + ix) * p.Ci * 2 + piece * 16 : OOB;      // OOB: the DMA writes zeros (padding, slack)
    }
    int b_off[BI];
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int row = (wave * BI + j) * 16 + lrow;
        const int lc = slot ^ ((row >> 2) & 3);
        b_off[j] = (int)((long)(n0 + row) * p.w_row_stride + lc * 8) * 2;
    }
    // the per-tap tables live in two VGPRs (lane t holds tap t) and are read with v_readlane: no scalar loads in the steps
    const int v_tp = g.tappix[lane & 15], v_tw = g.tapw[lane & 15];
    const int cpt = p.Ci / BK, ntaps = g.ntaps;

    auto issue_patch = [&](int pb, int c) {
        const unsigned cbyte = c < cpt ? (unsigned)(c * (BK * 2)) : (unsigned)OOB;
        lds_byte* d0 = (lds_byte*)(patch + pb * PB_BYTES + wave * (PI * 1024));
#pragma unroll
        for (int j = 0; j < PI; ++j) {
            if (WGS_QABL == 2) { asm volatile("" :: "v"((int)((unsigned)p_goff[j] + cbyte))); continue; }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, d0 + j * 1024, 16, (int)((unsigned)p_goff[j] + cbyte), 0, 0, 0);
        }
    };
    auto issue_w = [&](int stage_off, int c, int t) {        // weights of (chunk c, tap t) -> the stage at byte offset stage_off; past the end: zeros
        const unsigned delta = c < cpt ? (unsigned)(__builtin_amdgcn_readlane(v_tw, t) + c * (BK * 2)) : (unsigned)OOB;
        lds_byte* d0 = (lds_byte*)(wring + stage_off + wave * (BI * 1024));
#pragma unroll
        for (int j = 0; j < BI; ++j) {
            if (WGS_QABL == 1) { asm volatile("" :: "v"((int)((unsigned)b_off[j] + delta))); continue; }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rbh, d0 + j * 1024, 16, (int)((unsigned)b_off[j] + delta), 0, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- operand fragment addressing
    const int l31 = lane & 31, lh = lane >> 5;
    int q[TM];               // patch pixel of this lane's fragment row for the tap (dy, dx) = (-1, -1)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * WM + i * 32 + l31;
        q[i] = (r >> TWL) * PW + (r & (TW - 1));
    }
    const int bswz = (l31 >> 2) & 3;
    const int b_rd0 = (wn * WN + l31) * ROW + (((0 + lh) ^ bswz) << 4), b_rd1 = (wn * WN + l31) * ROW + (((2 + lh) ^ bswz) << 4);
    // A fragment of (tap t, patch buffer pb), k-step 0; k-step 1 = the same address ^ 32 (piece + 2)
    auto a_addr = [&](int pb, int t, int* aa) {
        const int tp = __builtin_amdgcn_readlane(v_tp, t) + pb * PPIX;      // (PPIX is a multiple of 16: the swizzle term is that of the pixel in its buffer)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int pp = q[i] + tp;
            const int sl = ((pp >> 2) & 3) ^ lh;
            aa[i] = (pp << 6) | (sl << 4);
        }
    };

    // ---- prologue: patch of chunk 0, weights of steps 0 and 1.  (The two table registers are the kernel's only register loads: waited for
    // before the first DMA is issued, since counts over a mix of the two kinds prove nothing.)
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(v_tp), "v"(v_tw) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    issue_patch(0, 0);
    int c2 = 0, t2 = 0;                      // (chunk, tap) of step s + NST - 1, advanced with the steps
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) {
        issue_w(st * WST, c2, t2);
        if (++t2 == ntaps) { t2 = 0; ++c2; }
    }
    int r_cur = 0;                           // byte offset of the stage step s reads; s + 1: the next one, s + NST - 1: the one before (ring)
    int aa[TM];
    a_addr(0, 0, aa);
    int bb0 = b_rd0, bb1 = b_rd1;
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        if (WGS_QABL != 6) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");    // (the bare s_barrier does not order memory operations for the compiler)
        __builtin_amdgcn_sched_barrier(0);
    };
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * BI) : "memory");      // own pieces of the patch and of step 0's weights
    bar();

    // One tap of one chunk.  FIRST (tap 0): the next chunk's patch is issued too.  LATE = what may stay in flight at the end of the step
    // behind the weights of step s + 1 (vmcnt counts in issue order): the weights of steps s + 2 .. s + NST - 1, and in taps 0 .. NST - 2 the
    // patch, which was issued after tap 0's weight DMAs — from tap NST - 1 on it is older than the weights waited for, i.e. it has about NST
    // steps to land.
    auto step = [&](auto first_tag, auto late_tag, int c, int t) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int LATE = decltype(late_tag)::value;
        frag af[2][TM], bf[2][TN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (WGS_QABL == 5) {
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "=v"(bf[ks][j]) : "v"(bb0), "v"(bb1));
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" : "=v"(af[ks][i]) : "v"(aa[i]));
                continue;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[ks][j] = *reinterpret_cast<const frag*>(wring + (ks ? bb1 : bb0) + j * 32 * ROW);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[ks][i] = *reinterpret_cast<const frag*>(patch + (aa[i] ^ (ks * 32)));
        }
        __builtin_amdgcn_sched_barrier(0);
        const int r_n1 = r_cur + WST == NST * WST ? 0 : r_cur + WST;
        issue_w(r_cur == 0 ? (NST - 1) * WST : r_cur - WST, c2, t2);
        if (FIRST) issue_patch((c + 1) & 1, c + 1);
        // the next step's fragment addresses, under this step's MFMAs
        int t1 = t + 1, c1 = c;
        if (t1 == ntaps) { t1 = 0; ++c1; }
        a_addr(c1 & 1, t1, aa);
        bb0 = r_n1 + b_rd0; bb1 = r_n1 + b_rd1;
        __builtin_amdgcn_sched_barrier(0);
        if (WGS_PD_PRIO) __builtin_amdgcn_s_setprio(1);      // the wave that has its operands goes first on its SIMD (the other workgroup's wave is staging)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (WGS_QABL == 4) { asm volatile("" :: "v"(af[ks][i]), "v"(bf[ks][j])); continue; }
                    acc[i][j] = SC::mma(&af[ks][i], &bf[ks][j], acc[i][j]);
                }
        if (WGS_PD_PRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LATE) : "memory");      // step s + 1's weights (and everything older) have landed
        bar();
        r_cur = r_n1;
        if (++t2 == ntaps) { t2 = 0; ++c2; }
    };
    for (int c = 0; c < cpt; ++c) {
        step(Tag<1>{}, Tag<(NST - 2) * BI + PI>{}, c, 0);
#pragma unroll
        for (int t = 1; t < NST - 1; ++t) step(Tag<0>{}, Tag<(NST - 2) * BI + PI>{}, c, t);
        for (int t = NST - 1; t < ntaps; ++t) step(Tag<0>{}, Tag<(NST - 2) * BI>{}, c, t);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the zero-fill DMAs past the last step: the epilogue re-uses the LDS
    __syncthreads();

    if (WGS_QABL == 7) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
        if (sum == 12345.678f) p.y[tid] = sum;
        return;
    }
    // ---- epilogue (contract of conv_igemm.hip; rows are the TH x TW tile pixels of sample b)
    float op_mult = 1.f, op_inv = 1.f;       // the producer scaled the plane by the power-of-two operand scale (conv_scheme.h): undone here
    wgsconv::operand_scale(p.a_amax, p.a_amax2, p.a_bound, op_mult, op_inv);
    int* r_pix = reinterpret_cast<int*>(smem_b);
    int* r_b = r_pix + BM;
    float* r_nz = reinterpret_cast<float*>(r_b + BM);
    int* r_add = reinterpret_cast<int*>(r_nz + BM);
    if (tid < BM) {
        const int ty = tid >> TWL, tx = tid & (TW - 1);
        const int oy = (ty0 + ty) * p.osy + p.oy0, ox = (tx0 + tx) * p.osx + p.ox0;
        const int hw = oy * p.Wo + ox;
        r_pix[tid] = b * p.Ho * p.Wo + hw;
        r_b[tid] = b;
        r_nz[tid] = (p.noise && p.noise_w) ? p.noise_w[0] * p.noise[hw] : 0.f;
        r_add[tid] = (b * (p.Ho >> p.add_ups) + (oy >> p.add_ups)) * (p.Wo >> p.add_ups) + (ox >> p.add_ups);
    }
    __syncthreads();
    if constexpr (RGB) {          // ToRGB in the epilogue (the tile holds all 128 output channels)
        wgsconv::conv_epilogue_rgb<BM, TM, TN, WM, WN, WAVES_N>(p, acc, smem_b, wm, wn, l31, lh, tid, op_inv);
        return;
    }
    wgsconv::conv_epilogue_apply<BM, TM, TN, WM, WN>(p, acc, smem_b, n0, wm, wn, l31, lh, op_inv);
}

}  // namespace

namespace wgsconv {

// 0 = launch taken.  a: as handed to the patch kernels (16-bit weight extents), fp16 activation plane, scheme 1; the caller has checked
// stride 1 / same size / pre-split weights.  Taken for power-of-two maps >= 16 wide with a multiple of 8 rows, a 3x3 neighbourhood
// (3 .. 16 taps inside dy, dx in -1 .. 1), Cin % 32 == 0, Cout % 128 == 0 (256 columns per workgroup where Cout % 256 == 0).
int launch_patch_dma(const ConvArgs& a, hipStream_t st) {
    if (!a.a_hi || a.sch != 1 || a.a_scale || !a.w_hi || a.ups || a.isy != 1 || a.isx != 1) return 1;
    if (a.Ci % BK || a.Co % 128 || a.ntaps < 3 || a.ntaps > 16) return 1;
    const int bm = wgs_flags().patch_dma_bm;
    // 256 output columns per workgroup (8 waves, one workgroup per CU; WGS_PATCH_DMA_BN256=1): a weight stage and a patch serve twice the MFMAs.
    // Bit-identical to the register-staged 256 x 256 patch tile; measured against the LDS-DMA kernel that owns these launches (conv_igemm_dma.hip,
    // whose two wave groups alternate between memory and MFMA phases): 512 -> 512 @64^2 0.596-0.604 vs 0.580-0.588 ms, 256 -> 256 @128^2 0.621-0.624
    // vs 0.605-0.613 — 2-3 % slower, so it is not the default route.
    const int bn = (bm == 256 && a.Co % 256 == 0 && wgs_flags().patch_dma_bn256) ? 256 : 128;
    const int TW = bm == 128 ? Tile<128, 128>::TW : Tile<256, 128>::TW, PW = TW + 2;
    const int th = bm / TW;
    if (a.Wg < TW || (a.Wg & (a.Wg - 1)) || a.Hg % th || a.Hg != a.Hi || a.Wg != a.Wi) return 1;
    if (a.rgb_out && a.Co != bn) return 1;
    PdGeom g;
    for (int t = 0; t < 16; ++t) { g.tappix[t] = 0; g.tapw[t] = 0; }
    for (int t = 0; t < a.ntaps; ++t) {
        if (a.dy[t] < -1 || a.dy[t] > 1 || a.dx[t] < -1 || a.dx[t] > 1) return 1;
        g.tappix[t] = (a.dy[t] + 1) * PW + (a.dx[t] + 1);
        g.tapw[t] = a.tap_w[t] >> 1;
    }
    g.ntaps = a.ntaps;
    g.tiles_x = a.Wg / TW;
    g.tiles_per_img = (a.Hg / th) * g.tiles_x;
    const int nblocks = a.B * g.tiles_per_img * (a.Co / bn);
    if (nblocks < 200) return 1;
    auto go = [&](auto k, int nt, int smem, const char* name) {
        wgs_note_kernel("%s", name);
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        WGS_LAUNCH(k, dim3((unsigned)nblocks), dim3(nt), smem, st, a, g);
    };
    if (bn == 256) {
        typedef Tile<256, 256> T;
        if (a.rgb_out) go(patch_dma_kernel<256, 256, true>, T::NT, T::SMEM, "patch_dma_kernel<256, 256, true>");
        else go(patch_dma_kernel<256, 256, false>, T::NT, T::SMEM, "patch_dma_kernel<256, 256, false>");
    } else if (bm == 256) {
        typedef Tile<256, 128> T;
        if (a.rgb_out) go(patch_dma_kernel<256, 128, true>, T::NT, T::SMEM, "patch_dma_kernel<256, 128, true>");
        else go(patch_dma_kernel<256, 128, false>, T::NT, T::SMEM, "patch_dma_kernel<256, 128, false>");
    } else {
        typedef Tile<128, 128> T;
        if (a.rgb_out) go(patch_dma_kernel<128, 128, true>, T::NT, T::SMEM, "patch_dma_kernel<128, 128, true>");
        else go(patch_dma_kernel<128, 128, false>, T::NT, T::SMEM, "patch_dma_kernel<128, 128, false>");
    }
    return 0;
}

}  // namespace wgsconv
