// StyleGAN2 up-sampling layer in ONE kernel (fp16 operand schemes and split-bf16): modulated stride-2 transposed 3x3 conv, demodulation,
// 4x4 blur, noise, bias, leaky-relu*sqrt(2)   (models/StyleGAN2/model.py:201-212 conv_transpose2d + Blur, :231-241, :264).
//
// The unfused path runs the transposed conv as four sub-pixel phase GEMMs that write a (2H+1)^2 intermediate t (1.08 GB at
// 128->256 px, B = 32), and a second kernel reads t back through the blur.  Here one workgroup owns a 16 x 12 block of INPUT
// cells of one sample and 64 output channels:
//
//   * GEMM rows = the 18 x 14 grid g of cell positions (the block + a one-cell halo: the 4-tap blur of the 32 x 24 output
//     block needs t one position beyond it on every side; 252 of the tile's 256 rows);  t[2g + p] for the four parities p = (py, px) are FOUR accumulator sets
//     of the same rows:  t_p[g] = sum over the phase's taps (ky = py mod 2, kx = px mod 2) of x[g + (p - k)/2] * w[k]
//   * the activation operand is staged ONCE per 32-channel chunk as the 19 x 15 input patch; the nine (phase, tap) products
//     read it with four different row shifts, so one A fragment feeds up to four MFMAs (phases) and is read 4x, not 9x
//   * epilogue: accumulators * demodulation -> LDS as the 36 x 28 x 32-channel t tile (fp32, 126 KB), then the blur +
//     noise + bias + activation over the 32 x 24 outputs with a sliding row window, stored as 128-B channel runs.
//
// Grid positions outside the image produce exact zeros (their patch pixels are out of range -> 0), which is
// upfirdn2d's zero padding of t.  MFMA efficiency = 192 useful cells of 256 rows (x tile-edge waste: none in x for 32 / 64 / 128-wide maps); in exchange the layer
// loses the intermediate's write + two reads and the blur kernel.
//
// LDS: two patch buffers + two weight stages of TS products each (one barrier per stage), re-used by the epilogue's t tile.
#include "wgs_common.h"
#include "conv_scheme.h"
#include "../../include/wgs.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

#ifndef WGS_UABL
#define WGS_UABL 0   // development ablations (tools/build_abl.sh uabl): 1 no epilogue, 2 no MFMA, 3 no patch global loads,
                     // 4 no weight DMA, 5 no K loop, 6 no blur arithmetic, 7 no output stores, 8 no patch conversion + LDS stores, 9 = 3 + 8
#endif

constexpr int BK = 32;
constexpr int ROW = 64;                  // bytes per LDS row of a weight plane (DMA: unpadded, XOR-swizzled 16-B slots)
constexpr int PROW = 80;                 // bytes per patch pixel in LDS (32 channels of one 16-bit plane + 16 B)
// Patch ROWS are PW pixels + PPAD bytes apart (round 6).  An A fragment's 32 GEMM rows are consecutive grid positions, and the grid is GX = PW - 1
// wide: where a fragment runs from one grid row into the next the pixel index jumps by 2.  ds_read_b128 is served in the lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32), a group is conflict-free iff its 16 addresses differ mod 256 B, i.e. iff 5 * pixel is a
// bijection mod 16 over the group: true for consecutive pixels, false behind a jump of 2 (both groups of nearly every fragment: 2-way, 24 - 27 %
// of this kernel's LDS cycles in profiles/r5_conv_pmc.md).  With (row stride / 16 - 5 (GX - 1)) = 5 (mod 16) a row change looks like one pixel
// step to the banks: 11 extra 16-B slots per patch row for both tile shapes (18- and 16-wide grids).
constexpr int PPAD = 176;
// (round 5, measured and reverted: unpadded 64-byte rows with an XOR swizzle make room for all nine products of an fp16 x2 chunk in ONE weight
//  stage — one full-drain barrier per chunk instead of two: 0.940 -> 0.936 ms at 512 -> 256 @64, 1.034 -> 1.055 ms at 256 -> 128 @128.  The barriers
//  are not what the kernel waits for either; neither are the 4-wave tiles at two workgroups per CU (WGS_UP_GH8: 1.05 / 1.24 ms).)
constexpr int OOB = (int)0x80000000;
constexpr int BN = 64;

typedef __attribute__((address_space(3))) unsigned char lds_byte;

// (phase, tap) products in issue order, grouped by the patch row shift they read
//   shift 0: (dy, dx) = (0, 0)   1: (0, -1)   2: (-1, 0)   3: (-1, -1);   phase = py * 2 + px;   weight tap = ky * 3 + kx
__device__ constexpr int T_PH[9] = {0, 1, 2, 3, 0, 2, 0, 1, 0};
__device__ constexpr int T_SH[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
constexpr int T_W_HOST[9] = {0, 1, 3, 4, 2, 5, 6, 7, 8};
// patch row shift of the four (dy, dx) groups, in patch pixels (PW = patch width of the tile shape)
__device__ constexpr int sh_off(int sh, int PRS) { return sh == 0 ? PRS + PROW : (sh == 1 ? PRS : (sh == 2 ? PROW : 0)); }   // in bytes (PRS: patch row stride)

struct UpArgs {
    const float* x; const unsigned short* w_hi; const unsigned short* w_lo; float* y;
    const float* a_scale; const float* col_scale; const float* bias; const float* noise; const float* noise_w; const float* kern;
    const float* a_amax; const float* a_amax2; float* y_amax;
    float a_bound, alpha;
    int B, H, Ci, Co, a_ld, col_ld;
    int x_bytes, w_bytes, s_bytes, y_bytes;
    int tiles_x, tiles_per_img;
    unsigned short* y_f16; const float* y_f16_scale; float* y_f16_bound;     // optional plane of the NEXT conv's operand (wgs.h)
    float y_f16_mul, y_f16_add; int y_f16_ld, yh_bytes;
    int w_row_stride;          // elements between output channels of a weight plane (9 * Ci)
    int tap_w[9];              // element offset of issue-order product t's weight tap (T_W[t] * Ci)
};

// GH selects the tile: 16 (8 waves, 16 x 12 cells, one workgroup per CU) or 8 (4 waves, 14 x 6 cells, 75 KB of
// LDS: TWO workgroups per CU, whose staging, barriers and blur epilogues overlap each other's MFMAs).
template <int SCH, int GH>
struct UpCfg {
    typedef wgsconv::Scheme<SCH> SC;
    static constexpr int NA = SC::NA, NB = SC::NB;
    static constexpr int NW = GH / 2, NT = 64 * NW;
    // grid of cell positions g: GH == 16 -> 18 x 14 (252 of the 256 GEMM rows; 16 x 12 cells: 128 / 64 / 32-wide maps divide by
    // 16, so only the tile rows have an edge remainder), GH == 8 -> 16 x 8 (14 x 6 cells)
    static constexpr int GX = GH == 16 ? 18 : 16, GY = GH == 16 ? 14 : 8;
    static constexpr int CX = GX - 2, CY = GY - 2;                     // cells of a tile
    static constexpr int PW = GX + 1, PH = GY + 1, NPIX = PW * PH, PALLOC = (NPIX + 7) / 8 * 8;
    static constexpr int TW = 2 * GX, TH = 2 * GY;                     // t tile positions
    static constexpr int OW = 2 * CX, OR = 2 * CY;                     // output block
    // products per weight stage (split-bf16: two planes of BOTH operands — three products fit beside the patch planes, one in the two-per-CU form)
    static constexpr int TS = GH == 16 ? (NA * NB == 1 ? 9 : (NA * NB == 2 ? 5 : 3)) : (NA * NB == 1 ? 5 : (NA * NB == 2 ? 3 : 1));
    static constexpr int NSTEP = (9 + TS - 1) / TS;
    static constexpr int PRS = PW * PROW + PPAD;                       // bytes between patch rows
    static_assert(((PRS / 16 - 5 * (GX - 1)) & 15) == 5, "patch row stride: a grid-row change must look like one pixel step to the LDS banks");
    static constexpr int P_BYTES = PH * PRS;
    static constexpr int B_BYTES = BN * ROW, B_TAP = NB * B_BYTES, B_STAGE = TS * B_TAP;
    static constexpr int K_BYTES = 2 * NA * P_BYTES + 2 * B_STAGE;
    static constexpr int T_BYTES = TW * TH * 32 * 4;                   // t tile: TH x TW positions x 32 channels fp32
    static constexpr int MAIN = K_BYTES > T_BYTES ? K_BYTES : T_BYTES;
    static constexpr int AUX_FLOATS = OR * OW + 3 * BN;                // noise of the output block | bias | demodulation | next layer's style
    static constexpr int SMEM = MAIN + AUX_FLOATS * 4;
};

template <int SCH, int GH>
__global__ __launch_bounds__(32 * GH, GH == 8 ? 2 : 1) void upconv_blur_kernel(const UpArgs p) {
    typedef UpCfg<SCH, GH> CF;
    static_assert(CF::SMEM <= 160 * 1024, "LDS budget");
    typedef wgsconv::Scheme<SCH> SC;
    typedef typename SC::frag frag;
    constexpr int NA = SC::NA, NB = SC::NB;
    constexpr int NW = CF::NW, NT = CF::NT;
    constexpr int WM = 32, TN = 2;              // a wave: 32 grid rows x all 64 channels x 4 phases = 128 accumulator registers
    constexpr int TS = CF::TS, NSTEP = CF::NSTEP;
    constexpr int P_BYTES = CF::P_BYTES, B_BYTES = CF::B_BYTES, B_TAP = CF::B_TAP, B_STAGE = CF::B_STAGE;
    constexpr int PALLOC = CF::PALLOC, NPIX = CF::NPIX, PRS = CF::PRS;
    constexpr int NPL = (PALLOC * 8 + NT - 1) / NT;     // float4 patch loads per thread and chunk (5)
    constexpr int OR = CF::OR, OW = CF::OW, GX = CF::GX, GY = CF::GY, PW = CF::PW, TW = CF::TW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* patch = smem_b;                      // two buffers of NA planes
    unsigned char* bst = smem_b + 2 * NA * P_BYTES;     // two weight stages
    float* aux_nz = reinterpret_cast<float*>(smem_b + CF::MAIN);
    float* aux_bias = aux_nz + OR * OW;
    float* aux_cs = aux_bias + BN;
    float* aux_sn = aux_cs + BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave;
    const int ntn = (p.Co + BN - 1) / BN;       // Co = 32 (StyleGAN2-1024's last up-sampling layer): one half-filled column tile
    int bid;
    {
        const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int qn = nb >> 3, rn = nb & 7;
        bid = xcd * qn + min(xcd, rn) + slot;
    }
    const int tile = bid / ntn, n0 = (bid % ntn) * BN;
    const int b = tile / p.tiles_per_img;
    const int trem = tile - b * p.tiles_per_img;
    const int y0 = (trem / p.tiles_x) * CF::CY, x0 = (trem % p.tiles_x) * CF::CX;
    const int Ho = 2 * p.H;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbh = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w_hi), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbl = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>((NB == 2 && p.w_lo) ? p.w_lo : p.w_hi), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a_scale), 0, p.s_bytes, 0x00020000);

    // ---- patch staging: element e = tid + j*NT -> patch pixel e / 8, float4 q = e % 8
    const int q = tid & 7;
    int p_goff[NPL], p_loff[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        // pixel of this 8-lane group: within every aligned run of eight pixels the groups take them in the order 0 4 1 5 2 6 3 7, so that the two pixels
        // of a ds_write_b64 lane group are 4 pixels = 320 B = 16 banks (mod 32) apart: adjacent pixels (80 B) overlapped on 4 of the 32 banks
        const int pl = (tid + j * NT) >> 3;
        const int pp = (pl & ~7) | ((pl & 7) >> 1) | ((pl & 1) << 2);
        const int pr = pp / PW, pc = pp - pr * PW;
        const int iy = y0 - 2 + pr, ix = x0 - 2 + pc;
        const bool v = pp < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.H;
        p_goff[j] = v ? (((b * p.H + iy) * p.H + ix) * p.Ci + q * 4) * 4 : OOB;
        p_loff[j] = pp < NPIX ? pr * PRS + pc * PROW + q * 8 : -1;
    }
    float op_mult = 1.f, op_inv = 1.f;
    if (SCH != 0) wgsconv::operand_scale(p.a_amax, p.a_amax2, p.a_bound, op_mult, op_inv);      // (bf16 has fp32's exponent range: no operand scale)
    const int cpt = p.Ci / BK;
    float4 pr_[NPL];
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
    auto load_patch = [&](int c) {
        const int cbyte = c < cpt ? c * (BK * 4) : OOB;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (WGS_UABL != 3 && WGS_UABL != 9) v = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)((unsigned)p_goff[j] + (unsigned)cbyte), 0, 0);
            else asm volatile("" : "+v"(v));
            pr_[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsc, c < cpt ? (b * p.a_ld + q * 4 + c * BK) * 4 : OOB, 0, 0);
        if (c < cpt) sc = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto store_patch = [&](int buf) {
        if (WGS_UABL == 8 || WGS_UABL == 9) { asm volatile("" :: "v"(pr_[0].x), "v"(pr_[NPL - 1].w)); return; }
        unsigned char* pb = patch + buf * NA * P_BYTES;
        // the power-of-two operand scale folded into the style HERE (not where the style is loaded: a use there waits for the chunk's loads at
        // their issue): fl(x * s) * 2^k == fl(x * (s * 2^k)) bit for bit, both scalings are exact
        const float4 scm = make_float4(sc.x * op_mult, sc.y * op_mult, sc.z * op_mult, sc.w * op_mult);
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            float4 v = pr_[j];
            v.x = __fmul_rn(v.x, scm.x); v.y = __fmul_rn(v.y, scm.y); v.z = __fmul_rn(v.z, scm.z); v.w = __fmul_rn(v.w, scm.w);
            const f32x4 f = {v.x, v.y, v.z, v.w};
            uint2 h, l;
            SC::cvt4(f, h, l);
            if (p_loff[j] >= 0) {
                *reinterpret_cast<uint2*>(pb + p_loff[j]) = h;
                if (NA == 2) *reinterpret_cast<uint2*>(pb + P_BYTES + p_loff[j]) = l;
            }
        }
    };

    // ---- weight DMA: instruction k of a stage = (product u, plane pl, 16-row group grp); waves take k = wave, wave + NW, ...
    const int lrow = lane >> 2, slot = lane & 3;
    constexpr int NI = TS * NB * 4;                     // DMA instructions per stage
    constexpr int IPW = (NI + NW - 1) / NW;
    auto issue_b = [&](int stage, int c, int s) {       // weights of (chunk c, step s) -> stage; past the end: zeros
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int k = wave + i * NW;
            if (k < NI) {
                const int u = k / (NB * 4), pl = (k >> 2) % NB, grp = k & 3;
                const int t = s * TS + u;
                const int row = grp * 16 + lrow;
                const int lc = slot ^ ((row >> 2) & 3);
                const bool ok = c < cpt && t < 9;
                const int off = ok ? ((n0 + row) * p.w_row_stride + p.tap_w[t < 9 ? t : 0] + c * BK + lc * 8) * 2 : OOB;
                lds_byte* d = (lds_byte*)(bst + stage * B_STAGE + u * B_TAP + pl * B_BYTES + grp * 16 * ROW);
                if (WGS_UABL == 4) { asm volatile("" :: "v"(off)); continue; }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(pl ? rbl : rbh, d, 16, off, 0, 0, 0);
            }
        }
    };

    f32x16 acc[4][TN];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ph][j][r] = 0.f;

    // ---- operand fragment addressing
    const int l31 = lane & 31, lh = lane >> 5;
    const int m_a = wm * WM + l31;
    // GEMM rows past the grid (252 .. 255 of the 18 x 14 grid) shadow the position 16 rows back: the same addresses as lanes 12 - 15, which
    // ds_read_b128 serves in the OTHER lane group — shadowing the last position (251) put four lanes on the bank slot of lane 11 (row 235)
    const int m_c = m_a < GX * GY ? m_a : m_a - 16;
    const int pa0 = (m_c / GX) * PRS + (m_c % GX) * PROW + lh * 16;
    const int bswz = (l31 >> 2) & 3;
    const int b_rd = l31 * ROW;
    const int bk0 = ((0 + lh) ^ bswz) << 4, bk1 = ((2 + lh) ^ bswz) << 4;

    // A step's products in issue order (k-step outer, product inner), software-pipelined by hand: the two weight fragments of
    // product q+1 — and its activation fragments when the row shift or the k-step changes — are read from LDS BEFORE the MFMAs of
    // product q (two register sets each, pinned with sched_barrier).  The compiler's own schedule was {ds_read B; wait; 4 MFMAs}
    // per product: the fragment-read latency in front of every MFMA group, with only the partner wave to cover it.
    auto mma_step = [&](int buf, int stage, int s) {
        const unsigned char* pb = patch + buf * NA * P_BYTES;
        const unsigned char* bb = bst + stage * B_STAGE + b_rd;
        constexpr int Q = 2 * TS;                    // (k-step, product) slots; s is a compile-time constant at every call site
        frag af[2][NA], bf[2][TN][NB];
        auto ld_a = [&](int q, frag* a) {
            const int ks = q / TS, t = s * TS + q % TS;
#pragma unroll
            for (int pl = 0; pl < NA; ++pl) a[pl] = *reinterpret_cast<const frag*>(pb + pl * P_BYTES + pa0 + sh_off(T_SH[t], PRS) + ks * 32);
        };
        auto ld_b = [&](int q, frag (*b)[NB]) {
            const int ks = q / TS, u = q % TS;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < NB; ++pl)
                    b[j][pl] = *reinterpret_cast<const frag*>(bb + u * B_TAP + pl * B_BYTES + j * 32 * ROW + (ks ? bk1 : bk0));
        };
        int qa = 0, qb = 0;                          // register set holding the current product's A / B fragments
        ld_a(0, af[0]);
        ld_b(0, bf[0]);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int t = s * TS + q % TS;
            if (t >= 9) continue;                    // (the last step of a chunk may be short)
            int nq = q + 1;
            while (nq < Q && s * TS + nq % TS >= 9) ++nq;
            int qa_next = qa;
            if (nq < Q) {
                ld_b(nq, bf[qb ^ 1]);
                if (nq / TS != q / TS || T_SH[s * TS + nq % TS] != T_SH[t]) {
                    qa_next = qa ^ 1;
                    ld_a(nq, af[qa_next]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);       // (without it hipcc sinks the reads behind three of the product's four MFMAs)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (WGS_UABL == 2) { asm volatile("" :: "v"(af[qa][0]), "v"(bf[qb][j][0])); continue; }
                acc[T_PH[t]][j] = SC::mma(af[qa], bf[qb][j], acc[T_PH[t]][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            qa = qa_next;
            qb ^= 1;
        }
    };
    auto step_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- the epilogue's per-tile scalars go to LDS now (noise of the output block, bias, demodulation): fetched there, every
    // one of them would expose a memory latency with nothing to overlap it
    auto fill_aux = [&]() {
        const float nw = p.noise ? p.noise_w[0] : 0.f;
        for (int e = tid; e < OR * OW; e += NT) {
            const int ly = e / OW, lx = e - ly * OW;
            const int oy = 2 * y0 + ly, ox = 2 * x0 + lx;
            aux_nz[e] = (p.noise && oy < Ho && ox < Ho) ? nw * p.noise[oy * Ho + ox] : 0.f;
        }
        if (tid < BN) {
            const bool cok = n0 + tid < p.Co;           // (columns past Cout: their weight rows read as zeros, their outputs are not stored)
            aux_bias[tid] = cok ? p.bias[n0 + tid] : 0.f;
            aux_cs[tid] = cok ? p.col_scale[(size_t)b * p.col_ld + n0 + tid] : 0.f;
            aux_sn[tid] = (p.y_f16 && cok) ? p.y_f16_scale[(size_t)b * p.y_f16_ld + n0 + tid] : 0.f;
        }
    };

    // ---- main loop: one barrier per weight stage; patch buffers alternate per chunk
    load_patch(0);
    issue_b(0, 0, 0);
    fill_aux();                 // behind the first patch loads and weight DMAs: its own load latency overlaps theirs
    store_patch(0);
    step_barrier();
    int stage = 0;
    for (int c = 0; c < (WGS_UABL == 5 ? 0 : cpt); ++c) {
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const bool last = (s + 1 == NSTEP);
            issue_b(stage ^ 1, last ? c + 1 : c, last ? 0 : s + 1);
            if (s == 0) load_patch(c + 1);
            mma_step(c & 1, stage, s);
            if (last && c + 1 < cpt) store_patch((c + 1) & 1);
            step_barrier();
            stage ^= 1;
        }
    }

    if (WGS_UABL == 1) {
        float sacc = 0.f;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[ph][j][r];
        if (sacc == 12345.678f) p.y[tid] = sacc;
        return;
    }
    // ---- epilogue: t tile -> LDS, blur + noise + bias + activation, 32 channels at a time
    float* T = reinterpret_cast<float*>(smem_b);          // [(2 GH) * 32 positions][32 channels]
    float kf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kf[i] = p.kern[15 - i];
    // rank-1 test of the (flipped) kernel: kf[i][j] == kv[i] * kh[j] with kh = row 0, kv = column 0 / kf[0][0]
    float kv[4], kh[4];
    bool sep = kf[0] != 0.f;
    {
        const float r00 = sep ? 1.f / kf[0] : 0.f;
        float kmax = 0.f, dev = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { kh[i] = kf[i]; kv[i] = kf[i * 4] * r00; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            kmax = fmaxf(kmax, fabsf(kf[i]));
            dev = fmaxf(dev, fabsf(kf[i] - kv[i >> 2] * kh[i & 3]));
        }
        sep = sep && dev <= 2e-7f * kmax;
        // uniform values: as scalar registers the eight taps feed the packed FMAs through op_sel (no {k, k} vector pairs: 16 registers fewer in
        // the blur, which runs while the other channel half's 64 accumulators are still live)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            kh[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, kh[i])));
            kv[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, kv[i])));
        }
    }
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y ? p.y : reinterpret_cast<float*>(p.y_f16), 0, p.y ? p.y_bytes : 0, 0x00020000);
    // The consumer's fp16 operand plane, written here: f16_rn(y * style_next * 2^k).  k comes from an A-PRIORI bound of |y * style_next|
    // (this kernel cannot know its own output maximum before it has run): |t| <= sqrt(taps * Ci) * max|x| by Cauchy-Schwarz (the
    // demodulation factor is 1 / ||w * s||), the blur sums to <= 4, then noise, bias and the activation gain — folded by the caller
    // into y_f16_mul / y_f16_add — times max|style| (a_amax2).  The value is published for the consumer's operand_scale().
    const __amdgpu_buffer_rsrc_t ryh = __builtin_amdgcn_make_buffer_rsrc(p.y_f16 ? p.y_f16 : const_cast<unsigned short*>(p.w_hi), 0, p.y_f16 ? p.yh_bytes : 0, 0x00020000);
    float pl_mult = 1.f;
    if (p.y_f16) {
        const float am = (p.a_amax[0] * p.y_f16_mul + p.y_f16_add) * (p.a_amax2 ? p.a_amax2[0] : 1.f);
        float pl_inv;
        wgsconv::scale_of_bound(am, pl_mult, pl_inv);
        if (p.y_f16_bound && blockIdx.x == 0 && tid == 0) p.y_f16_bound[0] = am;
    }
    constexpr int NSTRIP = (NT / 8) / OW, RS = OR / NSTRIP;  // row strips of the blur stage (2 x 12 rows / 1 x 12 rows)
    static_assert(NSTRIP >= 1 && RS * NSTRIP == OR, "blur strips");
    const int bslot = tid >> 3;                           // (strip, column) of the blur stage
    const int lx = bslot % OW, strip = bslot / OW;
    const bool bl_live = bslot < OW * NSTRIP;
    float vmax = 0.f;
    // T-tile byte offset of this lane's 16 accumulator rows (grid position of row r, phase (0, 0), channel l31), computed ONCE: the four phases
    // and both channel halves add compile-time constants.  (Round 6: the offsets used to be re-derived — two wrap tests, a bound test under a
    // saved exec mask, a 64-bit multiply-add — in front of every one of the 128 stores of a lane; now per group of four rows, 4 x 2 times per tile: ~1 500 of a wave's ~3 200 vector instructions
    // per tile.)  GEMM rows past the grid (252 .. 255 of the 18 x 14 grid) shadow the position 16 rows back in the A-fragment reads, so their
    // accumulators ARE that position's values: they are stored to its slot (same bits from two lanes) instead of being masked out.
    const int mb_t = wm * WM + 4 * lh;                   // this lane's first grid row; its 16 rows are mb_t + (r & 3) + 8 * (r >> 2)
    auto t_off = [&](int r) {
        int m = mb_t + (r & 3) + 8 * (r >> 2);
        m = m < GX * GY ? m : m - 16;
        const int gy = m / GX, gx = m - gy * GX;
        return ((2 * gy * TW + 2 * gx) * 32 + l31) * 4;
    };
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (n0 + half * 32 >= p.Co) break;               // a half-filled column tile (Co = 32)
        {
            const float cs = aux_cs[half * 32 + l31];
            const float al = p.alpha * op_inv;
#pragma unroll
            for (int rg = 0; rg < 16; rg += 4) {
                const int o0 = t_off(rg), o1 = t_off(rg + 1), o2 = t_off(rg + 2), o3 = t_off(rg + 3);
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) {
                    const int po = ((ph >> 1) * TW + (ph & 1)) * 128;        // the phase's position offset: an immediate of the store
                    unsigned char* tb = reinterpret_cast<unsigned char*>(T) + po;
                    // (acc * alpha) * column scale for two rows per packed multiply
                    const f32x2 ta = f32x2{half ? acc[ph][1][rg] : acc[ph][0][rg], half ? acc[ph][1][rg + 1] : acc[ph][0][rg + 1]} * al * cs;
                    const f32x2 tc = f32x2{half ? acc[ph][1][rg + 2] : acc[ph][0][rg + 2], half ? acc[ph][1][rg + 3] : acc[ph][0][rg + 3]} * al * cs;
                    *reinterpret_cast<float*>(tb + o0) = ta.x;
                    *reinterpret_cast<float*>(tb + o1) = ta.y;
                    *reinterpret_cast<float*>(tb + o2) = tc.x;
                    *reinterpret_cast<float*>(tb + o3) = tc.y;
                }
            }
        }
        __syncthreads();
        if (bl_live) {
            const int c = n0 + half * 32 + q * 4;
            const float4 bv = *reinterpret_cast<const float4*>(aux_bias + half * 32 + q * 4);
            const float4 sn = *reinterpret_cast<const float4*>(aux_sn + half * 32 + q * 4);
            const int ox = 2 * x0 + lx;
            // finish one output pixel: noise, bias, activation, magnitude, store
            // (two channels per instruction throughout: v_pk_add / v_pk_mul; the leaky ReLU as max(a, 0.2 a), which equals the
            //  select for every finite a, signed zeros included)
            const f32x2 bv0 = {bv.x, bv.y}, bv1 = {bv.z, bv.w}, sn0 = {sn.x, sn.y}, sn1 = {sn.z, sn.w};
            // per-thread row base of the output address, the noise slot and the row bound: a row of the strip then costs one add, one
            // compare and one select (round 6: each emit used to rebuild its address with a 64-bit multiply-add and a 32-bit multiply
            // under a saved exec mask)
            const int oy_s = 2 * y0 + strip * RS;                                  // first output row of this thread's strip
            const int rstride = Ho * p.Co * 4;                                     // bytes between output rows (uniform)
            const int off_s = ox < Ho ? (((b * Ho + oy_s) * Ho + ox) * p.Co + c) * 4 : OOB;
            const int rows_ok = ox < Ho ? Ho - oy_s : 0;                           // rows k of the strip with k < rows_ok lie inside the image
            const float* nz_s = aux_nz + strip * RS * OW + lx;
            auto emit = [&](f32x2 a0, f32x2 a1, int k) {                           // k: row of the strip (compile-time at every call site)
                const bool ok = k < rows_ok;
                const float nz = nz_s[k * OW];
                const f32x2 nz2 = {nz, nz};
                a0 += nz2 + bv0; a1 += nz2 + bv1;
                const f32x2 s0 = a0 * 0.2f, s1 = a1 * 0.2f;
                a0 = f32x2{fmaxf(a0.x, s0.x), fmaxf(a0.y, s0.y)} * 1.4142135623730951f;
                a1 = f32x2{fmaxf(a1.x, s1.x), fmaxf(a1.y, s1.y)} * 1.4142135623730951f;
                const float rmax = fmaxf(fmaxf(fmaxf(fabsf(a0.x), fabsf(a0.y)), fabsf(a1.x)), fabsf(a1.y));
                vmax = fmaxf(vmax, ok ? rmax : 0.f);
                const int off = ok ? off_s + k * rstride : OOB;
                const u32x4 sv = {__float_as_uint(a0.x), __float_as_uint(a0.y), __float_as_uint(a1.x), __float_as_uint(a1.y)};
                if (WGS_UABL == 7) { asm volatile("" :: "v"(sv), "v"(off)); return; }
                __builtin_amdgcn_raw_buffer_store_b128(sv, ry, off, 0, 0);          // (y == NULL: zero-extent descriptor, the store is dropped)
                if (p.y_f16) {
                    // same roundings as the consumer's own staging of the fp32 tensor: fl32(y * s), exact power of two, f16_rn
                    f32x2 v0 = a0 * sn0, v1 = a1 * sn1;
                    asm volatile("" : "+v"(v0), "+v"(v1));
                    v0 *= pl_mult; v1 *= pl_mult;
                    const f32x4 f = {v0.x, v0.y, v1.x, v1.y};
                    uint2 h, l;
                    wgsconv::Scheme<1>::cvt4(f, h, l);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), ryh, ok ? off >> 1 : OOB, 0, 0);
                }
            };
            if (sep) {
                // separable kernel (StyleGAN2's [1,3,3,1] outer product): horizontal pass on each loaded row, vertical pass over a
                // four-row window of the results: 8 instead of 16 multiply-adds per output and channel — as PACKED fp32 FMAs
                // (v_pk_fma_f32: two channels per instruction; the epilogue is a third of this kernel and pure vector work)
                f32x2 hlo[4], hhi[4];
#pragma unroll
                for (int rr = 0; rr < RS + 3; ++rr) {
                    const int uy = strip * RS + rr + 1;
                    f32x2 h0 = {0.f, 0.f}, h1 = {0.f, 0.f};
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(T + ((uy * TW + lx + 1 + jx) * 32 + q * 4));
                        const f32x2 kk = {kh[jx], kh[jx]};
                        h0 = __builtin_elementwise_fma(f32x2{v.x, v.y}, kk, h0);
                        h1 = __builtin_elementwise_fma(f32x2{v.z, v.w}, kk, h1);
                    }
                    hlo[rr & 3] = h0; hhi[rr & 3] = h1;
                    if (rr >= 3) {
                        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
                        for (int ky = 0; ky < 4; ++ky) {
                            const f32x2 kk = {kv[ky], kv[ky]};
                            a0 = __builtin_elementwise_fma(hlo[(rr - 3 + ky) & 3], kk, a0);
                            a1 = __builtin_elementwise_fma(hhi[(rr - 3 + ky) & 3], kk, a1);
                        }
                        emit(a0, a1, rr - 3);
                    }
                }
            } else {
                float4 win[4][4];
#pragma unroll
                for (int rr = 0; rr < RS + 3; ++rr) {
                    const int uy = strip * RS + rr + 1;
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) win[rr & 3][jx] = *reinterpret_cast<const float4*>(T + ((uy * TW + lx + 1 + jx) * 32 + q * 4));
                    if (rr >= 3) {
                        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 4; ++kx) {
                                const float wv = kf[ky * 4 + kx];
                                const float4 v = win[(rr - 3 + ky) & 3][kx];
                                a.x = fmaf(v.x, wv, a.x); a.y = fmaf(v.y, wv, a.y); a.z = fmaf(v.z, wv, a.z); a.w = fmaf(v.w, wv, a.w);
                            }
                        emit(f32x2{a.x, a.y}, f32x2{a.z, a.w}, rr - 3);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (p.y_amax) {
        vmax = wave_max(vmax);
        if (lane == 0) raise_amax(p.y_amax, vmax);
    }
}

template <int SCH, int GH>
void launch_up(const UpArgs& a0, hipStream_t st) {
    typedef UpCfg<SCH, GH> CF;
    UpArgs a = a0;
    a.tiles_x = (a.H + CF::CX - 1) / CF::CX;
    a.tiles_per_img = a.tiles_x * ((a.H + CF::CY - 1) / CF::CY);
    const int nblocks = a.B * a.tiles_per_img * ((a.Co + BN - 1) / BN);
    auto k = upconv_blur_kernel<SCH, GH>;
    wgs_note_kernel("upconv_blur_kernel<%d, %d>", SCH, GH);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
    WGS_LAUNCH(k, dim3((unsigned)nblocks), dim3(CF::NT), CF::SMEM, st, a);
}

}  // namespace

extern "C" int wgs_sg2_upconv_blur_act(const wgs_upconv_desc* d, wgs_stream_t stream) {
    WGS_CHECK_ARG(d && d->x && d->w_hi && (d->y || d->y_f16) && d->a_scale && d->col_scale && d->bias && d->kernel4x4,
                  "wgs_sg2_upconv_blur_act: null pointer");
    WGS_CHECK_ARG(!d->y_f16 || (d->y_f16_scale && d->a_amax && d->y_f16_ld >= d->Co && d->y_f16_mul > 0.f && d->y_f16_add >= 0.f),
                  "wgs_sg2_upconv_blur_act: y_f16 needs y_f16_scale, y_f16_ld >= Co, a_amax and the bound coefficients y_f16_mul > 0, y_f16_add >= 0");
    WGS_CHECK_ARG(d->precision >= 1 && d->precision <= 3, "wgs_sg2_upconv_blur_act: precision %d (split-bf16 1, fp16 schemes 2 / 3)", d->precision);
    WGS_CHECK_ARG(d->precision != 1 || d->w_lo, "wgs_sg2_upconv_blur_act: split-bf16 needs both weight planes (w_lo)");
    WGS_CHECK_ARG(d->B > 0 && d->H >= 4 && d->Ci % 32 == 0 && d->Ci > 0 && (d->Co % 64 == 0 || d->Co == 32) && d->Co > 0,
                  "wgs_sg2_upconv_blur_act: B=%d H=%d Ci=%d (%%32) Co=%d (%%64, or 32)", d->B, d->H, d->Ci, d->Co);
    WGS_CHECK_ARG(!d->noise || d->noise_w, "wgs_sg2_upconv_blur_act: noise needs noise_w");
    const long xb = (long)d->B * d->H * d->H * d->Ci * 4, yb = (long)d->B * 4 * d->H * d->H * d->Co * 4;
    const long wb = (long)d->Co * 9 * d->Ci * 2, sb = ((long)(d->B - 1) * d->a_ld + d->Ci) * 4;
    WGS_CHECK_ARG(xb < 0x7fffffffL && yb < 0x7fffffffL && wb < 0x7fffffffL && sb < 0x7fffffffL,
                  "wgs_sg2_upconv_blur_act: tensor larger than 2 GB (32-bit buffer offsets)");
    UpArgs a;
    a.x = d->x; a.w_hi = (const unsigned short*)d->w_hi; a.w_lo = (const unsigned short*)d->w_lo; a.y = d->y;
    a.a_scale = d->a_scale; a.col_scale = d->col_scale; a.bias = d->bias; a.noise = d->noise; a.noise_w = d->noise_w;
    a.kern = d->kernel4x4; a.a_amax = d->a_amax; a.a_amax2 = d->a_amax2; a.y_amax = d->y_amax;
    a.a_bound = d->a_bound > 0.f ? d->a_bound : 1.f; a.alpha = d->alpha;
    a.B = d->B; a.H = d->H; a.Ci = d->Ci; a.Co = d->Co; a.a_ld = d->a_ld; a.col_ld = d->col_ld;
    a.x_bytes = (int)xb; a.w_bytes = (int)wb; a.s_bytes = (int)sb; a.y_bytes = (int)yb;
    a.tiles_x = a.tiles_per_img = 0;
    a.y_f16 = (unsigned short*)d->y_f16; a.y_f16_scale = d->y_f16_scale; a.y_f16_bound = d->y_f16_bound;
    a.y_f16_mul = d->y_f16_mul; a.y_f16_add = d->y_f16_add; a.y_f16_ld = d->y_f16_ld; a.yh_bytes = (int)(yb / 2);
    a.w_row_stride = 9 * d->Ci;
    for (int t = 0; t < 9; ++t) a.tap_w[t] = T_W_HOST[t] * d->Ci;
    // 16 x 12-cell tiles (one 8-wave workgroup per CU) unless they would leave the chip short of workgroups: then 14 x 6-cell
    // tiles, two 4-wave workgroups per CU (they re-fetch the weights twice as often, which is what bounds the large layers)
    const long big_tiles = (long)((d->H + 15) / 16) * ((d->H + 11) / 12);        // 16 x 12-cell tiles of the 8-wave form
    const bool gh16 = !wgs_flags().up_gh8 && (wgs_flags().up_gh16 || (long)d->B * big_tiles * ((d->Co + BN - 1) / BN) >= 1536);
    // precision 3 (fp16 x2) splits the ACTIVATION operand here (Scheme<3>: same two MFMAs, same error class as the weight split of
    // the GEMM kernels): the second weight plane would double the LDS-DMA traffic that bounds this kernel.  w_lo is not read.
    // precision 1 (split-bf16 x3, fp32-class): both operands as hi + lo planes, three MFMAs per product (Scheme<0>); no operand scale
    if (d->precision == 1) { if (gh16) launch_up<0, 16>(a, (hipStream_t)stream); else launch_up<0, 8>(a, (hipStream_t)stream); }
    else if (d->precision == 2) { if (gh16) launch_up<1, 16>(a, (hipStream_t)stream); else launch_up<1, 8>(a, (hipStream_t)stream); }
    else { if (gh16) launch_up<3, 16>(a, (hipStream_t)stream); else launch_up<3, 8>(a, (hipStream_t)stream); }
    WGS_CHECK_LAUNCH("upconv_blur_kernel");
    return WGS_OK;
}
