// 16-bit-operand implicit-GEMM convolution (split-bf16 x3 / fp16 / fp16 x2, conv_scheme.h), PATCH form, for stride-1 convs
// with a small tap neighbourhood (3x3): the
// activation operand of a 256-pixel tile is staged ONCE per 32-channel chunk as the tile's input patch (tile + halo), and
// all taps of the chunk read their rows from that patch with a shifted row index.
//
//   register-staged / DMA forms: A tile (256 rows x 32 ch) loaded + split + written to LDS for EVERY tap  (9x per chunk)
//   patch form:                  (R+2) x (Wt+2) pixels loaded + split + written once per chunk            (2.0x .. 1.3x)
//
// i.e. 4.4x .. 6.8x less vector-memory, split-VALU and ds_write work on the activation side — which is what bounds the
// Cout = 128 layers at 256x256 (20 % of the training step): with only 128 output columns to amortise an activation row
// over, the register-staged kernel spent more time staging than multiplying (no-MFMA ablation: 1.87 of 2.3 ms).
//
// Tile = R x Wt output pixels of ONE sample (Wt = min(W, 128), R = 256 / Wt): every tile has one style vector.
// Weights come pre-split (wgs_split_bf16) and are copied global -> LDS by DMA per (chunk, tap), double-buffered.
// LDS: patch planes (64-B rows padded to 80 B: conflict-free ds_read_b128 with addresses that are LINEAR in the patch pixel,
// so a tap's fragment address is one add of a per-launch table entry — the fp16 form is instruction-issue bound, PMC:
// 4.5 VALU per MFMA with the XOR-swizzled image the DMA-fed weight stages still use) single-buffered — the next chunk's patch waits in
// registers during the tap steps and is written between two barriers at the chunk boundary — plus two weight stages of
// TPS taps each: one barrier per TPS taps.  The single-plane fp16 schemes spend a third of the MFMA time per tap, so they
// take a whole tap row (TPS = 3) per step to keep the MFMAs-per-barrier ratio of the split-bf16 form (48 per wave).
#include "wgs_common.h"
#include "conv_args.h"
#include "conv_epilogue.h"
#include "conv_scheme.h"

typedef wgsconv::epi_f32x16 f32x16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

using wgsconv::ConvArgs;

#ifndef WGS_PABL
#define WGS_PABL 0   // development ablations of the patch kernel (tools/build_abl.sh): 1 no weight DMA, 2 no patch global loads,
                     // 3 no patch LDS stores, 4 no MFMA, 5 no fragment LDS reads, 6 no barriers in the tap steps, 7 no epilogue stores
#endif

constexpr int BK = 32;
constexpr int ROW = 64;                  // bytes per LDS row of a weight plane (32 x 16 bit; filled by DMA: unpadded, swizzled)
constexpr int PROW = 80;                 // bytes per LDS row of a patch plane (filled from registers: padded, linear)
constexpr int OOB = (int)0x80000000;
// patch pixels the LDS image holds: 256-pixel tiles 4 x 130 = 520, 128-pixel tiles 4 x 66 = 264 (3x3 taps)
constexpr int pmax_of(int bm) { return bm == 256 ? 528 : 272; }

typedef __attribute__((address_space(3))) unsigned char lds_byte;

struct PatchGeom {       // uniform per launch
    int Wt, R, PW, PH, tiles_x, tiles_per_img, dy_min, dx_min;
    // flat != 0: a tile is BM CONSECUTIVE pixels of the (Hg x Wg) output grid of one sample (any Wg, e.g. the 65 x 65 and
    // 129 x 129 grids of the up-conv phases) and its patch is the full-width band of input rows those pixels touch;
    // flat == 0: rectangular R x Wt tiles (the 256-wide maps, whose full-width band would not fit)
    int flat, Wg, HW;
    unsigned pw_magic, wg_magic, wt_magic;   // wgs_div_magic of PW, Wg, Wt
    int npl;             // float4 patch loads per thread and chunk that the PH x PW patch needs (<= the kernel's NPL)
    int tapoff[16];      // byte offset of tap t's rows in a patch plane: ((dy - dy_min) * PW + (dx - dx_min)) * PROW
};

// BM = 256 (8 waves, one workgroup per CU) or 128 (4 waves, 100 KB less LDS: two or three workgroups per CU, whose barriers,
// patch stores and epilogues overlap each other's MFMAs — the better shape when K is short, i.e. Cin = 128).
// NTF: number of taps when it is known at compile time (9: every 3x3 conv), else 0.  With it the tap loop is unrolled and the
// per-tap table entries (LDS row offsets, weight slab offsets) become loop-invariant scalar loads hoisted out of the chunk loop:
// a scalar load inside a step shares lgkmcnt with the step's ds_reads, so waiting for it drained the fragment reads in flight.
// XF16: the activation operand arrives as its fp16 plane (ConvArgs.a_hi = wgs_conv_desc.x_f16, written by the producing kernel
// already multiplied by its style vector and the power-of-two operand scale): a staging element is 16 bytes = EIGHT channels of a
// patch pixel, loaded and written to LDS as it is — no style multiply, no scale, no conversion, half the load bytes and
// instructions.  (Single-plane A only: schemes 1 and 2.)
// RGB: ToRGB in the epilogue (wgs_conv_desc.rgb_out) — its own instantiation, so that the epilogue's extra registers do not touch the
// plain kernel's allocation (170 VGPRs at three workgroups per CU)
template <int SCH, int BM, int BN, int WAVES_M, int WAVES_N, int TPS, int NTF, bool XF16, bool RGB = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, BM == 128 ? (TPS == 1 && SCH == 1 ? 3 : 2) : 1) void igemm_patch_kernel(const ConvArgs p, const PatchGeom g) {
    typedef wgsconv::Scheme<SCH> SC;
    typedef typename SC::frag frag;
    constexpr int NA = SC::NA, NB = SC::NB;
    constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW;
    constexpr int PMAX = pmax_of(BM);
    static_assert(!XF16 || NA == 1, "an fp16 activation plane is a single-plane A operand");
    constexpr int EPP = XF16 ? 4 : 8;                 // 16-byte staging elements per patch pixel and chunk (8 fp16 / 4 fp32 channels each)
    constexpr int EPS = XF16 ? 2 : 3;                 // log2(EPP)
    constexpr int NPL = (PMAX * EPP + NT - 1) / NT;   // 16-byte patch loads per thread and chunk at most (9); g.npl are issued
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
    constexpr int P_BYTES = PMAX * PROW;                // one patch plane
    constexpr int B_BYTES = BN * ROW;                   // one weight plane of one tap
    constexpr int B_TAP = NB * B_BYTES;
    constexpr int B_STAGE = TPS * B_TAP;
    constexpr int BI = BN / 16 / NW;                    // 16-row DMA instructions per wave and plane
    static_assert(BI >= 1 && BI * 16 * NW == BN, "tile / wave count mismatch");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* patch = smem_b;                      // hi (| lo)
    unsigned char* bst = smem_b + NA * P_BYTES;         // two weight stages

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ntn = p.Co / BN;
    // XCD-aware order over the (tile-major, column-tile minor) list, as in the other kernels
    int bid;
    {
        const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int qn = nb >> 3, rn = nb & 7;
        bid = xcd * qn + min(xcd, rn) + slot;
    }
    const int tile = bid / ntn, n0 = (bid % ntn) * BN;
    const int b = tile / g.tiles_per_img;
    const int trem = tile - b * g.tiles_per_img;
    int ty0, tx0;                 // origin of the tile in the output grid (flat: first grid row, column 0)
    const int tile_m0 = trem * BM;      // flat: first grid pixel of the tile
    if (g.flat) { ty0 = tile_m0 / g.Wg; tx0 = 0; }
    else { ty0 = (trem / g.tiles_x) * g.R; tx0 = (trem % g.tiles_x) * g.Wt; }
    // tile row r -> grid pixel (gy, gx) relative to (ty0, tx0); false = past the sample's last pixel (flat tiles only)
    auto row_of = [&](int r, int& ry, int& rx) {
        if (g.flat) {
            const int m = tile_m0 + r;
            const int mm = m < g.HW ? m : g.HW - 1;
            const int gy = wgs_div_fast(mm, g.wg_magic);
            ry = gy - ty0; rx = mm - gy * g.Wg;
            return m < g.HW;
        }
        ry = wgs_div_fast(r, g.wt_magic); rx = r - ry * g.Wt;
        return true;
    };

    const __amdgpu_buffer_rsrc_t rx = XF16 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.a_hi), 0, p.x_bytes, 0x00020000)
                                           : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbh = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w_hi), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbl = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w_lo), 0, p.w_bytes, 0x00020000);

    // ---- patch staging: element e = tid + j*NT -> patch pixel e / EPP, 16-byte piece q = e % EPP (= tid % EPP for every j)
    const int q = tid & (EPP - 1);
    const int npatch = g.PH * g.PW;
    int p_goff[NPL];        // byte offset of (pixel, q) in x for chunk 0, or OOB
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        if (j >= g.npl) break;
        const int pp = (tid + j * NT) >> EPS;
        const int pr = wgs_div_fast(pp, g.pw_magic), pc = pp - pr * g.PW;
        const int iy = ty0 + g.dy_min + pr, ix = tx0 + g.dx_min + pc;
        const bool v = pp < npatch && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        p_goff[j] = v ? ((b * p.Hi + iy) * p.Wi + ix) * p.Ci * (XF16 ? 2 : 4) + q * 16 : OOB;
    }
    // byte offset of staging element j in the hi patch plane: linear in j (one base register + immediates)
    const int p_lbase = (tid >> EPS) * PROW + q * (XF16 ? 16 : 8);
    auto p_loff_of = [&](int j) { return ((tid + j * NT) >> EPS) < PMAX ? p_lbase + j * (NT / EPP) * PROW : -1; };
    const float* sc_ptr = (p.a_scale && !XF16) ? p.a_scale + (size_t)b * p.a_ld + q * 4 : nullptr;
    constexpr bool PIPE = (NA == 1 && NB == 1);       // single-plane (fp16) form: hand-pipelined steps, explicit waits
    const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a_scale ? p.a_scale : p.x), 0, p.a_scale ? p.s_bytes : 0, 0x00020000);
    u32x4 pr_[NPL];
    const int npl = g.npl;      // staging slots past the patch's own size are skipped (uniform branch), not loaded-as-zero
    // fp16 schemes: dynamic power-of-two operand scale (conv_scheme.h), folded into the style vector / undone in the epilogue
    float op_mult = 1.f, op_inv = 1.f;
    if (SCH != 0) wgsconv::operand_scale(p.a_amax, p.a_amax2, p.a_bound, op_mult, op_inv);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
    const int cpt = p.Ci / BK;
    auto load_patch = [&](int c) {
        const int cbyte = c < cpt ? c * (BK * (XF16 ? 2 : 4)) : OOB;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            if (j >= npl) break;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (WGS_PABL != 2) v = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)((unsigned)p_goff[j] + (unsigned)cbyte), 0, 0);
            else asm volatile("" : "+v"(v));
            pr_[j] = v;
        }
        if (XF16) return;
        // (no use of the loaded style vector here: touching it would make the compiler wait for it — and, vmcnt being in-order,
        // for the nine patch loads in front of it — at the top of every chunk; the operand scale is applied in store_patch)
        if (PIPE) {      // the style vector through the same buffer path as the patch (range-checked past the end)
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsc, (sc_ptr && c < cpt) ? (b * p.a_ld + q * 4 + c * BK) * 4 : OOB, 0, 0);
            if (sc_ptr && c < cpt) sc = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        } else if (sc_ptr && c < cpt) sc = *reinterpret_cast<const float4*>(sc_ptr + c * BK);
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            if (j >= npl) break;
            if (XF16) {
                const int lo16 = p_loff_of(j);
                if (WGS_PABL == 3) { asm volatile("" :: "v"(pr_[j].x), "v"(pr_[j].y)); continue; }
                if (lo16 >= 0) *reinterpret_cast<u32x4*>(patch + lo16) = pr_[j];
                continue;
            }
            float4 v = make_float4(__uint_as_float(pr_[j].x), __uint_as_float(pr_[j].y), __uint_as_float(pr_[j].z), __uint_as_float(pr_[j].w));
            v.x = __fmul_rn(v.x, sc.x); v.y = __fmul_rn(v.y, sc.y); v.z = __fmul_rn(v.z, sc.z); v.w = __fmul_rn(v.w, sc.w);
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));   // keep the rounded product (no fma into the residual)
            if (SCH != 0) { v.x *= op_mult; v.y *= op_mult; v.z *= op_mult; v.w *= op_mult; }   // power of two: exact
            const f32x4 f = {v.x, v.y, v.z, v.w};
            uint2 h, l;
            SC::cvt4(f, h, l);
            const int lo_ = p_loff_of(j);
            if (WGS_PABL == 3) { asm volatile("" :: "v"(h.x), "v"(h.y)); continue; }
            if (lo_ >= 0) {
                *reinterpret_cast<uint2*>(patch + lo_) = h;
                if (NA == 2) *reinterpret_cast<uint2*>(patch + P_BYTES + lo_) = l;
            }
        }
    };

    // ---- weight DMA: this lane feeds LDS row 16*instr + lane/4, slot lane%4
    const int lrow = lane >> 2, slot = lane & 3;
    int b_off[BI];
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int row = (wave * BI + j) * 16 + lrow;
        const int lc = slot ^ ((row >> 2) & 3);
        b_off[j] = (int)((long)(n0 + row) * p.w_row_stride + lc * 8) * 2;
    }
    auto issue_b = [&](int stage, int c, int t) {        // weights of (chunk c, taps t .. t+TPS-1) -> stage; past the end: zeros
#pragma unroll
        for (int u = 0; u < TPS; ++u) {
            const unsigned delta = c < cpt ? (unsigned)((p.tap_w[t + u] >> 1) + c * (BK * 2)) : (unsigned)OOB;
            lds_byte* st = (lds_byte*)(bst + stage * B_STAGE + u * B_TAP);
#pragma unroll
            for (int j = 0; j < BI; ++j) {
                const int off = (int)((unsigned)b_off[j] + delta);
                lds_byte* d = st + (wave * BI + j) * 16 * ROW;
                if (WGS_PABL == 1) { asm volatile("" :: "v"(off)); continue; }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rbh, d, 16, off, 0, 0, 0);
                if (NB == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rbl, d + B_BYTES, 16, off, 0, 0, 0);
            }
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
    int pa0[TM];             // byte address (in a patch plane) of this lane's fragment row for tap offset 0, k-step 0
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * WM + i * 32 + l31;
        int ty, tx;
        row_of(r, ty, tx);
        pa0[i] = (ty * g.PW + tx) * PROW + lh * 16;
    }
    const int bswz = (l31 >> 2) & 3;
    const int b_rd = (wn * WN + l31) * ROW;
    const int bk0 = ((0 + lh) ^ bswz) << 4, bk1 = ((2 + lh) ^ bswz) << 4;

    auto mma_tap = [&](int stage, int u, int tapoff) {
        const unsigned char* bb = bst + stage * B_STAGE + u * B_TAP + b_rd;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag bf[TN][NB];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < NB; ++pl)
                    bf[j][pl] = *reinterpret_cast<const frag*>(bb + pl * B_BYTES + j * 32 * ROW + (ks ? bk1 : bk0));
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned char* pa = patch + pa0[i] + tapoff + ks * 32;
                frag af[NA];
#pragma unroll
                for (int pl = 0; pl < NA; ++pl) af[pl] = *reinterpret_cast<const frag*>(pa + pl * P_BYTES);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = SC::mma(af, bf[j], acc[i][j]);
            }
        }
    };

    // Single-plane schemes: one A fragment feeds only TN MFMAs (64 .. 128 matrix-pipe cycles) — less than an LDS round trip,
    // and the compiler's own schedule is {ds_read A; wait; TN MFMAs} per fragment, i.e. the latency is exposed every time.
    // So the step is software-pipelined by hand over its 2*TPS (tap, k-step) groups: the TM + TN fragment reads of group
    // g+1 are issued before the TM*TN MFMAs of group g (two register sets, pinned with sched_barrier).
    auto mma_step = [&](int stage, int t0) {
        constexpr int G = 2 * TPS;
        int tapoff[TPS];
#pragma unroll
        for (int u = 0; u < TPS; ++u) tapoff[u] = g.tapoff[t0 + u];
        frag af[2][TM], bf[2][TN];
        auto load_group = [&](int gi, frag* a, frag* bq) {
            const int u = gi >> 1, ks = gi & 1;
            const unsigned char* bb = bst + stage * B_STAGE + u * B_TAP + b_rd + (ks ? bk1 : bk0);
            if (WGS_PABL == 5) {
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "=v"(bq[j]));
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" : "=v"(a[i]));
                return;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) bq[j] = *reinterpret_cast<const frag*>(bb + j * 32 * ROW);
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const frag*>(patch + pa0[i] + tapoff[u] + ks * 32);
        };
        load_group(0, af[0], bf[0]);
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            if (gi + 1 < G) load_group(gi + 1, af[(gi + 1) & 1], bf[(gi + 1) & 1]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (WGS_PABL == 4) { asm volatile("" :: "v"(af[gi & 1][i]), "v"(bf[gi & 1][j])); continue; }
                    acc[i][j] = SC::mma(&af[gi & 1][i], &bf[gi & 1][j], acc[i][j]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- main loop: chunk outer, tap inner
    load_patch(0);
    issue_b(0, 0, 0);
    store_patch();
    __syncthreads();
    int stage = 0;
    for (int c = 0; c < cpt; ++c) {
        if (!PIPE) load_patch(c + 1);            // waits in registers through the tap steps (OOB past the last chunk)
        const int ntaps = NTF ? NTF : p.ntaps;
        auto tap_step = [&](int t) {
            const bool last = (t + TPS >= ntaps);
            issue_b(stage ^ 1, last ? c + 1 : c, last ? 0 : t + TPS);
            if (PIPE) {
                // The next chunk's patch loads go out behind the first step's weight DMAs.  NOTE (measured, round 2): a COUNTED
                // wait here — vmcnt(NPL + 1), "DMAs landed, patch loads still in flight" — is WRONG on gfx950: LDS-DMA loads
                // (buffer_load ... lds) and VGPR-destination loads do not retire in issue order relative to each other, so the
                // count does not prove that the older DMAs have landed (sporadic stale weight stages: errors of 0.2, NaNs; found
                // by the shared-gate gradient test, reproduced in isolation on the 128->128 @256^2 dgrad).  Everything is drained.
                if (t == 0) load_patch(c + 1);
                mma_step(stage, t);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (WGS_PABL != 6) __builtin_amdgcn_s_barrier();
                // the bare s_barrier intrinsic does not order memory operations for the compiler (unlike __syncthreads(), whose
                // fences would drain the patch loads): pin the next step's DMA issue and fragment reads behind it
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int u = 0; u < TPS; ++u) mma_tap(stage, u, g.tapoff[t + u]);
                if (WGS_PABL != 6) __syncthreads();  // weight stage swap; after the last tap also: patch no longer read
            }
            stage ^= 1;
        };
        if (NTF) {
#pragma unroll
            for (int t = 0; t < NTF; t += TPS) tap_step(t);
        } else {
            for (int t = 0; t < ntaps; t += TPS) tap_step(t);
        }
        if (c + 1 < cpt) {
            store_patch();
            __syncthreads();
        }
    }

    // ---- epilogue (contract of conv_igemm.hip; rows are the R x Wt tile pixels of sample b)
    int* r_pix = reinterpret_cast<int*>(smem_b);
    int* r_b = r_pix + BM;
    float* r_nz = reinterpret_cast<float*>(r_b + BM);
    int* r_add = reinterpret_cast<int*>(r_nz + BM);
    if (tid < BM) {
        int ty, tx;
        const bool ok = row_of(tid, ty, tx);
        const int oy = (ty0 + ty) * p.osy + p.oy0, ox = (tx0 + tx) * p.osx + p.ox0;
        const int hw = oy * p.Wo + ox;
        r_pix[tid] = ok ? b * p.Ho * p.Wo + hw : -1;
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

template <int SCH, int BM, int BN, int WAVES_M, int WAVES_N, int TPS, int NTF, bool XF16>
void launch_patch_n(const ConvArgs& a, const PatchGeom& g, int nblocks, hipStream_t st) {
    typedef wgsconv::Scheme<SCH> SC;
    const size_t sm = (size_t)SC::NA * pmax_of(BM) * PROW + (size_t)2 * TPS * SC::NB * BN * ROW;
    auto k = igemm_patch_kernel<SCH, BM, BN, WAVES_M, WAVES_N, TPS, NTF, XF16>;
    wgs_note_kernel("igemm_patch_kernel<%d, %d, %d, %d, %d, %d, %d, %s>", SCH, BM, BN, WAVES_M, WAVES_N, TPS, NTF, XF16 ? "true" : "false");
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    WGS_LAUNCH(k, dim3((unsigned)nblocks), dim3(64 * WAVES_M * WAVES_N), sm, st, a, g);
}
template <int SCH, int BM, int BN, int WAVES_M, int WAVES_N, int TPS>
void launch_patch_t(const ConvArgs& a, const PatchGeom& g, int nblocks, hipStream_t st) {
    // (256-row tiles only: +2.5 %; the three-per-CU 128-row tiles lose 7 % to the registers the unrolled steps take)
    const bool ntf9 = a.ntaps == 9 && SCH != 0 && BM == 256 && !wgs_flags().patch_ntf0;
    if constexpr (SCH == 1 || SCH == 2) {
        if (a.a_hi) {     // producer-written fp16 activation plane
            if constexpr (SCH == 1 && BM == 128 && BN == 128 && TPS == 1) {
                if (a.rgb_out) {
                    typedef wgsconv::Scheme<SCH> SC;
                    const size_t sm = (size_t)SC::NA * pmax_of(BM) * PROW + (size_t)2 * TPS * SC::NB * BN * ROW;
                    auto k = igemm_patch_kernel<SCH, BM, BN, WAVES_M, WAVES_N, TPS, 0, true, true>;
                    wgs_note_kernel("igemm_patch_kernel<%d, %d, %d, %d, %d, %d, 0, true, true>", SCH, BM, BN, WAVES_M, WAVES_N, TPS);
                    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
                    WGS_LAUNCH(k, dim3((unsigned)nblocks), dim3(64 * WAVES_M * WAVES_N), sm, st, a, g);
                    return;
                }
            }
            if (ntf9) launch_patch_n<SCH, BM, BN, WAVES_M, WAVES_N, TPS, 9, true>(a, g, nblocks, st);
            else launch_patch_n<SCH, BM, BN, WAVES_M, WAVES_N, TPS, 0, true>(a, g, nblocks, st);
            return;
        }
    }
    if (ntf9) launch_patch_n<SCH, BM, BN, WAVES_M, WAVES_N, TPS, 9, false>(a, g, nblocks, st);
    else launch_patch_n<SCH, BM, BN, WAVES_M, WAVES_N, TPS, 0, false>(a, g, nblocks, st);
}

// tile shape x scheme dispatch; the fp16 schemes take a tap row per barrier when the tap count allows it
template <int BM, int BN, int WAVES_M, int WAVES_N>
void launch_patch(const ConvArgs& a, const PatchGeom& g, int nblocks, hipStream_t st) {
    const bool tps3 = a.ntaps % 3 == 0 && !wgs_flags().patch_tps1;
    if (a.sch == 0) launch_patch_t<0, BM, BN, WAVES_M, WAVES_N, 1>(a, g, nblocks, st);
    // (128-row fp16 tiles: one tap per stage = 38 KB of LDS and 154 VGPRs -> THREE workgroups per CU; with K = 4 chunks the
    // prologue / epilogue of a tile is a third of its life, and a third resident workgroup hides more of it than the longer
    // steps save: 128->128 @256^2 0.95 -> 0.90 ms)
    else if (a.sch == 1) { if (tps3 && BM != 128) launch_patch_t<1, BM, BN, WAVES_M, WAVES_N, 3>(a, g, nblocks, st); else launch_patch_t<1, BM, BN, WAVES_M, WAVES_N, 1>(a, g, nblocks, st); }
    else { if (tps3 && BN == 128 && BM == 256) launch_patch_t<2, BM, BN, WAVES_M, WAVES_N, 3>(a, g, nblocks, st); else launch_patch_t<2, BM, BN, WAVES_M, WAVES_N, 1>(a, g, nblocks, st); }
}

}  // namespace

namespace wgsconv {

// Returns 0 when the launch was taken: stride-1, same-size conv (Hg = Hi = Ho ...), pre-split weights, taps within a
// neighbourhood whose patch fits, power-of-two width >= 32, Cout a multiple of 128, enough tiles for the chip.
int launch_patch_bf16x3(const ConvArgs& a0, hipStream_t st) {
    const ConvArgs& a = a0;
    if (!a.w_hi || (!a.w_lo && a.sch != 1) || a.ups || a.isy != 1 || a.isx != 1) return 1;
    if (a.a_hi && (a.sch == 0 || a.a_scale)) return 1;      // an fp16 activation plane: fp16 schemes, style already folded in by the producer
    const int epp = a.a_hi ? 4 : 8;                         // 16-byte staging elements per patch pixel (XF16: 8 fp16 channels each)
    if (a.Ci % 32 || a.Co % 128 || a.ntaps < 2 || a.ntaps > 16 || a.Wg < 2) return 1;
    int dy0 = 127, dy1 = -127, dx0 = 127, dx1 = -127;
    for (int t = 0; t < a.ntaps; ++t) {
        dy0 = a.dy[t] < dy0 ? a.dy[t] : dy0; dy1 = a.dy[t] > dy1 ? a.dy[t] : dy1;
        dx0 = a.dx[t] < dx0 ? a.dx[t] : dx0; dx1 = a.dx[t] > dx1 ? a.dx[t] : dx1;
    }
    const int Wg = a.Wg, Hg = a.Hg, HW = a.Hg * a.Wg;
    // tile shape: 256 pixels (8 waves) x 256 or 128 columns in general; 128 pixels (4 waves, two workgroups per CU) x 128
    // columns when the K loop is short (Cin <= 128 with Cout = 128) or when the larger tiles would leave CUs idle.
    // Geometry: consecutive grid pixels with a full-width patch band when that fits (grids up to ~130 wide, any size —
    // the sub-pixel phases' 65 x 65 / 129 x 129 grids included), else rectangular tiles of a power-of-two-wide grid.
    PatchGeom g;
    int bm = 0, bn = 0, nblocks = 0;
    auto try_shape = [&](int tbm, int tbn) {
        if (bm || a.Co % tbn) return;
        PatchGeom t;
        t.dy_min = dy0; t.dx_min = dx0; t.Wg = Wg; t.HW = HW;
        const int rows_max = (tbm + Wg - 2) / Wg + 1;            // grid rows a run of tbm consecutive pixels can touch
        // Power-of-two grids: COMPACT rectangular tiles (8 rows x 32 or 16 pixels).  The patch of a tile is (R+2) x (Wt+2)
        // pixels, staged (loaded, scaled, converted, written to LDS) once per 32-channel chunk: 10 x 34 = 340 pixels for a
        // 256-pixel tile against 4 x 130 = 520 for the 2 x 128 row pair (and 7 x 66 = 462 for the flat band of a 64-wide map),
        // 10 x 18 = 180 against 4 x 66 = 264 for a 128-pixel tile: a third less staging work and patch traffic.  (Measured: the
        // launch times do not move — the tiles wait on load latency, not on staging instructions, DESIGN.md 3.5 — it is
        // kept for the L2 / HBM patch traffic it saves.)
        const int cwt = tbm == 256 ? 32 : 16, crr = tbm / cwt;
        if (!wgs_flags().patch_wide && Wg >= cwt && !(Wg & (Wg - 1)) && Hg % crr == 0) {
            t.flat = 0; t.Wt = cwt; t.R = crr; t.PH = crr + dy1 - dy0; t.PW = cwt + dx1 - dx0;
            if (t.PH * t.PW > pmax_of(tbm)) return;
            t.tiles_x = Wg / cwt; t.tiles_per_img = (Hg / crr) * t.tiles_x;
        } else if ((rows_max + dy1 - dy0) * (Wg + dx1 - dx0) <= pmax_of(tbm)) {
            t.flat = 1; t.Wt = Wg; t.R = rows_max; t.PH = rows_max + dy1 - dy0; t.PW = Wg + dx1 - dx0;
            t.tiles_x = 1; t.tiles_per_img = (HW + tbm - 1) / tbm;
            if ((long)t.tiles_per_img * tbm * 100 > (long)HW * 113) return;      // > 13 % of the rows would be padding
        } else {
            if (Wg < 16 || (Wg & (Wg - 1))) return;
            const int wt = tbm == 256 ? (Wg < 128 ? Wg : 128) : (Wg < 64 ? Wg : 64);
            const int r = tbm / wt;
            if (r < 1 || Hg % r) return;
            t.flat = 0; t.Wt = wt; t.R = r; t.PH = r + dy1 - dy0; t.PW = wt + dx1 - dx0;
            if (t.PH * t.PW > pmax_of(tbm)) return;
            t.tiles_x = Wg / wt; t.tiles_per_img = (Hg / r) * t.tiles_x;
        }
        t.pw_magic = wgs_div_magic(t.PW); t.wg_magic = wgs_div_magic(Wg); t.wt_magic = wgs_div_magic(t.Wt);
        t.npl = (t.PH * t.PW * epp + tbm - 1) / tbm;         // threads per workgroup = tile rows (64 x 8 or 64 x 4)
        const int nb = a.B * t.tiles_per_img * (a.Co / tbn);
        if (nb < 200) return;
        bm = tbm; bn = tbn; nblocks = nb; g = t;
        for (int i = 0; i < 16; ++i) g.tapoff[i] = i < a.ntaps ? ((a.dy[i] - dy0) * t.PW + (a.dx[i] - dx0)) * PROW : 0;
    };
    if (a.Co == 128 && a.Ci <= 128 && (!wgs_flags().patch_bm256 || a.rgb_out)) {
        if (a.a_hi && a.sch == 1 && !wgs_flags().patch_nodma) {     // fp16 plane in plain fp16: the all-DMA form of the 128 x 128 tile
            ConvArgs b = a;
            b.w_bytes = a.w_bytes / 2;
            if (!launch_patch_dma(b, st)) return 0;
        }
        try_shape(128, 128);
    }
    if (a.rgb_out && !(bm == 128 && bn == 128 && a.a_hi && a.sch == 1)) return 1;      // ToRGB epilogue: the 128 x 128 fp16-plane tile only
    try_shape(256, 256);
    try_shape(256, 128);
    try_shape(128, 128);
    if (!bm) return 1;
    ConvArgs b = a;
    b.w_bytes = a.w_bytes / 2;          // extents of the 16-bit weight planes (x_bytes: fp32, or the fp16 plane's when a.a_hi)
    if (bm == 256 && bn == 256) launch_patch<256, 256, 2, 4>(b, g, nblocks, st);
    else if (bm == 256) launch_patch<256, 128, 4, 2>(b, g, nblocks, st);
    else launch_patch<128, 128, 2, 2>(b, g, nblocks, st);
    return 0;
}

}  // namespace wgsconv
