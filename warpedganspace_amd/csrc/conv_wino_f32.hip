// 3x3 stride-1 convolution in fp32 through the Winograd minimal-filtering form F(2x2, 3x3) on the fp32 matrix cores: 16 multiplies
// per 2x2 output block and input channel instead of 36, fp32 operands, fp32 transforms, fp32 accumulate — 3e-6 against fp64
// convolutions, no wider than the direct fp32 kernel (tests/test_conv_wino_gpu.py); the op it stands for: models/StyleGAN2/model.py:187-228.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          d: 4x4 input patch (stride 2), g: 3x3 filter, Y: 2x2 outputs
//
// which over channels is 16 independent GEMMs   M_pos[tile, co] = sum_ci V_pos[tile, ci] * U_pos[ci, co],   pos = (xi, nu) in 4x4.
//
// ONE kernel, nothing but x, U and y touches HBM:
//   * U = G g G^T is transformed once per (frozen) weight tensor by wino_weight_kernel, straight into the B-operand fragment order
//     of the main kernel: a wave needs only the U of its own two positions, so its fragments are plain coalesced 1 KB global loads
//     into registers — U never passes through LDS.
//   * a workgroup (8 waves) owns 32 * TI Winograd tiles of one sample x 32 * TJ output channels and ALL 16 positions: wave w
//     accumulates the four positions of row xi = w >> 1 for half of the channel blocks (128 accumulator registers per lane).
//   * per 16-channel chunk: thread (tile, channel quad, patch column) loads its 4 patch rows (16 B each, zero padding = buffer range
//     check), does the column pass B^T d in registers and the row pass across the 4 lanes of its quad with DPP quad_perm, scales by the
//     style and writes its 4 positions to LDS (two such tasks per thread).  Patch loads fly for three quarters of a chunk, B fragments
//     for half a chunk; loads and stores are issued inside the MFMA slots (conv_nt_kernel.inc's scheme); A fragments are one
//     ds_read_b128 per 8 MFMAs (k pairing of conv_scheme.h Scheme<4>); one barrier per 64 MFMAs of a wave.
//   * epilogue: the row half of A^T . A in registers (a wave holds all four nu of its xi), the 8 remaining blocks per (tile, channel)
//     through LDS in one pass (<= 147 KB), every thread finishes four tiles of one channel, then demodulation / noise / bias /
//     leaky-relu as conv_epilogue.h and stores 128-byte channel runs.
// LDS rows are 64 B (16 channels), conflict-free by an XOR swizzle of row and 16-byte slot (see the staging role below).
#include "wgs_common.h"
#include "../../include/wgs.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef WGS_WINO_SWZ
#define WGS_WINO_SWZ 1            // row shift of the LDS swizzle term: 1 = conflict-free for runs of 8 lanes (shipped since round 3), 2 = conflict-free for
                                  // ds_read_b128's real lane groups.  Same bits; measured round 5 (tools/bench_wino.py, two rounds on one box, us per
                                  // launch pair): 512->512 @64^2 2131-2148 vs 2178-2281 with 2; 256->256 @128^2 2211-2224 vs 2237-2382; 128->128
                                  // @256^2 2400-2406 vs 2430-2570 — one 16-byte read per 8 fp32 MFMAs (512 matrix-pipe cycles): the LDS is idle
                                  // either way, and the conflict-free form is not faster.
#endif
constexpr int KC = 16;            // input channels per chunk
constexpr int OOB = (int)0x80000000;

// Workgroup tile: 32 * TI Winograd tiles (8 columns x 4 * TI rows of 2x2 outputs = 16 x 8 * TI pixels) x 32 * TJ output channels.
// The input transform costs the same per tile whatever TJ is, the MFMAs grow with it: (TI, TJ) = (1, 4) where Cout % 128 == 0
// (0.5 vector instructions per MFMA), (2, 2) for Cout % 64 == 0.
template <int TI, int TJ>
struct Cfg {
    static constexpr int TB = 32 * TI, BN = 32 * TJ, PH = 8 * TI;
    static constexpr int PS = TB * KC * 4;             // one position of a staged chunk: TB rows of 64 B
    static constexpr int STAGE = 16 * PS;              // one staged chunk of V (64 KB / 32 KB)
    static constexpr int EPI_ROW = TB * 4 + 16;        // one (xi, j2, n) row of the epilogue exchange: TB tiles + 16 B (bank spread)
    static constexpr int EPI_BYTES = 8 * BN * EPI_ROW; // (xi, output column j2) x channel rows: one pass
    static constexpr int SMEM = EPI_BYTES > 2 * STAGE ? EPI_BYTES : 2 * STAGE;
    static constexpr int U_CHUNK = 16 * BN * KC * 4;   // bytes of U per (channel block, chunk)
};

struct WinoArgs {
    const float* x;
    const float* U;
    float* y;
    const float* a_scale;
    const float* col_scale;
    const float* bias;
    const float* noise;
    const float* noise_w;
    float* y_amax;
    double* col_stats;           // wgs_conv_desc.col_stats: sum y / sum y^2 per output channel into the BatchNorm scratch (conv_epilogue.h), or null
    int B, H, W, Ci, Co, a_ld, col_ld;
    float alpha, act_slope, gain;
    int u_order;                 // 1: an XCD's workgroups share ONE channel block of U (see the kernel's workgroup order)
};

struct WinoTaps { int w_of[9]; };      // weight slab index of spatial tap (ky, kx), -1 = absent

// voff: per-lane byte offset (VGPR, fixed for the whole kernel; OOB = beyond any buffer -> the load returns 0), soff: uniform byte
// offset of the chunk / fragment (SGPR): no vector address arithmetic inside the main loop
__device__ __forceinline__ f32x4 buf_load4(const __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return __builtin_bit_cast(f32x4, v);
}
// The transform's arithmetic written as the instructions it is (the backend splits 4-vectors into scalars and keeps a quad permute
// as a separate v_mov_b32_dpp): packed fp32 adds / multiplies on register pairs ...
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
// ... and v += s * (v of lane {2, 2, 1, 1}[lane & 3] of the quad), in place on the four values of a loaded float4 (s_nop 1: a VGPR
// written by the preceding VALU instruction may not be read through DPP for two cycles)
__device__ __forceinline__ void quad_row_pass(f32x4& v, float s) {
    float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %1, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %2, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %3, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf"
        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(s));
    v = (f32x4){v0, v1, v2, v3};
}

// U[pos] = G g G^T of every (co, ci) pair, written as the main kernel's B-operand fragments:
// [Co/BN][Ci/16][pos][k group g of 8][column block j of 32][lane = 32 * (k half) + column][4 consecutive k].
// Positions with nu = 3 carry a minus sign: the kernel's row pass produces -(T1 - T3) there (own minus permuted column, the form
// the other three columns have), and the product of the two signs is +.
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Ci, int Co, int tj,
                                                          long w_row_stride, long w_tap_stride, const WinoTaps tp) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Ci * Co) return;
    const int ci = (int)(idx % Ci), co = (int)(idx / Ci);
    float g[3][3];
#pragma unroll
    for (int s = 0; s < 9; ++s) g[s / 3][s % 3] = tp.w_of[s] >= 0 ? w[(size_t)co * w_row_stride + (size_t)tp.w_of[s] * w_tap_stride + ci] : 0.f;
    float t[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        t[0][c] = g[0][c];
        t[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
        t[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
        t[3][c] = g[2][c];
    }
    float u[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        u[r][0] = t[r][0];
        u[r][1] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
        u[r][2] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
        u[r][3] = -t[r][2];
    }
    const int bn = 32 * tj, nchunks = Ci / KC;
    const int nb = co / bn, n = co % bn, chunk = ci / KC, kk = ci % KC;
    const int lane = ((kk >> 2) & 1) * 32 + (n & 31);
    float* base = U + ((size_t)nb * nchunks + chunk) * (16 * bn * KC);
#pragma unroll
    for (int pos = 0; pos < 16; ++pos)
        base[(((pos * 2 + (kk >> 3)) * tj + (n >> 5)) * 64 + lane) * 4 + (kk & 3)] = u[pos >> 2][pos & 3];
}

// NW = 8: one workgroup per CU, wave = (position row xi, half of the channel blocks); NW = 4 (TI = 1, TJ = 2): two workgroups per CU, wave =
// position row xi with both channel blocks — one workgroup's prologue / epilogue runs under the other's MFMAs (the short-K layers).
template <int TI, int TJ, bool STY, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void wino_f32_kernel(const WinoArgs p) {
    typedef Cfg<TI, TJ> C;
    constexpr int NT = 64 * NW;
    constexpr int NTASK = 32 * TI * 16 / NT;          // staging tasks (tile, channel quad, column) per thread: 1 or 2
    constexpr int PS = C::PS, STAGE = C::STAGE, EPI_ROW = C::EPI_ROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int ntn = p.Co / C::BN, tbx = p.W >> 4, tby = p.H / C::PH;
    // XCD-aware order (hardware sends workgroup i to XCD i % 8): every XCD gets a contiguous range of the (pixel block major,
    // channel block minor) list, so the channel blocks of a pixel block share their input patch in that XCD's L2
    int bid;
    {
        const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, qn = nb >> 3, rn = nb & 7;
        bid = xcd * qn + min(xcd, rn) + slot;
    }
    int tmi = bid / ntn, nb0 = bid - tmi * ntn;
    if (p.u_order) {
        // (round 6; WGS_WINO_UORD=0 switches it off) channel block FIXED per XCD instead: XCD x runs channel block x % ntn over the pixel blocks of
        // part x / ntn — its resident workgroups stream ONE 16 * Ci * BN-float slab of U (4.2 MB at 512 channels: the XCD's L2) instead of ntn of
        // them, and a pixel block's input patch is fetched by ntn XCDs instead of one.  512 -> 512 @64^2, B = 32: FETCH_SIZE 2.39 -> 1.58 GB, L2
        // misses 21.2 M -> 14.7 M, time 2 127 -> 2 122 us (the kernel is not traffic-bound: VERDICT r5 asked for the bytes, not the time).  The
        // launcher sets it only where 8 % ntn == 0 and the counts divide.
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int per = (int)gridDim.x >> 3;             // workgroups per XCD = pixel blocks per part
        nb0 = xcd % ntn;
        tmi = (xcd / ntn) * per + slot;
    }
    const int b = tmi / (tbx * tby), rr = tmi - b * (tbx * tby), by = rr / tbx, bx = rr - by * tbx;
    const int nchunks = p.Ci / KC;

    // per-sample descriptors: offsets stay far below 2 GiB whatever the batch
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)b * p.H * p.W * p.Ci), 0, p.H * p.W * p.Ci * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.U), 0, 16 * p.Ci * p.Co * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(STY ? p.a_scale + (size_t)b * p.a_ld : p.x), 0, STY ? p.Ci * 4 : 0, 0x00020000);

    // ---- staging role: thread = (tile t, patch column nu, channel quad q); TI = 2: two tasks per thread, quads ql and ql + 2 ----
    // LDS image of a chunk: [pos][row 64 B = 16 channels]; logical (t, 16-byte slot s) of position pos lives at row t ^ (nu & 1), slot
    // s ^ ((t >> WGS_WINO_SWZ) & 3) ^ (nu & 2), nu = pos & 3: the quad's four positions and the two channel quads of 8 neighbouring lanes fall
    // into 8 different 16-byte bank groups (ds_write_b128: runs of 8 lanes).  Fragment reads: ds_read_b128 is serviced in the lane groups
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32) (MI355X_MICROARCH.md, section LDS); with the shift 1 every group is 2-way conflicted (the
    // 26-31 % conflict cycles of profiles/r5_conv_pmc.json), with 2 rows 0-3 / 12-15 / 20-23 / 24-27 carry the terms 0 / 3 / 1 / 2: conflict-free.
    const int t = tid >> (NTASK == 2 ? 3 : 4), ql = (tid >> 2) & (NTASK == 2 ? 1 : 3), nu = tid & 3, ty = t >> 3, tx = t & 7;
    int a_off[4];
    {
        const int ix = bx * 16 + 2 * tx + nu - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iy = by * C::PH + 2 * ty + r - 1;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            a_off[r] = ok ? ((iy * p.W + ix) * p.Ci + ql * 4) * 4 : OOB;
        }
    }
    // row pass of lane nu (columns of the patch live in the quad's four lanes): own + sb * column {2, 2, 1, 1}[nu]
    //   nu 0: d0 - d2    nu 1: d1 + d2    nu 2: d2 - d1    nu 3: d3 - d1 = -(B^T row 3; the sign sits in U)
    const float sb = nu == 1 ? 1.f : -1.f;
    unsigned char* v_dst[NTASK];
#pragma unroll
    for (int u = 0; u < NTASK; ++u) v_dst[u] = smem + nu * PS + (t ^ (nu & 1)) * 64 + (((ql + 2 * u) ^ ((t >> WGS_WINO_SWZ) & 3) ^ (nu & 2)) * 16);

    f32x4 ra[NTASK][4], rsv[NTASK];
    auto load_A = [&](int c, int u) {
        const int cb = min(c, nchunks - 1) * (KC * 4) + u * 32;       // past the end: re-read the last chunk (stored into a dead buffer)
#pragma unroll
        for (int r = 0; r < 4; ++r) ra[u][r] = buf_load4(rx, a_off[r], cb);
        if (STY) rsv[u] = buf_load4(rs, ql * 16, cb);
    };
    auto row_pass = [&](int u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) quad_row_pass(ra[u][r], sb);
    };
    // column pass B^T (over the patch rows, per lane) of output row xi, times the style, stored as position (xi, nu)
    auto col_store = [&](int buf, int u, int xi) {
        const int ia = xi == 0 ? 0 : (xi == 2 ? 2 : 1), ib = xi == 3 ? 3 : (xi == 2 ? 1 : 2);
        f32x2 h[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const f32x2 a = {ra[u][ia][2 * k], ra[u][ia][2 * k + 1]}, bb = {ra[u][ib][2 * k], ra[u][ib][2 * k + 1]};
            h[k] = xi == 1 ? pk_add(a, bb) : pk_sub(a, bb);
            if (STY) h[k] = pk_mul(h[k], (f32x2){rsv[u][2 * k], rsv[u][2 * k + 1]});
        }
        const f32x4 v = {h[0][0], h[0][1], h[1][0], h[1][1]};
        *reinterpret_cast<f32x4*>(v_dst[u] + buf * STAGE + xi * 4 * PS) = v;
    };
    auto store_A = [&](int buf, int u) {
        row_pass(u);
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) col_store(buf, u, xi);
    };

    // ---- MFMA role: wave w owns the four positions (xi = w >> 1, nu = 0..3) for half of the workgroup's channel blocks (w & 1): the
    // row half of the output transform then needs no other wave's accumulators.  Its B fragments come straight from global memory. ----
    constexpr int TJH = NW == 8 ? TJ / 2 : TJ;       // 32-channel blocks per wave
    const int xi_w = NW == 8 ? wave >> 1 : wave, nh = NW == 8 ? wave & 1 : 0;
    int f_off[4][2];       // [pp = nu][g]: LDS byte offset of the A fragment (row block 0; row block i: + i * 2048)
    int u_off[4];          // [pp]: byte offset of the wave's U fragments inside a chunk (+ (g * TJ + j) * 1024)
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
        const int pos = 4 * xi_w + pp;
        const int sx = lh ^ ((l31 >> WGS_WINO_SWZ) & 3) ^ (pp & 2);
#pragma unroll
        for (int g = 0; g < 2; ++g) f_off[pp][g] = pos * PS + (l31 ^ (pp & 1)) * 64 + ((sx ^ (2 * g)) * 16);
        u_off[pp] = pos * (2 * TJ * 1024) + nh * TJH * 1024 + lane * 16;
    }
    // unit u of a chunk: k group g = u >> 2, position pp = u & 3: 4 * TI * TJH = 8 MFMAs, 8 units per chunk
    f32x4 bfr[4][TJH];     // ring over units: unit u lives in slot u % 4 and is requested three units ahead
    auto load_B = [&](int c, int u) {
        c += u >> 3; u &= 7;
        const int g = u >> 2, pp = u & 3;
        const int cb = (nb0 * nchunks + min(c, nchunks - 1)) * C::U_CHUNK + g * TJ * 1024;
#pragma unroll
        for (int jl = 0; jl < TJH; ++jl) bfr[u % 4][jl] = buf_load4(ru, u_off[pp], cb + jl * 1024);
    };
    f32x16 acc[4][TI][TJH];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJH; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][i][j][r] = 0.f;

    // Chunk kt multiplies LDS buffer kt & 1 in 8 units of 8 MFMAs.  The staging work of chunk kt + 1 (row pass, four column pass +
    // store pieces per task) and the requests of chunk kt + 2 are spread over the units, and inside a unit the scheduler is told to
    // alternate one MFMA with a few vector instructions: the waves of a workgroup run in step (one barrier per chunk), and next to
    // the fp32 MFMA every vector instruction costs 4-5 cycles of matrix-pipe time (profiles/r3_ubench_mfma_valu.txt) — a staging
    // block issued as one run idles the pipe for its whole length.
    auto mma_chunk = [&](int cur, int kt) {
        const unsigned char* base = smem + cur * STAGE;
        const int nxt = cur ^ 1;
        f32x4 af[2][TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(base + f_off[0][0] + i * 2048);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int g = u >> 2, pp = u & 3;
            if (u + 1 < 8) {
#pragma unroll
                for (int i = 0; i < TI; ++i) af[(u + 1) & 1][i] = *reinterpret_cast<const f32x4*>(base + f_off[(u + 1) & 3][(u + 1) >> 2] + i * 2048);
            }
            load_B(kt, u + 3);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int jl = 0; jl < TJH; ++jl)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[pp][i][jl] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u & 1][i][e], bfr[u % 4][jl][e], acc[pp][i][jl], 0, 0, 0);
            if (NTASK == 2) {
                if (u == 0) row_pass(0);
                if (u == 1) { col_store(nxt, 0, 0); col_store(nxt, 0, 1); }
                if (u == 2) { col_store(nxt, 0, 2); col_store(nxt, 0, 3); load_A(kt + 2, 0); }
                if (u == 3) row_pass(1);
                if (u == 4) { col_store(nxt, 1, 0); col_store(nxt, 1, 1); }
                if (u == 5) { col_store(nxt, 1, 2); col_store(nxt, 1, 3); }
                if (u == 6) load_A(kt + 2, 1);
            } else {
                if (u == 0) row_pass(0);
                if (u >= 1 && u <= 4) col_store(nxt, 0, u - 1);
                if (u == 4) load_A(kt + 2, 0);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

#pragma unroll
    for (int u = 0; u < NTASK; ++u) load_A(0, u);
    load_B(0, 0); load_B(0, 1); load_B(0, 2);
#pragma unroll
    for (int u = 0; u < NTASK; ++u) store_A(0, u);
#pragma unroll
    for (int u = 0; u < NTASK; ++u) load_A(1, u);
    __syncthreads();
    for (int kt = 0; kt < nchunks; ++kt) {
        mma_chunk(kt & 1, kt);
        __syncthreads();
    }

    // ---- epilogue: A^T . A, then demodulation, noise, bias, activation ----
    // Row half in registers (the wave holds all four nu of its xi): z0 = m0 + m1 + m2, z1 = m1 - m2 - m3 — the exchange through LDS
    // carries 8 instead of 16 blocks per (tile, channel) and fits in ONE pass.  Then thread = (channel n, tile quad tq), twice: four
    // horizontally adjacent tiles = 2 rows x 8 columns of output pixels; all operands through buffer descriptors (a missing one has
    // zero records and reads as 0), the noise values requested in one batch, the stores with a per-output uniform (scalar) offset.
    constexpr int BN = C::BN;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jl = 0; jl < TJH; ++jl) {
            const f32x16 m0 = acc[0][i][jl], m1 = acc[1][i][jl], m2 = acc[2][i][jl], m3 = acc[3][i][jl];
            acc[0][i][jl] = m0 + m1 + m2;
            acc[1][i][jl] = m1 - m2 - m3;
        }
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int jl = 0; jl < TJH; ++jl)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x16& a = acc[j2][i][jl];
                    const f32x4 v = {a[4 * rq], a[4 * rq + 1], a[4 * rq + 2], a[4 * rq + 3]};
                    *reinterpret_cast<f32x4*>(smem + ((xi_w * 2 + j2) * BN + (nh * TJH + jl) * 32 + l31) * EPI_ROW + (i * 32 + 8 * rq + 4 * lh) * 4) = v;
                }
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)b * p.H * p.W * p.Co, 0, p.H * p.W * p.Co * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.noise ? p.noise : p.x), 0, p.noise ? p.H * p.W * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.col_scale ? p.col_scale + (size_t)b * p.col_ld : p.x), 0, p.col_scale ? p.Co * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias ? p.bias : p.x), 0, p.bias ? p.Co * 4 : 0, 0x00020000);
    const float nw = p.noise ? p.noise_w[0] : 0.f;
    const float slope = p.act_slope, gain = p.gain, alpha = p.alpha;
    constexpr int NIT = BN * (C::TB / 4) / NT;       // (channel, tile quad) pairs per thread
    int n_l[NIT], oy0[NIT], ox0[NIT];
    float nz[NIT][2][8], cs_raw[NIT], bs[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NT, tq = idx / BN;
        n_l[it] = idx % BN;
        oy0[it] = by * C::PH + 2 * (tq >> 1); ox0[it] = bx * 16 + 8 * (tq & 1);
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                nz[it][i2][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rn, ((oy0[it] + i2) * p.W + ox0[it]) * 4, c * 4, 0));
        cs_raw[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rc, (nb0 * BN + n_l[it]) * 4, 0, 0));
        bs[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, (nb0 * BN + n_l[it]) * 4, 0, 0));
    }
    __syncthreads();
    float vmax = 0.f;
    float st1 = 0.f, st2 = 0.f;          // column statistics: a thread finishes ONE channel (NT % BN == 0: the same one in every trip), 16 outputs per trip
    static_assert(NT % BN == 0, "a thread's channel must not change between trips");
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int tq = (tid + it * NT) / BN;
        f32x4 z[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) z[xi][j2] = *reinterpret_cast<const f32x4*>(smem + ((xi * 2 + j2) * BN + n_l[it]) * EPI_ROW + tq * 16);
        const float cs = (p.col_scale ? cs_raw[it] : 1.f) * alpha;
        const int y_voff = ((oy0[it] * p.W + ox0[it]) * p.Co + nb0 * BN + n_l[it]) * 4;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            f32x4 yv[2];
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) yv[j2] = i2 == 0 ? z[0][j2] + z[1][j2] + z[2][j2] : z[1][j2] - z[2][j2] - z[3][j2];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) {
                    float v = __builtin_fmaf(yv[j2][k], cs, __builtin_fmaf(nw, nz[it][i2][2 * k + j2], bs[it]));
                    v = fmaxf(v, v * slope) * gain;
                    if (p.y_amax) vmax = fmaxf(vmax, fabsf(v));
                    if (p.col_stats) { st1 += v; st2 = __builtin_fmaf(v, v, st2); }
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, y_voff, ((i2 * p.W + 2 * k + j2) * p.Co) * 4, 0);
                }
        }
    }
    if (p.y_amax) {
        vmax = wave_max(vmax);
        if (lane == 0) raise_amax(p.y_amax, vmax);
    }
    if (p.col_stats) {
        double* wr = p.col_stats + (size_t)(blockIdx.x % (unsigned)wgs_bn_nrep(p.Co)) * 2 * p.Co + nb0 * BN + n_l[0];
        unsafeAtomicAdd(wr, (double)st1);
        unsafeAtomicAdd(wr + p.Co, (double)st2);
    }
}

// the launch's taps as a full 3x3 stride-1 'same' conv: w_of[(dy+1)*3 + dx+1] = weight slab; false if it is anything else
bool wino_taps(const wgs_conv_desc* d, WinoTaps& tp) {
    if (d->ntaps != 9) return false;
    for (int s = 0; s < 9; ++s) tp.w_of[s] = -1;
    for (int t = 0; t < 9; ++t) {
        const int dy = d->dy[t], dx = d->dx[t];
        if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
        int& slot = tp.w_of[(dy + 1) * 3 + dx + 1];
        if (slot >= 0 || d->wt[t] < 0) return false;
        slot = d->wt[t];
    }
    return true;
}

bool wino_ok(const wgs_conv_desc* d) {
    if (!d || !d->x || !d->w || !d->y || d->x_f16) return false;
    WinoTaps tp;
    if (!wino_taps(d, tp)) return false;
    return d->isy == 1 && d->isx == 1 && d->osy == 1 && d->osx == 1 && d->oy0 == 0 && d->ox0 == 0 && d->ups == 0 && d->Hg == d->Hi && d->Wg == d->Wi &&
           d->Ho == d->Hi && d->Wo == d->Wi && d->Hi % 16 == 0 && d->Wi % 16 == 0 && d->Ci % KC == 0 && d->Co % 64 == 0 && d->act == 0 && !d->addend &&
           d->act_slope >= 0.f && d->act_slope <= 1.f && d->B > 0 && (long)d->Hi * d->Wi * d->Ci * 4 < 0x7fffffffL && (long)d->Hi * d->Wi * d->Co * 4 < 0x7fffffffL &&
           (long)16 * d->Ci * d->Co * 4 < 0x7fffffffL && (!d->noise || d->noise_w);
}

// Workgroup shape of a launch: 1 = 32 tiles x 128 channels, 8 waves (Cout % 128 == 0); 0 = 64 tiles x 64 channels, 8 waves;
// 2 = 32 tiles x 64 channels, 4 waves, two workgroups per CU: half the transform reuse, so 6 - 10 % slower per MFMA on the large
// layers, but 4x the workgroups — taken when the 8-wave shape would leave CUs empty (256 -> 256 @16^2, B = 32: 71 -> 49 us).
// Development switches: WGS_WINO_SMALL pins shape 2, WGS_WINO_NARROW shape 0.
int wino_shape(const wgs_conv_desc* d) {
    if (wgs_flags().wino_small) return 2;
    if (wgs_flags().wino_narrow) return 0;
    const int big = d->Co % 128 == 0 ? 1 : 0;
    const long items = (long)d->B * (d->Hi / (big ? 8 : 16)) * (d->Wi / 16) * (d->Co / (big ? 128 : 64));
    return items < 200 ? 2 : big;
}
int wino_tj(const wgs_conv_desc* d) { return wino_shape(d) == 1 ? 4 : 2; }

}  // namespace

extern "C" {

int wgs_conv_wino_supported(const wgs_conv_desc* d) { return wino_ok(d) ? 1 : 0; }

// U's fragment order depends on the workgroup shape the launch will take (TJ output-channel blocks per wave), and the shape on
// the batch and map size: a cached U is valid for launches with the same layout id only (a cache keyed on the taps alone
// handed a B = 1 launch the B = 32 launch's layout).
int wgs_conv_wino_layout(const wgs_conv_desc* d) { return wino_ok(d) ? wino_tj(d) : -1; }

int wgs_conv_wino_weight(const wgs_conv_desc* d, float* U, wgs_stream_t stream) {
    WGS_CHECK_ARG(wino_ok(d) && U, "wgs_conv_wino_weight: not a 3x3 stride-1 'same' launch the Winograd kernel covers (wgs_conv_wino_supported)");
    WinoTaps tp;
    wino_taps(d, tp);
    const long n = (long)d->Ci * d->Co;
    WGS_LAUNCH(wino_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d->w, U, d->Ci, d->Co, wino_tj(d),
               (long)d->w_row_stride, (long)d->w_tap_stride, tp);
    WGS_CHECK_LAUNCH("wino_weight_kernel");
    return WGS_OK;
}

int wgs_conv_wino(const wgs_conv_desc* d, const float* U, wgs_stream_t stream) {
    WGS_CHECK_ARG(wino_ok(d) && U, "wgs_conv_wino: not a 3x3 stride-1 'same' launch the Winograd kernel covers (wgs_conv_wino_supported)");
    WinoArgs a;
    a.x = d->x; a.U = U; a.y = d->y; a.a_scale = d->a_scale; a.col_scale = d->col_scale; a.bias = d->bias; a.noise = d->noise; a.noise_w = d->noise_w;
    a.y_amax = d->y_amax; a.col_stats = d->col_stats;
    a.B = d->B; a.H = d->Hi; a.W = d->Wi; a.Ci = d->Ci; a.Co = d->Co;
    a.a_ld = d->a_ld > 0 ? d->a_ld : d->Ci; a.col_ld = d->col_ld > 0 ? d->col_ld : d->Co;
    a.alpha = d->alpha != 0.f ? d->alpha : 1.f; a.act_slope = d->act_slope; a.gain = d->gain;
    hipStream_t st = (hipStream_t)stream;
#define WGS_WINO_LAUNCH(TI, TJ, STY, NW)                                                                                    \
    {                                                                                                                       \
        auto k = wino_f32_kernel<TI, TJ, STY, NW>;                                                                          \
        const unsigned grid = (unsigned)((long)d->B * (d->Hi / Cfg<TI, TJ>::PH) * (d->Wi / 16) * (d->Co / Cfg<TI, TJ>::BN)); \
        wgs_note_kernel("wino_f32_kernel<%d, %d, %s, %d>", TI, TJ, STY ? "true" : "false", NW);                            \
        {                                                                                                                   \
            const int ntn_ = d->Co / Cfg<TI, TJ>::BN;                                                                       \
            a.u_order = (wgs_flags().wino_uord && ntn_ > 1 && 8 % ntn_ == 0 && grid % 8 == 0 && (grid / ntn_) % (8 / ntn_) == 0) ? 1 : 0; \
        }                                                                                                                   \
        const int sm = NW == 4 ? 8 * Cfg<TI, TJ>::BN * Cfg<TI, TJ>::EPI_ROW : Cfg<TI, TJ>::SMEM;                            \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, sm);                          \
        WGS_LAUNCH(k, dim3(grid), dim3(64 * NW), sm, st, a);                                                                \
    }
    const int shape = wino_shape(d);
    if (shape == 2) { if (d->a_scale) WGS_WINO_LAUNCH(1, 2, true, 4) else WGS_WINO_LAUNCH(1, 2, false, 4) }
    else if (shape == 1) { if (d->a_scale) WGS_WINO_LAUNCH(1, 4, true, 8) else WGS_WINO_LAUNCH(1, 4, false, 8) }
    else { if (d->a_scale) WGS_WINO_LAUNCH(2, 2, true, 8) else WGS_WINO_LAUNCH(2, 2, false, 8) }
#undef WGS_WINO_LAUNCH
    WGS_CHECK_LAUNCH("wino_f32_kernel");
    return WGS_OK;
}

}  // extern "C"
