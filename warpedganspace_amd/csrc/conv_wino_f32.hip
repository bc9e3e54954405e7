// 3x3 stride-1 convolution in fp32 through the Winograd minimal-filtering form F(2x2, 3x3) on the fp32 matrix cores: 16 multiplies
// per 2x2 output block and input channel instead of 36 — the same fp32 arithmetic class the reference's convolutions get from
// cuDNN's algorithm search (lib/trainer.py:166 sets cudnn.benchmark = True; models/StyleGAN2/model.py:187-228 is F.conv2d).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          d: 4x4 input patch (stride 2), g: 3x3 filter, Y: 2x2 outputs
//
// which over channels is 16 independent GEMMs   M_pos[tile, co] = sum_ci V_pos[tile, ci] * U_pos[ci, co],   pos = (xi, nu) in 4x4.
//
// ONE kernel, nothing but x, U and y touches HBM:
//   * U = G g G^T is transformed once per (frozen) weight tensor by wino_weight_kernel, straight into the LDS image order of
//     the main kernel ([Co/64][Ci/8][pos][64 n][8 k], bank swizzle included) — a chunk of U is one contiguous 32 KB copy.
//   * a workgroup (8 waves) owns 8x8 Winograd tiles (16x16 output pixels) of one sample x 64 output channels, and ALL 16
//     positions: wave w accumulates positions 2w, 2w+1 as 64x64 blocks (128 accumulator registers per lane).
//   * per 8-channel chunk: thread (tile, channel quad, patch column) loads its 4 patch rows (16 B each, zero padding = buffer range
//     check), does the column pass B^T d in registers and the row pass across the 4 lanes of its quad with DPP quad_perm, scales by the
//     style and writes its 4 positions to LDS; the chunk's U block is copied beside it.  Loads fly for three quarters of a chunk, stores
//     and loads are issued inside the MFMA slots (conv_nt_kernel.inc's scheme), operand fragments are one ds_read_b128 per 4 MFMAs
//     (k pairing of conv_scheme.h Scheme<4>).
//   * epilogue: the 16 position blocks go through LDS (two passes of 32 channels, 136 KB), every thread applies A^T . A to four tiles
//     of one channel, then demodulation / noise / bias / leaky-relu as conv_epilogue.h and stores 128-byte channel runs.
// LDS rows are 32 B (8 channels); conflict-free by swizzle: 16-byte half h of logical row r of position pos lives at
// row r ^ (pos & 3), half h ^ ((r >> 2) & 1).
#include "wgs_common.h"
#include "../../include/wgs.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TB = 64;            // Winograd tiles per workgroup: 8 x 8  (16 x 16 output pixels)
constexpr int BN = 64;            // output channels per workgroup
constexpr int KC = 8;             // input channels per chunk
constexpr int NT = 512;
constexpr int V_BYTES = 16 * TB * KC * 4;      // 32 KB
constexpr int U_BYTES = 16 * BN * KC * 4;      // 32 KB
constexpr int STAGE = V_BYTES + U_BYTES;
constexpr int EPI_ROW = TB * 4 + 16;           // one (pos, n) row of the epilogue exchange: 64 tiles + 16 B (bank spread)
constexpr int EPI_BYTES = 16 * 32 * EPI_ROW;   // 136 KB
constexpr int SMEM = EPI_BYTES > 2 * STAGE ? EPI_BYTES : 2 * STAGE;
constexpr int OOB = (int)0x80000000;

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
    int B, H, W, Ci, Co, a_ld, col_ld;
    float alpha, act_slope, gain;
};

struct WinoTaps { int w_of[9]; };      // weight slab index of spatial tap (ky, kx), -1 = absent

__device__ __forceinline__ f32x4 buf_load4(const __amdgpu_buffer_rsrc_t r, int voff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}
// value of lane {2, 2, 1, 1}[lane & 3] of the same quad
__device__ __forceinline__ float quad_other(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x5A, 0xf, 0xf, true));
}

// U[pos] = G g G^T of every (co, ci) pair, written in the main kernel's LDS image order
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Ci, int Co,
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
        u[r][3] = t[r][2];
    }
    const int nchunks = Ci / KC;
    const int nb = co / BN, n = co % BN, chunk = ci / KC, k = ci % KC;
    float* base = U + ((size_t)nb * nchunks + chunk) * (16 * BN * KC);
#pragma unroll
    for (int pos = 0; pos < 16; ++pos)
        base[((pos * BN + (n ^ (pos & 3))) * 2 + ((k >> 2) ^ ((n >> 2) & 1))) * 4 + (k & 3)] = u[pos >> 2][pos & 3];
}

template <bool STY>
__global__ __launch_bounds__(NT, 1) void wino_f32_kernel(const WinoArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int ntn = p.Co / BN, tbx = p.W >> 4, tby = p.H >> 4;
    // XCD-aware order (hardware sends workgroup i to XCD i % 8): every XCD gets a contiguous range of the (pixel block major,
    // channel block minor) list, so the channel blocks of a pixel block share their input patch in that XCD's L2
    int bid;
    {
        const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, qn = nb >> 3, rn = nb & 7;
        bid = xcd * qn + min(xcd, rn) + slot;
    }
    const int tmi = bid / ntn, nb0 = bid - tmi * ntn;
    const int b = tmi / (tbx * tby), rr = tmi - b * (tbx * tby), by = rr / tbx, bx = rr - by * tbx;
    const int nchunks = p.Ci / KC;

    // per-sample descriptors: offsets stay far below 2 GiB whatever the batch
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)b * p.H * p.W * p.Ci), 0, p.H * p.W * p.Ci * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.U), 0, 16 * p.Ci * p.Co * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(STY ? p.a_scale + (size_t)b * p.a_ld : p.x), 0, STY ? p.Ci * 4 : 0, 0x00020000);

    // ---- staging role: thread = (tile t, channel quad q, patch column nu) ----
    const int t = tid >> 3, q = (tid >> 2) & 1, nu = tid & 3, ty = t >> 3, tx = t & 7;
    int a_off[4];
    {
        const int ix = bx * 16 + 2 * tx + nu - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iy = by * 16 + 2 * ty + r - 1;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            a_off[r] = ok ? ((iy * p.W + ix) * p.Ci + q * 4) * 4 : OOB;
        }
    }
    const int u_off = nb0 * nchunks * U_BYTES + tid * 16;
    const int v_st = nu * 2048 + (t ^ nu) * 32 + (q ^ ((t >> 2) & 1)) * 16;       // + xi * 8192
    const float sa = nu == 3 ? -1.f : 1.f, sb = (nu & 1) ? 1.f : -1.f;

    f32x4 ra[4], rw[4], rsv = {1.f, 1.f, 1.f, 1.f};
    auto load_U = [&](int c) {
        const int cb = u_off + min(c, nchunks - 1) * U_BYTES;      // past the end: re-read the last chunk (stored into a dead buffer)
#pragma unroll
        for (int i = 0; i < 4; ++i) rw[i] = buf_load4(ru, cb + i * 8192);
        if (STY) rsv = buf_load4(rs, min(c, nchunks - 1) * 32 + q * 16);
    };
    auto load_A = [&](int c) {
        const int cb = min(c, nchunks - 1) * 32;
#pragma unroll
        for (int r = 0; r < 4; ++r) ra[r] = buf_load4(rx, a_off[r] < 0 ? OOB : a_off[r] + cb);
    };
    auto store_U = [&](int buf) {
        unsigned char* base = smem + buf * STAGE + V_BYTES + tid * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(base + i * 8192) = rw[i];
    };
    auto store_A = [&](int buf) {
        // column pass B^T d (rows of the patch), then the row pass over the quad's four columns
        f32x4 T[4];
        T[0] = ra[0] - ra[2];
        T[1] = ra[1] + ra[2];
        T[2] = ra[2] - ra[1];
        T[3] = ra[1] - ra[3];
        unsigned char* base = smem + buf * STAGE + v_st;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            f32x4 o, v;
            o[0] = quad_other(T[xi][0]); o[1] = quad_other(T[xi][1]); o[2] = quad_other(T[xi][2]); o[3] = quad_other(T[xi][3]);
            v = sa * T[xi] + sb * o;
            if (STY) v *= rsv;
            *reinterpret_cast<f32x4*>(base + xi * 8192) = v;
        }
    };

    // ---- MFMA role: wave w owns positions 2w, 2w + 1 ----
    int f_off[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int pos = 2 * wave + pp;
        f_off[pp] = pos * 2048 + (l31 ^ (pos & 3)) * 32 + (lh ^ ((l31 >> 2) & 1)) * 16;
    }
    f32x16 acc[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][i][j][r] = 0.f;

    // chunk kt multiplies LDS buffer kt & 1.  Slot s = (position pp, row block i): 8 MFMAs.  Slot 0 stores the U block of chunk
    // kt + 1 (requested during chunk kt - 1), slot 1 transforms + stores its patch and requests U of chunk kt + 2, slot 2 requests
    // the patch of chunk kt + 2: every load flies for three slots.
    auto mma_chunk = [&](int cur, int kt) {
        const unsigned char* base = smem + cur * STAGE;
        f32x4 af[2], bf[2][2];
        af[0] = *reinterpret_cast<const f32x4*>(base + f_off[0]);
        bf[0][0] = *reinterpret_cast<const f32x4*>(base + V_BYTES + f_off[0]);
        bf[0][1] = *reinterpret_cast<const f32x4*>(base + V_BYTES + f_off[0] + 1024);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pp = s >> 1, i = s & 1;
            if (s + 1 < 4) {
                const int pn = (s + 1) >> 1, in = (s + 1) & 1;
                af[(s + 1) & 1] = *reinterpret_cast<const f32x4*>(base + f_off[pn] + in * 1024);
                if (in == 0) {
                    bf[pn][0] = *reinterpret_cast<const f32x4*>(base + V_BYTES + f_off[pn]);
                    bf[pn][1] = *reinterpret_cast<const f32x4*>(base + V_BYTES + f_off[pn] + 1024);
                }
            }
            if (s == 2) load_A(kt + 2);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[pp][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][e], bf[pp][j][e], acc[pp][i][j], 0, 0, 0);
            if (s == 0) store_U(cur ^ 1);
            if (s == 1) { store_A(cur ^ 1); load_U(kt + 2); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    load_U(0);
    load_A(0);
    store_U(0);
    store_A(0);
    load_U(1);
    load_A(1);
    __syncthreads();
    for (int kt = 0; kt < nchunks; ++kt) {
        mma_chunk(kt & 1, kt);
        __syncthreads();
    }

    // ---- epilogue: positions -> LDS -> A^T . A per (tile, channel) -> demodulation, noise, bias, activation ----
    const float alpha = p.alpha;
    const float nw = p.noise ? p.noise_w[0] : 0.f;
    const int n_l = tid & 31, tq = tid >> 5;
    const int oy0 = by * 16 + 2 * (tq >> 1), ox0 = bx * 16 + 8 * (tq & 1);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)b * p.H * p.W * p.Co, 0, p.H * p.W * p.Co * 4, 0x00020000);
    const float slope = p.act_slope, gain = p.gain;
    float vmax = 0.f;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        if (jj) __syncthreads();
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 v = {acc[pp][i][jj][4 * rq], acc[pp][i][jj][4 * rq + 1], acc[pp][i][jj][4 * rq + 2], acc[pp][i][jj][4 * rq + 3]};
                    *reinterpret_cast<f32x4*>(smem + ((2 * wave + pp) * 32 + l31) * EPI_ROW + (i * 32 + 8 * rq + 4 * lh) * 4) = v;
                }
        __syncthreads();
        f32x4 m[4][4];
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) m[pos >> 2][pos & 3] = *reinterpret_cast<const f32x4*>(smem + (pos * 32 + n_l) * EPI_ROW + tq * 16);
        f32x4 z[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            z[xi][0] = m[xi][0] + m[xi][1] + m[xi][2];
            z[xi][1] = m[xi][1] - m[xi][2] - m[xi][3];
        }
        const int n = nb0 * BN + jj * 32 + n_l;
        const float cs = (p.col_scale ? p.col_scale[(size_t)b * p.col_ld + n] : 1.f) * alpha;
        const float bs = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            f32x4 yv[2];
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) yv[j2] = i2 == 0 ? z[0][j2] + z[1][j2] + z[2][j2] : z[1][j2] - z[2][j2] - z[3][j2];
            const int oy = oy0 + i2;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) {
                    const int ox = ox0 + 2 * k + j2;
                    float v = yv[j2][k] * cs;
                    if (p.noise) v += nw * p.noise[oy * p.W + ox];
                    v += bs;
                    v = fmaxf(v, v * slope) * gain;
                    vmax = fmaxf(vmax, fabsf(v));
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, ((oy * p.W + ox) * p.Co + n) * 4, 0, 0);
                }
        }
    }
    if (p.y_amax) {
        vmax = wave_max(vmax);
        if (lane == 0) raise_amax(p.y_amax, vmax);
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
           d->Ho == d->Hi && d->Wo == d->Wi && d->Hi % 16 == 0 && d->Wi % 16 == 0 && d->Ci % KC == 0 && d->Co % BN == 0 && d->act == 0 && !d->addend &&
           d->act_slope >= 0.f && d->act_slope <= 1.f && d->B > 0 && (long)d->Hi * d->Wi * d->Ci * 4 < 0x7fffffffL && (long)d->Hi * d->Wi * d->Co * 4 < 0x7fffffffL &&
           (long)16 * d->Ci * d->Co * 4 < 0x7fffffffL && (!d->noise || d->noise_w);
}

}  // namespace

extern "C" {

int wgs_conv_wino_supported(const wgs_conv_desc* d) { return wino_ok(d) ? 1 : 0; }

int wgs_conv_wino_weight(const wgs_conv_desc* d, float* U, wgs_stream_t stream) {
    WGS_CHECK_ARG(wino_ok(d) && U, "wgs_conv_wino_weight: not a 3x3 stride-1 'same' launch the Winograd kernel covers (wgs_conv_wino_supported)");
    WinoTaps tp;
    wino_taps(d, tp);
    const long n = (long)d->Ci * d->Co;
    WGS_LAUNCH(wino_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d->w, U, d->Ci, d->Co, (long)d->w_row_stride,
               (long)d->w_tap_stride, tp);
    WGS_CHECK_LAUNCH("wino_weight_kernel");
    return WGS_OK;
}

int wgs_conv_wino(const wgs_conv_desc* d, const float* U, wgs_stream_t stream) {
    WGS_CHECK_ARG(wino_ok(d) && U, "wgs_conv_wino: not a 3x3 stride-1 'same' launch the Winograd kernel covers (wgs_conv_wino_supported)");
    WinoArgs a;
    a.x = d->x; a.U = U; a.y = d->y; a.a_scale = d->a_scale; a.col_scale = d->col_scale; a.bias = d->bias; a.noise = d->noise; a.noise_w = d->noise_w;
    a.y_amax = d->y_amax;
    a.B = d->B; a.H = d->Hi; a.W = d->Wi; a.Ci = d->Ci; a.Co = d->Co;
    a.a_ld = d->a_ld > 0 ? d->a_ld : d->Ci; a.col_ld = d->col_ld > 0 ? d->col_ld : d->Co;
    a.alpha = d->alpha != 0.f ? d->alpha : 1.f; a.act_slope = d->act_slope; a.gain = d->gain;
    const unsigned grid = (unsigned)((long)d->B * (d->Hi / 16) * (d->Wi / 16) * (d->Co / BN));
    hipStream_t st = (hipStream_t)stream;
    if (d->a_scale) {
        auto k = wino_f32_kernel<true>;
        wgs_note_kernel("wino_f32_kernel<true>");
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        WGS_LAUNCH(k, dim3(grid), dim3(NT), SMEM, st, a);
    } else {
        auto k = wino_f32_kernel<false>;
        wgs_note_kernel("wino_f32_kernel<false>");
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        WGS_LAUNCH(k, dim3(grid), dim3(NT), SMEM, st, a);
    }
    WGS_CHECK_LAUNCH("wino_f32_kernel");
    return WGS_OK;
}

}  // extern "C"
