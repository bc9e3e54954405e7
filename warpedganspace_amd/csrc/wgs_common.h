// Common device/host helpers for libwgs_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define WGS_OK 0
#define WGS_EINVAL (-22)
#define WGS_ELAUNCH (-5)

// Thread-local error text (read back through wgs_last_error()).
void wgs_set_error(const char* fmt, ...);

#define WGS_CHECK_ARG(cond, ...)                    \
    do {                                            \
        if (!(cond)) {                              \
            wgs_set_error(__VA_ARGS__);             \
            return WGS_EINVAL;                      \
        }                                           \
    } while (0)

#define WGS_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            wgs_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return WGS_ELAUNCH;                                                  \
        }                                                                        \
    } while (0)

// Development A/B switches (WGS_DMA_ALWAYS, WGS_PHASE_PATCH, WGS_NO_PATCH, WGS_PATCH_BM256, WGS_PATCH_TPS1, WGS_UP_GH16,
// WGS_PATCH_NTF0, WGS_WGRAD_PER_TAP, WGS_PATCH_WIDE, WGS_F32_SMALL, WGS_F32_OLD, WGS_WINO_NARROW, WGS_WINO_SMALL, WGS_PATCH_NODMA,
// WGS_PATCH_DMA_BM ...: the full list is wgs_flags()'s initialiser in core.hip): read from the
// environment ONCE, when the first launch asks for them, and immutable afterwards — no getenv on launch paths, no mutable
// global state.  All default to off = the measured-best path.
struct WgsFlags { bool dma_always, phase_patch, no_patch, patch_bm256, patch_tps1, up_gh16, patch_ntf0, wgrad_per_tap, patch_wide, f32_small, f32_old, wino_narrow, wino_small;
                  bool no_halo, rbf_split, check_ws, wgrad_staged, up_gh8, patch_nodma, patch_dma_bn256; int plane_patch_max_co, halo_min_tiles, patch_dma_bm, wino16_min_wg;
                  bool wino_uord; };
const WgsFlags& wgs_flags();

// Launch accounting for bench.py (statistics only; nothing reads them on a launch path):
//   wgs_count_launch()  every kernel launch of the library bumps one relaxed atomic (wgs_dev_launch_count()).
//   wgs_note_kernel()   with wgs_dev_trace_kernels(1): the symbol of the last implicit-GEMM kernel launched on this thread,
//                       spelled as rocprofv3 prints it, for the per-symbol roofline (wgs_dev_last_kernel()).
void wgs_count_launch();
void wgs_note_kernel(const char* fmt, ...);
#define WGS_LAUNCH(...) do { wgs_count_launch(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

// Replicas of the BatchNorm / column-sum scratch (double [nrep][2 * C], include/wgs.h WGS_BN_WS_DOUBLES): workgroup b adds into replica
// b % nrep so that hundreds of workgroups do not serialise on the same 2 * C fp64 addresses.  32 replicas up to 64 channels, fewer for wide
// layers (whose reductions have few workgroups): every launch that SUMS the replicas then reads at most 32 KB — cheap enough for every
// workgroup of the BatchNorm apply kernels to do it in its prologue (bn_apply_fused_kernel: no finalise / collapse launch in between).
__host__ __device__ __forceinline__ int wgs_bn_nrep(int C) { const int r = 2048 / (C > 0 ? C : 1); return r < 1 ? 1 : (r > 32 ? 32 : r); }

static inline int wgs_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// n / d for a launch-uniform divisor d >= 2 as one multiply-high: magic = ceil(2^32 / d), exact for n * d < 2^32
// (a 32-bit integer division is ~35 VALU instructions on gfx950; the short-K conv tiles do a dozen of them per lane).
static inline unsigned wgs_div_magic(int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }
__device__ __forceinline__ int wgs_div_fast(int n, unsigned magic) { return (int)__umulhi((unsigned)n, magic); }

// ---- wave64 reductions (gfx950: wavefront = 64 lanes) -------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// Sum EIGHT values over the 64 lanes with 10 shuffles instead of 8 x 6: at the xor-32 / 16 / 8 steps a lane keeps half of its
// values and sends the other half, so the data halves while the partial sums double; the last three steps finish one value per
// lane.  On return lane l holds the total of v[idx], idx = 4*bit5(l) + 2*bit4(l) + bit3(l) (lane 8*j holds v[j]); the additions
// form the same tree for every idx, so a value's total does not depend on which other values were reduced with it.
__device__ __forceinline__ float wave_sum8(const float (&v)[8], int lane) {
    float t[4], u[2];
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = (b5 ? v[4 + i] : v[i]) + __shfl_xor(b5 ? v[i] : v[4 + i], 32, 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) u[i] = (b4 ? t[2 + i] : t[i]) + __shfl_xor(b4 ? t[i] : t[2 + i], 16, 64);
    float s = (b3 ? u[1] : u[0]) + __shfl_xor(b3 ? u[0] : u[1], 8, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 1, 64);
    return s;
}
// Raise a device-resident non-negative fp32 maximum (bit patterns of non-negative floats order like unsigned ints).
// Thousands of waves target ONE address: an L2 atomic costs ~12 ns when they queue (16 k of them = 0.2 ms per launch), so
// a wave first looks at the current value — after the first few arrivals almost every wave sees a value >= its own and
// skips the atomic (a stale L1 line only costs a redundant atomic, never a wrong result).
__device__ __forceinline__ void raise_amax(float* addr, float v) {
    if (v > 0.f && v > __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Block-wide sum for blockDim.x = 64*NW threads; `red` is LDS scratch of >= NW floats.
// Every thread gets the total. Contains two __syncthreads().
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}
