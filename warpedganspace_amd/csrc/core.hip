// Error reporting + ABI version for libwgs_hip.so.
#include "wgs_common.h"
#include "../../include/wgs.h"
#include <stdarg.h>
#include <stdlib.h>
#include <atomic>

static thread_local char g_err[512] = "";

void wgs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static std::atomic<long> g_launches{0};
static std::atomic<int> g_trace{0};
static thread_local char g_kernel[160] = "";
void wgs_count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
void wgs_note_kernel(const char* fmt, ...) {
    if (!g_trace.load(std::memory_order_relaxed)) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}

static WgsFlags read_flags() {
    WgsFlags g;
    g.dma_always = getenv("WGS_DMA_ALWAYS") != nullptr;
    g.phase_patch = getenv("WGS_PHASE_PATCH") != nullptr;
    g.no_patch = getenv("WGS_NO_PATCH") != nullptr;
    g.patch_bm256 = getenv("WGS_PATCH_BM256") != nullptr;
    g.patch_tps1 = getenv("WGS_PATCH_TPS1") != nullptr;
    g.up_gh16 = getenv("WGS_UP_GH16") != nullptr;
    g.patch_dma_bm = (getenv("WGS_PATCH_DMA_BM") && atoi(getenv("WGS_PATCH_DMA_BM")) == 128) ? 128 : 256;      // ... its tile rows (conv_patch_dma.hip: 256 measured best)
    g.patch_dma_bn256 = getenv("WGS_PATCH_DMA_BN256") != nullptr;   // ... its 8-wave 256 x 256 tile for Cout % 256 == 0 (measured 2-3 % slower than the LDS-DMA kernel: off)
    g.wino16_min_wg = getenv("WGS_WINO16_MIN_WG") ? atoi(getenv("WGS_WINO16_MIN_WG")) : 200;      // conv_wino_bf16.hip: launches with fewer workgroups are left to the direct kernels' split-K forms
    g.patch_nodma = getenv("WGS_PATCH_NODMA") != nullptr;      // fp16-plane 128 x 128 tiles: the register-staged patch kernel instead of conv_patch_dma.hip
    g.up_gh8 = getenv("WGS_UP_GH8") != nullptr;      // fused up-sampling kernel: the 4-wave 14 x 6-cell tiles (two workgroups per CU) whatever the launch size
    g.patch_ntf0 = getenv("WGS_PATCH_NTF0") != nullptr;
    g.wgrad_per_tap = getenv("WGS_WGRAD_PER_TAP") != nullptr;
    g.patch_wide = getenv("WGS_PATCH_WIDE") != nullptr;
    g.f32_small = getenv("WGS_F32_SMALL") != nullptr;      // exact fp32: 4-wave 128-row tiles only (no 8-wave tiles, no merged up-conv phases)
    g.wino_small = getenv("WGS_WINO_SMALL") != nullptr;      // Winograd fp32: 4-wave workgroups of 32 tiles x 64 channels, two per CU
    g.wino_uord = !(getenv("WGS_WINO_UORD") && atoi(getenv("WGS_WINO_UORD")) == 0);      // Winograd fp32: an XCD's workgroups share one channel block of U instead of one input patch (0: the round-3 order)
    g.wino_narrow = getenv("WGS_WINO_NARROW") != nullptr;    // Winograd fp32: the 64-tile x 64-channel workgroup shape even where Cout % 128 == 0
    g.halo_min_tiles = getenv("WGS_HALO_MIN_TILES") ? atoi(getenv("WGS_HALO_MIN_TILES")) : 512;      // (tests: 1 = every covered shape)
    g.rbf_split = getenv("WGS_RBF_SPLIT") != nullptr;      // RBF forward as two launches (support vectors split over workgroups + finish)
    g.no_halo = getenv("WGS_NO_HALO") != nullptr;      // few-channel 3x3 convs on large maps through the GEMM-tiled kernels (conv_halo16.hip off)
    g.wgrad_staged = getenv("WGS_WGRAD_STAGED") != nullptr;      // weight gradients: the LDS-staged kernels everywhere (conv_wgrad_direct.hip off)
    g.check_ws = getenv("WGS_CHECK_WS") != nullptr;      // debug: verify (synchronously) that the BatchNorm / column-sum scratch is zero on entry
    g.f32_old = getenv("WGS_F32_OLD") != nullptr;      // exact fp32: the plain three-phase kernel of conv_igemm.hip everywhere
    // producer-written fp16 activation planes (x_f16): stride-1 3x3 launches with fewer output columns than this take the patch form,
    // the others the LDS-DMA kernel (development: WGS_PLANE_PATCH_MAX_CO=0 pins the LDS-DMA kernel, 100000 the patch form)
    g.plane_patch_max_co = getenv("WGS_PLANE_PATCH_MAX_CO") ? atoi(getenv("WGS_PLANE_PATCH_MAX_CO")) : 256;
    return g;
}
static WgsFlags& flags_storage() {
    static WgsFlags f = read_flags();      // first use: thread-safe one-time initialisation
    return f;
}
const WgsFlags& wgs_flags() { return flags_storage(); }

extern "C" {
const char* wgs_last_error(void) { return g_err; }
int wgs_abi_version(void) { return 10; }     // 10: wgs_sg2_blur_bwd_f16_x16 (transposed blur of an fp16 gradient plane).  9: wgs_conv_wino16* (split-bf16 F(2,3) form of the 3x3 stride-1 convs).  8: wgs_wgrad_desc.ws / ws_bytes (direct-fragment weight gradients); split-bf16 in wgs_sg2_upconv_blur_act.  7: wgs_sample_step; split-bf16 weight gradients with few input channels.  6: wgs_conv_wino_layout; fp16 activation planes
void wgs_dev_reload_flags(void) { flags_storage() = read_flags(); }
int64_t wgs_dev_launch_count(void) { return (int64_t)g_launches.load(std::memory_order_relaxed); }
void wgs_dev_trace_kernels(int on) { g_trace.store(on ? 1 : 0, std::memory_order_relaxed); g_kernel[0] = 0; }
const char* wgs_dev_last_kernel(void) { return g_kernel; }
}
