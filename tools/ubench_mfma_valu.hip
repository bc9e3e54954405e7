// Micro-benchmark: how much matrix-pipe time does a vector (VALU) instruction cost when it is issued between the MFMAs of the same wave?
// 8 waves per CU (2 per SIMD, as the conv kernels run), every wave: loop of { 1 MFMA, V independent v_fma_f32 } with 8 independent
// accumulators; V = 0..12.  Two MFMA types: f32 32x32x2 (64 cycles) and f16 32x32x16 (32 cycles).  Prints TFLOP/s and the implied
// MFMA-busy fraction next to the two candidate models: full overlap (1.0 until the issue port saturates) and T / (T + 4 V).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_valu.hip -o /tmp/ubench_mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int V, bool F16>
__global__ __launch_bounds__(512, 1) void loop_kernel(float* out, int iters, float seed) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed * (threadIdx.x % 7), b = seed * (threadIdx.x % 5);
    f16x8 ah, bh;
    for (int k = 0; k < 8; ++k) { ah[k] = (_Float16)(a + k); bh[k] = (_Float16)(b - k); }
    float v[12];
    for (int k = 0; k < 12; ++k) v[k] = seed + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (F16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < V; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[k]) : "v"(seed));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int k = 0; k < 12; ++k) s += v[k];
    if (s == 1234.5f) out[0] = s;
}

template <int V, bool F16>
void run(float* out) {
    const int blocks = 256 * 4, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    loop_kernel<V, F16><<<blocks, 512>>>(out, 50, 0.5f);
    hipEventRecord(e0);
    loop_kernel<V, F16><<<blocks, 512>>>(out, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 8 * iters * 8 * 32 * 32 * (F16 ? 16 : 2) * 2;
    const double tf = flops / ms / 1e9, peak = F16 ? 2500.0 : 157.3, T = F16 ? 32.0 : 64.0;
    printf("%s  V = %2d vector instr per MFMA: %8.1f TFLOP/s = %.3f of peak   | model T/(T+4V) = %.3f\n", F16 ? "f16 32x32x16" : "f32 32x32x2 ", V, tf, tf / peak,
           T / (T + 4.0 * V));
}

int main() {
    float* out; hipMalloc(&out, 64);
    run<0, false>(out); run<1, false>(out); run<2, false>(out); run<3, false>(out); run<4, false>(out); run<6, false>(out); run<8, false>(out); run<12, false>(out);
    run<0, true>(out); run<1, true>(out); run<2, true>(out); run<3, true>(out); run<4, true>(out); run<6, true>(out); run<8, true>(out); run<12, true>(out);
    return 0;
}
