// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate and the clock it runs at (s_memtime cycles vs wall time).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512, 1) void mfma_loop(float* out, unsigned long long* cyc, int iters, int zero) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(zero ? 0.f : (float)((threadIdx.x * 7 + k) % 13) * 0.01f); b[k] = (__bf16)(zero ? 0.f : (float)((threadIdx.x * 3 + k) % 11) * 0.02f); }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 1234.5f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 64); hipMalloc(&cyc, 64);
    for (int zero = 0; zero < 2; ++zero)
    for (int blocks : {256, 1024}) {
        const int iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        mfma_loop<<<blocks, 512>>>(out, cyc, 100, zero);
        hipEventRecord(e0);
        mfma_loop<<<blocks, 512>>>(out, cyc, iters, zero);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        double flops = (double)blocks * 8 * iters * 16 * 32 * 32 * 16 * 2;
        printf("zero=%d blocks %4d: %.3f ms  %.1f TFLOP/s   wave cycles %llu -> %.1f cyc per MFMA per wave, s_memtime rate %.3f GHz-equivalent (block0)\n",
               zero, blocks, ms, flops / ms / 1e9, c, (double)c / (iters * 16.0), (double)c / (ms * 1e6) * (blocks / 256.0 > 1 ? 256.0 / blocks : 1.0));
    }
    return 0;
}
