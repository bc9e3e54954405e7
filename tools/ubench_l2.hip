// Micro-benchmark: what global->register bandwidth does one CU get for the implicit-GEMM A/B access pattern
// (256 threads, 8 lanes x 16 B per row, rows `stride` bytes apart), as a function of stride, footprint and
// loads in flight.  Development tool, not part of the product library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int NLD>
__global__ __launch_bounds__(256, 2) void reader(const float* __restrict__ base, size_t stride_f, int nrows, int nchunks,
                                                 int iters, int rot, float* out) {
    const int tid = threadIdx.x, q = tid & 7, r0 = tid >> 3;
    const int bid = blockIdx.x;
    float4 acc = make_float4(0, 0, 0, 0);
    int row0 = (int)(((size_t)bid * 128) % nrows);
    int chunk = rot ? (bid >> 3) % nchunks : 0;
    for (int it = 0; it < iters; ++it) {
        float4 v[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int row = row0 + r0 + (i & 3) * 32;
            if (row >= nrows) row -= nrows;
            int ch = chunk + (i >> 2);
            if (ch >= nchunks) ch -= nchunks;
            v[i] = *reinterpret_cast<const float4*>(base + (size_t)row * stride_f + ch * 32 + q * 4);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
        chunk += NLD / 4;
        if (chunk >= nchunks) { chunk -= nchunks; row0 += 128; if (row0 >= nrows) row0 -= nrows; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int NLD>
void run(const char* name, float* d, float* out, size_t stride_f, int nrows, int nchunks, int blocks, int rot) {
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    reader<NLD><<<blocks, 256>>>(d, stride_f, nrows, nchunks, 50, rot, out);
    hipEventRecord(a);
    reader<NLD><<<blocks, 256>>>(d, stride_f, nrows, nchunks, iters, rot, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double bytes = (double)blocks * iters * NLD * 256 * 16;
    printf("%-28s stride %6zu B rows %7d chunks %3d blocks %5d nld %2d rot %d : %7.3f ms  %7.2f TB/s  %6.1f B/clk/CU(2.4GHz)\n",
           name, stride_f * 4, nrows, nchunks, blocks, NLD, rot, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
    size_t cap = (size_t)1 << 30;   // 1 GiB
    float* d; float* out;
    hipMalloc(&d, cap); hipMalloc(&out, 64);
    hipMemset(d, 0, cap);
    // footprint = nrows * stride.  Small (L2-resident per XCD), medium (MALL), large (HBM).
    for (int blocks : {512, 1024}) {
        run<8>("L2-small s2048", d, out, 512, 1024, 16, blocks, 0);      // 2 MB
        run<8>("L2-small s2048 rot", d, out, 512, 1024, 16, blocks, 1);
        run<8>("L2-small s2176(pad)", d, out, 544, 1024, 16, blocks, 0);
        run<8>("L2-small s512", d, out, 128, 4096, 4, blocks, 0);
        run<8>("L2-small s128(contig)", d, out, 32, 16384, 1, blocks, 0);
        run<8>("MALL 64MB s2048", d, out, 512, 32768, 16, blocks, 0);
        run<8>("MALL 64MB s2048 rot", d, out, 512, 32768, 16, blocks, 1);
        run<8>("HBM 1GB s2048", d, out, 512, 524288, 16, blocks, 0);
        run<8>("HBM 1GB s2048 rot", d, out, 512, 524288, 16, blocks, 1);
        run<16>("L2-small s2048 nld16", d, out, 512, 1024, 16, blocks, 0);
        run<16>("L2-small s2048 nld16 rot", d, out, 512, 1024, 16, blocks, 1);
        run<4>("L2-small s2048 nld4 rot", d, out, 512, 1024, 16, blocks, 1);
    }
    return 0;
}
