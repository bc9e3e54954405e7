// BigGAN generator glue (models/BigGAN/layers.py): class-conditional BatchNorm in eval mode = per-sample,
// per-channel affine (ccbn.forward :303-322 with F.batch_norm on the stored statistics) fused with the ReLU
// that always follows it in GBlock.forward (:393-405), and the row softmax of the self-attention block
// (Attention.forward :153-166).  HBM-bound streaming kernels on NHWC tensors.
#include "wgs_common.h"
#include "../../include/wgs.h"

namespace {

// y[b,p,c] = relu?(x[b,p,c]*scale[b,c] + shift[b,c])
__global__ __launch_bounds__(256) void affine_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float* __restrict__ y, int B,
                                                              int P, int C, int relu) {
    const int c4n = C >> 2;
    const int64_t total = (int64_t)B * P * c4n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % c4n) * 4;
        const int b = (int)(e / ((int64_t)P * c4n));
        const float4 v = reinterpret_cast<const float4*>(x)[e];
        const float4 s = *reinterpret_cast<const float4*>(scale + (size_t)b * C + c);
        const float4 t = *reinterpret_cast<const float4*>(shift + (size_t)b * C + c);
        float4 o = make_float4(fmaf(v.x, s.x, t.x), fmaf(v.y, s.y, t.y), fmaf(v.z, s.z, t.z), fmaf(v.w, s.w, t.w));
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        reinterpret_cast<float4*>(y)[e] = o;
    }
}

// g' = g*(y > 0);  dx = g'*scale[b,c];  dscale[b,c] += sum_p g'*x;  dshift[b,c] += sum_p g'
// grid = (pixel chunks, B); same thread layout as sg2_act_bwd_kernel (fp64 partials, fp32 atomics per chunk).
__global__ __launch_bounds__(256) void affine_relu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ g, const float* __restrict__ scale,
                                                              float* __restrict__ dx, float* __restrict__ dscale,
                                                              float* __restrict__ dshift, int P, int C, int chunk, int relu) {
    __shared__ double red[2][256][4];
    const int b = blockIdx.y;
    const int c4n = C >> 2;
    const int tpp = c4n < 256 ? c4n : 256;
    const int ppi = 256 / tpp;
    const int cl = threadIdx.x % tpp, sub = threadIdx.x / tpp;
    const int p_begin = blockIdx.x * chunk, p_end = min(P, p_begin + chunk);
    for (int c = cl * 4; c < C; c += tpp * 4) {
        const float4 s = *reinterpret_cast<const float4*>(scale + (size_t)b * C + c);
        double a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
        if (sub < ppi) {
            for (int p = p_begin + sub; p < p_end; p += ppi) {
                const size_t off = ((size_t)b * P + p) * C + c;
                float4 gv = *reinterpret_cast<const float4*>(g + off);
                const float4 xv = *reinterpret_cast<const float4*>(x + off);
                if (relu) {
                    const float4 yv = *reinterpret_cast<const float4*>(y + off);
                    gv.x = yv.x > 0.f ? gv.x : 0.f; gv.y = yv.y > 0.f ? gv.y : 0.f;
                    gv.z = yv.z > 0.f ? gv.z : 0.f; gv.w = yv.w > 0.f ? gv.w : 0.f;
                }
                *reinterpret_cast<float4*>(dx + off) = make_float4(gv.x * s.x, gv.y * s.y, gv.z * s.z, gv.w * s.w);
                a1[0] += (double)(gv.x * xv.x); a1[1] += (double)(gv.y * xv.y); a1[2] += (double)(gv.z * xv.z); a1[3] += (double)(gv.w * xv.w);
                a2[0] += gv.x; a2[1] += gv.y; a2[2] += gv.z; a2[3] += gv.w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { red[0][threadIdx.x][q] = a1[q]; red[1][threadIdx.x][q] = a2[q]; }
        __syncthreads();
        if (sub == 0) {
            for (int s2 = 1; s2 < ppi; ++s2)
#pragma unroll
                for (int q = 0; q < 4; ++q) { a1[q] += red[0][s2 * tpp + cl][q]; a2[q] += red[1][s2 * tpp + cl][q]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsafeAtomicAdd(dscale + (size_t)b * C + c + q, (float)a1[q]);
                unsafeAtomicAdd(dshift + (size_t)b * C + c + q, (float)a2[q]);
            }
        }
    }
}

// row softmax over the last axis (rows of length n <= 4096 handled by one block of 256 threads)
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int n) {
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    const float* xr = x + r * n;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < n; j += 256) m = fmaxf(m, xr[j]);
    m = wave_max(m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) s += expf(xr[j] - m);
    s = block_sum<4>(s, red);
    const float inv = 1.f / s;
    for (int j = threadIdx.x; j < n; j += 256) y[r * n + j] = expf(xr[j] - m) * inv;
}
// dx = y * (dy - sum_j dy_j*y_j)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int64_t rows, int n) {
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    float s = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) s = fmaf(dy[r * n + j], y[r * n + j], s);
    s = block_sum<4>(s, red);
    for (int j = threadIdx.x; j < n; j += 256) dx[r * n + j] = y[r * n + j] * (dy[r * n + j] - s);
}

}  // namespace

extern "C" {

int wgs_affine_relu_fwd(const float* x, const float* scale, const float* shift, float* y, int B, int P, int C, int relu,
                        wgs_stream_t stream) {
    WGS_CHECK_ARG(x && scale && shift && y && B > 0 && P > 0 && C >= 4 && C % 4 == 0, "wgs_affine_relu_fwd: bad arguments (C %% 4)");
    int grid = wgs_cdiv((int64_t)B * P * (C / 4), 256);
    if (grid > 8192) grid = 8192;
    WGS_LAUNCH(affine_relu_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y, B, P, C, relu);
    WGS_CHECK_LAUNCH("affine_relu_fwd_kernel");
    return WGS_OK;
}

int wgs_affine_relu_bwd(const float* x, const float* y, const float* g, const float* scale, float* dx, float* dscale,
                        float* dshift, int B, int P, int C, int relu, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && g && scale && dx && dscale && dshift && (y || !relu), "wgs_affine_relu_bwd: null pointer");
    WGS_CHECK_ARG(B > 0 && P > 0 && C >= 4 && C % 4 == 0, "wgs_affine_relu_bwd: bad sizes (C %% 4)");
    int chunks = wgs_cdiv(2048, B);
    int chunk = wgs_cdiv(P, chunks);
    if (chunk < 16) chunk = 16;
    chunks = wgs_cdiv(P, chunk);
    WGS_LAUNCH(affine_relu_bwd_kernel, dim3(chunks, B), dim3(256), 0, (hipStream_t)stream, x, y, g, scale, dx, dscale,
                       dshift, P, C, chunk, relu);
    WGS_CHECK_LAUNCH("affine_relu_bwd_kernel");
    return WGS_OK;
}

int wgs_softmax_rows_fwd(const float* x, float* y, int64_t rows, int n, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && y && rows > 0 && rows < (1LL << 31) && n > 0, "wgs_softmax_rows_fwd: bad arguments");
    WGS_LAUNCH(softmax_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, y, rows, n);
    WGS_CHECK_LAUNCH("softmax_fwd_kernel");
    return WGS_OK;
}
int wgs_softmax_rows_bwd(const float* y, const float* dy, float* dx, int64_t rows, int n, wgs_stream_t stream) {
    WGS_CHECK_ARG(y && dy && dx && rows > 0 && rows < (1LL << 31) && n > 0, "wgs_softmax_rows_bwd: bad arguments");
    WGS_LAUNCH(softmax_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, y, dy, dx, rows, n);
    WGS_CHECK_LAUNCH("softmax_bwd_kernel");
    return WGS_OK;
}

}  // extern "C"
