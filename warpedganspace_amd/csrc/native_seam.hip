// Drop-in kernels for the reference's two native ops (the pybind seam of models/StyleGAN2/op):
//   fused_bias_act  (fused_bias_act_kernel.cu:18-49)  -> wgs_bias_act
//   upfirdn2d       (upfirdn2d_kernel.cu:52-137; semantics = upfirdn2d_native, upfirdn2d.py:152-186)
//                                                      -> wgs_upfirdn2d
// Both are pure HBM-streaming ops: 16-B per-lane accesses where the layout allows, grid capped at
// 2048 workgroups with a grid-stride loop.  The training hot path uses fused variants of these
// (stylegan2_ops.hip); the entry points here keep the reference's generic operator contract.
#include "wgs_common.h"
#include "fir4.h"
#include "../../include/wgs.h"

namespace {

__device__ __forceinline__ float bias_act_one(float x, float ref, int mode, float alpha, float scale) {
    float y;
    switch (mode) {
        default:
        case 10: case 11: y = x; break;
        case 12: case 32: y = 0.f; break;
        case 30: y = (x > 0.f) ? x : x * alpha; break;
        case 31: y = (ref > 0.f) ? x : x * alpha; break;
        case 90: y = tanhf(x); break;                     // extension: tanh (SNGAN / BigGAN output)
        case 91: y = x * (1.f - ref * ref); break;        // its backward, ref = saved tanh output
    }
    return y * scale;
}

__global__ __launch_bounds__(256) void bias_act_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ ref,
                                                       float* __restrict__ y, int mode, float alpha,
                                                       float scale, int64_t n, int step_b, int size_b) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n >> 2;
    // vector body: a float4 never straddles a bias boundary when step_b % 4 == 0 (or bias absent)
    const bool vec_ok = (!bias) || (step_b % 4 == 0);
    if (vec_ok) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            float4 v = reinterpret_cast<const float4*>(x)[i];
            if (bias) {
                const float b = bias[((i * 4) / step_b) % size_b];
                v.x += b; v.y += b; v.z += b; v.w += b;
            }
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ref) r = reinterpret_cast<const float4*>(ref)[i];
            v.x = bias_act_one(v.x, r.x, mode, alpha, scale);
            v.y = bias_act_one(v.y, r.y, mode, alpha, scale);
            v.z = bias_act_one(v.z, r.z, mode, alpha, scale);
            v.w = bias_act_one(v.w, r.w, mode, alpha, scale);
            reinterpret_cast<float4*>(y)[i] = v;
        }
        for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            float v = x[i];
            if (bias) v += bias[(i / step_b) % size_b];
            y[i] = bias_act_one(v, ref ? ref[i] : 0.f, mode, alpha, scale);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            float v = x[i] + bias[(i / step_b) % size_b];
            y[i] = bias_act_one(v, ref ? ref[i] : 0.f, mode, alpha, scale);
        }
    }
}

struct UfdParams {
    int major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w;
};

__device__ __forceinline__ int pos_mod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

// out[m,oy,ox,c] = sum over taps (ky,kx) whose up-sampled coordinate lands on a real input sample:
//   u = oy*down_y + ky - pad_y0,  u % up_y == 0,  0 <= u/up_y < in_h   (same for x)
// weighted by the FLIPPED kernel, kernel[kh-1-ky][kw-1-kx].
template <int V>
__global__ __launch_bounds__(256) void upfirdn2d_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ kern,
                                                        float* __restrict__ y, UfdParams p) {
    __shared__ float sk[64];
    for (int t = threadIdx.x; t < p.kh * p.kw; t += blockDim.x) sk[t] = kern[t];
    __syncthreads();
    const int mv = p.minor / V;
    const int64_t total = (int64_t)p.major * p.out_h * p.out_w * mv;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        int64_t r = e;
        const int c = (int)(r % mv) * V; r /= mv;
        const int ox = (int)(r % p.out_w); r /= p.out_w;
        const int oy = (int)(r % p.out_h); r /= p.out_h;
        const int m = (int)r;
        const int by = oy * p.down_y - p.pad_y0, bx = ox * p.down_x - p.pad_x0;
        float acc[V];
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = 0.f;
        for (int ky = pos_mod(-by, p.up_y); ky < p.kh; ky += p.up_y) {
            const int uy = by + ky;
            if (uy < 0) continue;
            const int iy = uy / p.up_y;
            if (iy >= p.in_h) break;
            for (int kx = pos_mod(-bx, p.up_x); kx < p.kw; kx += p.up_x) {
                const int ux = bx + kx;
                if (ux < 0) continue;
                const int ix = ux / p.up_x;
                if (ix >= p.in_w) break;
                const float wv = sk[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
                const float* src = x + (((int64_t)m * p.in_h + iy) * p.in_w + ix) * p.minor + c;
                if (V == 4) {
                    const float4 xv = *reinterpret_cast<const float4*>(src);
                    acc[0] = fmaf(xv.x, wv, acc[0]);
                    acc[1 % V] = fmaf(xv.y, wv, acc[1 % V]);
                    acc[2 % V] = fmaf(xv.z, wv, acc[2 % V]);
                    acc[3 % V] = fmaf(xv.w, wv, acc[3 % V]);
                } else {
                    acc[0] = fmaf(src[0], wv, acc[0]);
                }
            }
        }
        float* dst = y + (((int64_t)m * p.out_h + oy) * p.out_w + ox) * p.minor + c;
        if (V == 4) *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1 % V], acc[2 % V], acc[3 % V]);
        else dst[0] = acc[0];
    }
}

}  // namespace

extern "C" {

int wgs_bias_act(const float* x, const float* bias, const float* ref, float* y, int act, int grad,
                 float alpha, float scale, int64_t size_x, int step_b, int size_b, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && y, "wgs_bias_act: null pointer");
    WGS_CHECK_ARG(size_x >= 0, "wgs_bias_act: negative size");
    WGS_CHECK_ARG(!bias || (step_b > 0 && size_b > 0), "wgs_bias_act: bias needs step_b,size_b > 0");
    if (size_x == 0) return WGS_OK;
    const int mode = act * 10 + grad;
    int grid = wgs_cdiv(size_x / 4 + 1, 256);
    if (grid > 2048) grid = 2048;
    WGS_LAUNCH(bias_act_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, bias, ref, y,
                       mode, alpha, scale, size_x, step_b > 0 ? step_b : 1, size_b > 0 ? size_b : 1);
    WGS_CHECK_LAUNCH("bias_act_kernel");
    return WGS_OK;
}

int wgs_upfirdn2d(const float* x, const float* kernel, float* y, int major, int in_h, int in_w,
                  int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                  int pad_x1, int pad_y0, int pad_y1, wgs_stream_t stream) {
    WGS_CHECK_ARG(x && kernel && y, "wgs_upfirdn2d: null pointer");
    WGS_CHECK_ARG(major > 0 && in_h > 0 && in_w > 0 && minor > 0, "wgs_upfirdn2d: bad input shape");
    WGS_CHECK_ARG(kh > 0 && kw > 0 && kh * kw <= 64, "wgs_upfirdn2d: kernel %dx%d unsupported (<= 64 taps)", kh, kw);
    WGS_CHECK_ARG(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0, "wgs_upfirdn2d: up/down must be >= 1");
    UfdParams p;
    p.major = major; p.in_h = in_h; p.in_w = in_w; p.minor = minor; p.kh = kh; p.kw = kw;
    p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
    p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
    p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
    WGS_CHECK_ARG(p.out_h > 0 && p.out_w > 0, "wgs_upfirdn2d: empty output %dx%d", p.out_h, p.out_w);
    const bool v4 = (minor % 4 == 0);
    if (v4 && kh == 4 && kw == 4 && up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1) {
        // the StyleGAN2 blur and its gradient on wide NHWC tensors: sliding-window form (fir4.h)
        wgsfir::launch_fir4<false>(x, kernel, y, major, in_h, in_w, p.out_h, p.out_w, minor, pad_y0, pad_x0, nullptr, nullptr,
                                   nullptr, (hipStream_t)stream);
        WGS_CHECK_LAUNCH("fir4_kernel");
        return WGS_OK;
    }
    const int64_t total = (int64_t)major * p.out_h * p.out_w * (v4 ? minor / 4 : minor);
    int grid = wgs_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    if (v4) WGS_LAUNCH(upfirdn2d_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, kernel, y, p);
    else    WGS_LAUNCH(upfirdn2d_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, kernel, y, p);
    WGS_CHECK_LAUNCH("upfirdn2d_kernel");
    return WGS_OK;
}

}  // extern "C"
