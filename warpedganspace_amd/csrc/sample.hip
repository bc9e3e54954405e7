// One launch draws a training step's batch in HBM (the reference's lib/trainer.py:195-221 and lib/aux.py:39-53 do it on the host with
// five torch calls and four .cuda() copies; the step engine did it with ~25 tiny device launches at the head of every step):
//
//   z[b, j]  ~ N(0, 1), truncated to [-trunc, trunc] when 0 < trunc != 1 (inverse CDF in fp64, as aux.sample_z's device path)
//   idx[b]   ~ U{0 .. K-1}                                              (torch.randint, :203)
//   mag[b]   : pool = [neg_0 .. neg_{B-1}, pos_0 .. pos_{B-1}],  neg_i = (lo - hi) u_i - lo,  pos_i = (lo - hi) u'_i + hi   (:212-216)
//              B of the 2B pool entries WITHOUT replacement, weights 0, 1, .., 2B-1 (:218-221: torch.multinomial of arange — entry 0 is
//              never drawn, the positive half is favoured), in the order torch.multinomial returns them: decreasing key w_i / e_i,
//              e_i ~ Exp(1) (ATen's multinomial without replacement is this exponential race followed by a top-k)
//
// Counter-based generator (Philox4x32-10, the construction of Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): the
// stream is a pure function of (seed, step, element), so the call is stateless — no generator object crosses the C ABI — and
// reproducible; every (seed, step) pair is its own stream.
#include <cstdint>
#include "wgs_common.h"
#include "../../include/wgs.h"

namespace {

struct u4 { unsigned x, y, z, w; };

__device__ __forceinline__ u4 philox4x32_10(u4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x, p1 = (unsigned long long)0xCD9E8D57u * c.z;
        const u4 n = {(unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0};
        c = n;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c;
}
// uniform in (0, 1): never 0 (logarithms below), never 1.  For the largest draw (2^24 - 1, resp. 2^53 - 1) the "+ 0.5" rounds up to the
// next power of two and the product would be exactly 1 (probability 2^-24 / 2^-53 per draw): clamped to the largest value below 1.
__device__ __forceinline__ float u01(unsigned v) { return fminf(((float)(v >> 8) + 0.5f) * (1.0f / 16777216.0f), 0x1.fffffep-1f); }
__device__ __forceinline__ double u01d(unsigned a, unsigned b) {
    return fmin(((double)(((unsigned long long)a << 21) ^ (unsigned long long)(b >> 11)) + 0.5) * (1.0 / 9007199254740992.0), 0x1.fffffffffffffp-1);
}

// counter = (element, 0, stream id, step_hi); key = seed.  stream id: 0 = z, 1 = idx, 2 = pool magnitudes, 3 = exponential race
__device__ __forceinline__ u4 draw(unsigned long long seed, unsigned long long step, unsigned sid, unsigned elem) {
    const u4 c = {elem, (unsigned)step, sid, (unsigned)(step >> 32)};
    return philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
}

constexpr int MAXB = 1024;

__global__ __launch_bounds__(256) void sample_step_kernel(float* __restrict__ z, int64_t* __restrict__ idx, float* __restrict__ mag, int B,
                                                          int d, int K, float lo, float hi, float trunc, unsigned long long seed,
                                                          unsigned long long step, int zblocks) {
    if ((int)blockIdx.x < zblocks) {
        // ---- latent codes: one Philox call -> four values ----
        const long n = (long)B * d;
        const long q = (long)blockIdx.x * 256 + threadIdx.x;          // quad index
        if (q * 4 >= n) return;
        float v[4];
        if (!(trunc > 0.f) || trunc == 1.0f) {
            const u4 r = draw(seed, step, 0u, (unsigned)q);
            // Box-Muller on two pairs
            const float r0 = sqrtf(-2.f * __logf(u01(r.x))), r1 = sqrtf(-2.f * __logf(u01(r.z)));
            float s0, c0, s1, c1;
            __sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
            __sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
            v[0] = r0 * c0; v[1] = r0 * s0; v[2] = r1 * c1; v[3] = r1 * s1;
        } else {
            // truncated standard normal by its inverse CDF: Phi^-1( Phi(-t) + u (Phi(t) - Phi(-t)) ), fp64 (the tails)
            const double t = (double)trunc, plo = 0.5 * (1.0 + erf(-t * 0.7071067811865476)), phi = 0.5 * (1.0 + erf(t * 0.7071067811865476));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u4 r = draw(seed, step, 0u, (unsigned)(2 * q + h));
                const double ua = u01d(r.x, r.y), ub = u01d(r.z, r.w);
                double pa = plo + ua * (phi - plo), pb = plo + ub * (phi - plo);
                pa = fmin(fmax(pa, 1e-12), 1.0 - 1e-12); pb = fmin(fmax(pb, 1e-12), 1.0 - 1e-12);
                v[2 * h] = (float)(1.4142135623730951 * erfinv(2.0 * pa - 1.0));
                v[2 * h + 1] = (float)(1.4142135623730951 * erfinv(2.0 * pb - 1.0));
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (q * 4 + e < n) z[q * 4 + e] = v[e];
        return;
    }
    // ---- last workgroup: path indices and shift magnitudes ----
    __shared__ float key[2 * MAXB], pool[2 * MAXB];
    const int n2 = 2 * B;
    for (int i = threadIdx.x; i < B; i += 256) {
        const u4 r = draw(seed, step, 1u, (unsigned)i);
        // unbiased enough for K << 2^32 and exact for powers of two: multiply-shift of a 32-bit draw
        idx[i] = (int64_t)(((unsigned long long)r.x * (unsigned long long)K) >> 32);
    }
    for (int i = threadIdx.x; i < n2; i += 256) {
        const u4 r = draw(seed, step, 2u, (unsigned)i);
        const float u = u01(r.x);
        pool[i] = i < B ? (lo - hi) * u - lo : (lo - hi) * u + hi;
        // > 0: no 0 / 0 for entry 0; and large enough that i / e stays finite for the largest i (2047 / 1e-30 = 2e33): with FLT_MIN here every
        // i >= 4 overflowed to +inf whenever the fast log returned <= 0 for a u just below 1 (~2^-24 per draw), and several inf keys tie
        const float e = fmaxf(-__logf(u01(draw(seed, step, 3u, (unsigned)i).x)), 1e-30f);
        key[i] = (float)i / e;                     // weight i: entry 0 has key 0 and is never among the B largest
    }
    __syncthreads();
    // rank by counting (2B <= 2048 keys): entry i goes to position #{j : key_j > key_i, or equal and j < i}
    for (int i = threadIdx.x; i < n2; i += 256) {
        const float ki = key[i];
        int rank = 0;
        for (int j = 0; j < n2; ++j) rank += (key[j] > ki) || (key[j] == ki && j < i);
        if (rank < B) mag[rank] = pool[i];
    }
}

}  // namespace

extern "C" {

int wgs_sample_step(float* z, int64_t* idx, float* mag, int B, int d, int K, float lo, float hi, float trunc, uint64_t seed, uint64_t step,
                    wgs_stream_t stream) {
    WGS_CHECK_ARG(z && idx && mag, "wgs_sample_step: null pointer");
    WGS_CHECK_ARG(B > 0 && B <= MAXB && d > 0 && K > 0, "wgs_sample_step: B=%d (<= %d), d=%d, K=%d", B, MAXB, d, K);
    WGS_CHECK_ARG((long)B * d < (1L << 33), "wgs_sample_step: B * d too large");
    const long quads = ((long)B * d + 3) / 4;
    const int zblocks = (int)((quads + 255) / 256);
    WGS_LAUNCH(sample_step_kernel, dim3((unsigned)(zblocks + 1)), dim3(256), 0, (hipStream_t)stream, z, idx, mag, B, d, K, lo, hi, trunc,
               (unsigned long long)seed, (unsigned long long)step, zblocks);
    WGS_CHECK_LAUNCH("sample_step_kernel");
    return WGS_OK;
}

}  // extern "C"
