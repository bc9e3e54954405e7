/*
 * rbf_ref.c — TEST INFRASTRUCTURE (oracle).  Plain-C restatement of the reference's RBF warping
 * field, lib/support_sets.py:81-101, evaluated in double precision:
 *     sv    = SUPPORT_SETS[idx[b]]  viewed [n2, d]          (:83-84, one-hot matmul == row gather)
 *     alpha = ALPHAS[idx[b]]                                (:87)
 *     gamma = exp(LOGGAMMA[idx[b]]) or the constant gamma   (:90-93)
 *     D_i   = z_b - sv_i                                    (:96)
 *     g     = -2 * sum_i alpha_i gamma exp(-gamma |D_i|^2) D_i   (:97-98)
 *     out   = g / |g|                                       (:101)
 * and the analytic gradient of  L = sum_b <gout_b, out_b>  w.r.t. SUPPORT_SETS / ALPHAS / LOGGAMMA / z
 * (what autograd produces for the reference module; pinned by tests/golden/support_sets_*.npz).
 * Build: gcc -O2 -shared -fPIC rbf_ref.c -o _build/librbf_ref.so -lm   (done by __graft_entry__.build()).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void rbf_ref_forward(const float* table, const float* alphas, const float* loggamma, double gamma_c,
                     const int64_t* idx, const float* z, double* out, double* g_raw, int B, int K,
                     int n2, int d) {
    (void)K;
    for (int b = 0; b < B; ++b) {
        const int k = (int)idx[b];
        const double gamma = loggamma ? exp((double)loggamma[k]) : gamma_c;
        double* g = g_raw + (size_t)b * d;
        memset(g, 0, sizeof(double) * d);
        for (int i = 0; i < n2; ++i) {
            const float* s = table + ((size_t)k * n2 + i) * d;
            double r2 = 0.0;
            for (int j = 0; j < d; ++j) {
                const double D = (double)z[(size_t)b * d + j] - (double)s[j];
                r2 += D * D;
            }
            const double c = (double)alphas[(size_t)k * n2 + i] * gamma * exp(-gamma * r2);
            for (int j = 0; j < d; ++j) g[j] += c * ((double)z[(size_t)b * d + j] - (double)s[j]);
        }
        double nn = 0.0;
        for (int j = 0; j < d; ++j) { g[j] *= -2.0; nn += g[j] * g[j]; }
        nn = sqrt(nn);
        for (int j = 0; j < d; ++j) out[(size_t)b * d + j] = g[j] / nn;
    }
}

/* Gradients are accumulated into zero-initialised dtable [K,n2*d], dalphas [K,n2], dloggamma [K];
 * dz [B,d] is overwritten. Any of dalphas/dloggamma/dz may be NULL. */
void rbf_ref_backward(const float* table, const float* alphas, const float* loggamma, double gamma_c,
                      const int64_t* idx, const float* z, const float* gout, double* dtable,
                      double* dalphas, double* dloggamma, double* dz, int B, int K, int n2, int d) {
    double* g = (double*)malloc(sizeof(double) * d);
    double* h = (double*)malloc(sizeof(double) * d);
    double* D = (double*)malloc(sizeof(double) * d);
    (void)K;
    for (int b = 0; b < B; ++b) {
        const int k = (int)idx[b];
        const double gamma = loggamma ? exp((double)loggamma[k]) : gamma_c;
        memset(g, 0, sizeof(double) * d);
        for (int i = 0; i < n2; ++i) {
            const float* s = table + ((size_t)k * n2 + i) * d;
            double r2 = 0.0;
            for (int j = 0; j < d; ++j) { D[j] = (double)z[(size_t)b * d + j] - (double)s[j]; r2 += D[j] * D[j]; }
            const double c = -2.0 * (double)alphas[(size_t)k * n2 + i] * gamma * exp(-gamma * r2);
            for (int j = 0; j < d; ++j) g[j] += c * D[j];
        }
        double nn = 0.0, ugo = 0.0;
        for (int j = 0; j < d; ++j) nn += g[j] * g[j];
        nn = sqrt(nn);
        for (int j = 0; j < d; ++j) ugo += g[j] / nn * (double)gout[(size_t)b * d + j];
        for (int j = 0; j < d; ++j) h[j] = ((double)gout[(size_t)b * d + j] - g[j] / nn * ugo) / nn;
        if (dz) for (int j = 0; j < d; ++j) dz[(size_t)b * d + j] = 0.0;
        double dgam = 0.0;
        for (int i = 0; i < n2; ++i) {
            const float* s = table + ((size_t)k * n2 + i) * d;
            double r2 = 0.0, hd = 0.0;
            for (int j = 0; j < d; ++j) {
                D[j] = (double)z[(size_t)b * d + j] - (double)s[j];
                r2 += D[j] * D[j];
                hd += h[j] * D[j];
            }
            const double a = (double)alphas[(size_t)k * n2 + i];
            const double e = exp(-gamma * r2);
            const double coef = 2.0 * a * gamma * e;
            for (int j = 0; j < d; ++j) {
                const double ds = coef * (h[j] - 2.0 * gamma * hd * D[j]);
                dtable[((size_t)k * n2 + i) * d + j] += ds;
                if (dz) dz[(size_t)b * d + j] -= ds;
            }
            if (dalphas) dalphas[(size_t)k * n2 + i] += -2.0 * gamma * e * hd;
            dgam += -2.0 * a * hd * e * (1.0 - gamma * r2);
        }
        if (dloggamma && loggamma) dloggamma[k] += gamma * dgam;
    }
    free(g); free(h); free(D);
}
