/*
 * mfs_oracle.c -- plain C fp64 restatement of the reference's 1-D moment-filter step (TEST INFRASTRUCTURE).
 *
 * Used only as (i) a second checker next to the NumPy oracle and (ii) the `cpu_baseline` leg of bench.py (kind
 * "port": the reference's JAX-CPU path cannot run in this image).  Never linked into
 * libmfs_hip.so, never imported by mfs_amd.
 *
 * Follows (paths relative to /root/reference):
 *   mfs/one_dim/quadtures.py:122-133   Hankel G/H, lower Cholesky, K = R^-1 H R^-T, eigh of (K+K^T)/2,
 *                                      weights V[0,:]^2, nodes scale*lambda+mean
 *   mfs/one_dim/filtering.py:73-86, 140-158, 217-237   the three scan bodies
 *   mfs/one_dim/moments.py:141-179     TME transition moments, here in operator form
 *       E[(X'-c)^n | x] = sum_k Q_k(u) n!/(n-k)! (x-c)^(n-k)  (tables derived by SymPy in oracle/tme_sympy.py)
 *   mfs/one_dim/moments.py:70-74,182-255   normal closure (binomial sum restated as the 3-term recurrence)
 * The symmetric eigensolver is the classical Householder tridiagonalisation + implicit QL (EISPACK tred2/tql2, the
 * algorithm family behind LAPACK's syev that XLA calls), written out here so the file has no dependencies.
 * OpenMP parallelises over replicates only.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXN 32
#define MAXK 8
#define MAXJ 15

/* ---- symmetric eigen decomposition: V (n x n, row-major, V[i*n+j]) in: symmetric matrix; out: eigenvectors in
 *      columns, d: eigenvalues ascending.  tred2 + tql2 (EISPACK / JAMA, public domain). ---- */
static void tred2(int n, double* V, double* d, double* e) {
    for (int j = 0; j < n; j++) d[j] = V[(n - 1) * n + j];
    for (int i = n - 1; i > 0; i--) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; k++) scale += fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; j++) { d[j] = V[(i - 1) * n + j]; V[i * n + j] = 0.0; V[j * n + i] = 0.0; }
        } else {
            for (int k = 0; k < i; k++) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1];
            double g = sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; j++) e[j] = 0.0;
            for (int j = 0; j < i; j++) {
                f = d[j];
                V[j * n + i] = f;
                g = e[j] + V[j * n + j] * f;
                for (int k = j + 1; k <= i - 1; k++) { g += V[k * n + j] * d[k]; e[k] += V[k * n + j] * f; }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; j++) { e[j] /= h; f += e[j] * d[j]; }
            double hh = f / (h + h);
            for (int j = 0; j < i; j++) e[j] -= hh * d[j];
            for (int j = 0; j < i; j++) {
                f = d[j]; g = e[j];
                for (int k = j; k <= i - 1; k++) V[k * n + j] -= (f * e[k] + g * d[k]);
                d[j] = V[(i - 1) * n + j];
                V[i * n + j] = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; i++) {
        V[(n - 1) * n + i] = V[i * n + i];
        V[i * n + i] = 1.0;
        double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; k++) d[k] = V[k * n + i + 1] / h;
            for (int j = 0; j <= i; j++) {
                double g = 0.0;
                for (int k = 0; k <= i; k++) g += V[k * n + i + 1] * V[k * n + j];
                for (int k = 0; k <= i; k++) V[k * n + j] -= g * d[k];
            }
        }
        for (int k = 0; k <= i; k++) V[k * n + i + 1] = 0.0;
    }
    for (int j = 0; j < n; j++) { d[j] = V[(n - 1) * n + j]; V[(n - 1) * n + j] = 0.0; }
    V[(n - 1) * n + n - 1] = 1.0;
    e[0] = 0.0;
}

static int tql2(int n, double* V, double* d, double* e) {
    for (int i = 1; i < n; i++) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n; l++) {
        double t = fabs(d[l]) + fabs(e[l]);
        if (t > tst1) tst1 = t;
        int m = l;
        while (m < n) { if (fabs(e[m]) <= eps * tst1) break; m++; }
        if (m >= n) m = n - 1;
        if (m > l) {
            int iter = 0;
            do {
                if (++iter > 200) return -1;
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; i++) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c, el1 = e[l + 1], s = 0.0, s2 = 0.0;
                for (int i = m - 1; i >= l; i--) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < n; k++) {
                        h = V[k * n + i + 1];
                        V[k * n + i + 1] = s * V[k * n + i] + c * h;
                        V[k * n + i] = c * V[k * n + i] - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (fabs(e[l]) > eps * tst1);
        }
        d[l] = d[l] + f;
        e[l] = 0.0;
    }
    return 0;
}

/* diagnostics of stable = 1: rules whose LDL^T had a negative pivot (completed by eps), counted per replicate and half-step
 * when a counter array [B][2] has been registered (tests / tools only; not thread-safe across concurrent callers) */
static int* g_completions = 0;
void mfs_oracle_set_completion_counter(int* counts) { g_completions = counts; }

/* ---- moment quadrature (quadtures.py:83-133).  Returns 0, or 1 when poisoned (outputs set to NaN); *completed = 1 when
 *      stable and some pivot was negative. ---- */
static int quadrature(int n, const double* ms, double mean, double scale, int stable, double* w, double* x, int* completed) {
    double R[MAXN * MAXN], K[MAXN * MAXN], d[MAXN], e[MAXN];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) { R[i * n + j] = ms[i + j]; K[i * n + j] = ms[i + j + 1]; }
    int bad = 0;
    if (!stable) {
        for (int j = 0; j < n && !bad; j++) {
            double s = R[j * n + j];
            for (int k = 0; k < j; k++) s -= R[j * n + k] * R[j * n + k];
            if (!(s > 0.0)) { bad = 1; break; }
            double r = sqrt(s);
            R[j * n + j] = r;
            for (int i = j + 1; i < n; i++) {
                double t = R[i * n + j];
                for (int k = 0; k < j; k++) t -= R[i * n + k] * R[j * n + k];
                R[i * n + j] = t / r;
            }
        }
    } else { /* mfs/utils.py:495-538 */
        double fro = 0.0, dd[MAXN];
        for (int i = 0; i < n * n; i++) fro += R[i] * R[i];
        const double eps = 1e-8 * sqrt(fro);
        for (int j = 0; j < n; j++) {
            double s = R[j * n + j];
            for (int k = 0; k < j; k++) s -= R[j * n + k] * R[j * n + k] * dd[k];
            dd[j] = s;
            for (int i = j + 1; i < n; i++) {
                double t = R[i * n + j];
                for (int k = 0; k < j; k++) t -= R[i * n + k] * R[j * n + k] * dd[k];
                R[i * n + j] = t / s;
            }
        }
        for (int j = 0; j < n; j++) {
            const double f = dd[j] < 0.0 ? eps : sqrt(dd[j]);
            if (dd[j] < 0.0 && completed) *completed = 1;
            R[j * n + j] = f;
            for (int i = j + 1; i < n; i++) R[i * n + j] *= f;
        }
    }
    if (!bad) {
        for (int c = 0; c < n; c++) /* X = R^-1 H */
            for (int i = 0; i < n; i++) {
                double s = K[i * n + c];
                for (int k = 0; k < i; k++) s -= R[i * n + k] * K[k * n + c];
                K[i * n + c] = s / R[i * n + i];
            }
        for (int i = 0; i < n; i++) /* K = X R^-T */
            for (int j = 0; j < n; j++) {
                double s = K[i * n + j];
                for (int k = 0; k < j; k++) s -= K[i * n + k] * R[j * n + k];
                K[i * n + j] = s / R[j * n + j];
            }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < i; j++) { double s = 0.5 * (K[i * n + j] + K[j * n + i]); K[i * n + j] = s; K[j * n + i] = s; }
        for (int i = 0; i < n * n; i++) if (!isfinite(K[i])) bad = 1;
    }
    if (!bad) { tred2(n, K, d, e); if (tql2(n, K, d, e)) bad = 1; }
    if (bad) { for (int i = 0; i < n; i++) { w[i] = NAN; x[i] = NAN; } return 1; }
    for (int i = 0; i < n; i++) { w[i] = K[i] * K[i]; x[i] = scale * d[i] + mean; }
    return 0;
}

static double horner(const double* row, int J, double u) {
    double acc = row[J];
    for (int j = J - 1; j >= 0; j--) acc = acc * u + row[j];
    return acc;
}

static double likelihood(int kind, const double* lp, double y, double x) {
    if (kind == 0) {
        const double z = lp[0] + x * (lp[1] + x * (lp[2] + x * lp[3]));
        const double p = 1.0 / (1.0 + exp(-z));
        return y > 0.5 ? p : 1.0 - p;
    } else if (kind == 1) {
        const double rate = log(1.0 + exp(lp[0] * x));
        return exp(y * log(rate) - rate - lgamma(y + 1.0));
    }
    const double r = y - (lp[0] * x + lp[1]);
    return exp(-0.5 * r * r / lp[2]) / sqrt(6.283185307179586476925 * lp[2]);
}

/*
 * mode 0 raw / 1 central / 2 scaled; trans_kind 0 operator (coef rows Q_1..Q_K, then variance) / 1 gaussian (rows
 * mean poly, variance poly; mu = mean_x_coef x + P_m(u)); umap 0 u = x / 1 u = tanh x.  Layouts as include/mfs_hip.h.
 */
int mfs_oracle_filter_1d(int mode, int N, int T, int B, int trans_kind, int umap, int K, int J, int n_rows,
                         const double* coef, int coef_batched, double mean_x_coef, int lik_kind, int n_lik,
                         const double* lik, int lik_batched, const double* m0, int m0_batched, const double* mean0,
                         const double* scale0, const double* ys, int stable, double* out_mom, double* out_mean,
                         double* out_scale, double* out_nell, int nthreads) {
    if (N < 2 || N > MAXN || K > MAXK || J > MAXJ) return -1;
    const int M2 = 2 * N, J1 = J + 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int b = 0; b < B; b++) {
        const double* cf = coef + (coef_batched ? (size_t)b * n_rows * J1 : 0);
        const double* lp = lik + (lik_batched ? (size_t)b * n_lik : 0);
        double lpp[4] = {0, 0, 0, 0};
        for (int i = 0; i < n_lik && i < 4; i++) lpp[i] = lp[i];
        double ms[2 * MAXN], w[MAXN], x[MAXN], nw[2 * MAXN];
        memcpy(ms, m0 + (m0_batched ? (size_t)b * M2 : 0), sizeof(double) * M2);
        double mean = mode ? mean0[m0_batched ? b : 0] : 0.0;
        double scale = mode == 2 ? scale0[m0_batched ? b : 0] : 1.0;
        double nell = 0.0;
        for (int t = 0; t < T; t++) {
            const double y = ys[(size_t)b * T + t];
            /* prediction */
            int comp = 0;
            quadrature(N, ms, mean, scale, stable, w, x, &comp);
            if (comp && g_completions) g_completions[2 * b] += 1;
            double c = 0.0, isc = 1.0;
            if (mode) {
                double mu = 0.0, var = 0.0;
                for (int i = 0; i < N; i++) {
                    const double u = umap ? tanh(x[i]) : x[i];
                    if (trans_kind == 0) { mu += w[i] * (x[i] + horner(cf, J, u)); var += w[i] * horner(cf + K * J1, J, u); }
                    else { mu += w[i] * (mean_x_coef * x[i] + horner(cf, J, u)); var += w[i] * horner(cf + J1, J, u); }
                }
                mean = mu; c = mu;
                if (mode == 2) { scale = sqrt(var); isc = 1.0 / scale; }
            }
            for (int n = 0; n < M2; n++) nw[n] = 0.0;
            for (int i = 0; i < N; i++) {
                const double u = umap ? tanh(x[i]) : x[i];
                if (trans_kind == 0) {
                    double Q[MAXK + 1], pw[2 * MAXN];
                    Q[0] = 1.0;
                    for (int k = 1; k <= K; k++) Q[k] = horner(cf + (k - 1) * J1, J, u);
                    const double dx = x[i] - c;
                    pw[0] = 1.0;
                    for (int n = 1; n < M2; n++) pw[n] = pw[n - 1] * dx;
                    double sn = w[i];
                    for (int n = 0; n < M2; n++) {
                        double val = 0.0, ff = 1.0;
                        for (int k = 0; k <= K && k <= n; k++) { val += Q[k] * ff * pw[n - k]; ff *= (double)(n - k); }
                        nw[n] += sn * val;
                        sn *= isc;
                    }
                } else {
                    const double m = mean_x_coef * x[i] + horner(cf, J, u) - c, v = horner(cf + J1, J, u);
                    double e2 = 1.0, e1 = m, sn = w[i];
                    nw[0] += sn; sn *= isc; nw[1] += sn * m;
                    for (int n = 2; n < M2; n++) {
                        const double en = m * e1 + (double)(n - 1) * v * e2;
                        sn *= isc; nw[n] += sn * en; e2 = e1; e1 = en;
                    }
                }
            }
            memcpy(ms, nw, sizeof(double) * M2);
            /* update */
            comp = 0;
            quadrature(N, ms, mean, scale, stable, w, x, &comp);
            if (comp && g_completions) g_completions[2 * b + 1] += 1;
            double py = 0.0, mx = 0.0;
            for (int i = 0; i < N; i++) { w[i] *= likelihood(lik_kind, lpp, y, x[i]); py += w[i]; mx += w[i] * x[i]; }
            if (mode) { mean = mx / py; c = mean; }
            if (mode == 2) {
                double v = 0.0;
                for (int i = 0; i < N; i++) v += w[i] * (x[i] - c) * (x[i] - c);
                scale = sqrt(v / py); isc = 1.0 / scale;
            }
            for (int n = 0; n < M2; n++) nw[n] = 0.0;
            for (int i = 0; i < N; i++) {
                const double dx = (x[i] - c) * isc;
                double p = w[i];
                for (int n = 0; n < M2; n++) { nw[n] += p; p *= dx; }
            }
            for (int n = 0; n < M2; n++) ms[n] = nw[n] / py;
            nell -= log(py);
            if (out_mom) memcpy(out_mom + ((size_t)b * T + t) * M2, ms, sizeof(double) * M2);
            if (out_mean) out_mean[(size_t)b * T + t] = mean;
            if (out_scale) out_scale[(size_t)b * T + t] = scale;
        }
        out_nell[b] = nell;
    }
    return 0;
}

int mfs_oracle_quadrature_1d(int N, int B, const double* ms, const double* mean, const double* scale, int stable,
                             double* w, double* x) {
    if (N < 2 || N > MAXN) return -1;
    for (int b = 0; b < B; b++)
        quadrature(N, ms + (size_t)b * 2 * N, mean ? mean[b] : 0.0, scale ? scale[b] : 1.0, stable, w + (size_t)b * N,
                   x + (size_t)b * N, 0);
    return 0;
}

int mfs_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
