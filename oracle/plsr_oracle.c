/*
 * CPU oracle (plain C, fp64) for the plskern / plsnipals hot path of Jchemo.jl.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker
 * (or as the timed CPU baseline).  The product (jchemo.jl_amd/) never links
 * or loads it and has no CPU fallback.
 *
 * PARITY UNPINNED: the reference is pure Julia (no toolchain in this image) and
 * ships no tests or fixtures for this path (test/runtests.jl:1-2).  This file is
 * pinned by invariants, by the independent numpy/LAPACK restatement in
 * plsr_oracle.py, by scikit-learn, and by plskern == plsnipals agreement
 * (tests/test_oracle.py).
 *
 * It keeps the REFERENCE's schedule (src/plskern.jl:112-178): weighted means
 * pass, in-place centring, XtY once, then TWO matrix-vector sweeps over X per
 * latent variable (t = X r at :162, zp = X' D t at :167), so that it is a
 * like-for-like stand-in for the Julia/OpenBLAS CPU path when timed.
 * Column-major storage as in Julia (F5).  OpenMP over rows / columns; every
 * reduction is computed by one thread in index order, so results do not depend
 * on the thread count.
 *
 * Paths cited below are relative to /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define XC(i, j) X[(size_t)(i) + (size_t)(j) * (size_t)ldx]
#define YC(i, k) Y[(size_t)(i) + (size_t)(k) * (size_t)ldy]

/* ---- portable generator: splitmix64 output k is a pure function of (seed,k) ---- */
static inline double sm64_u01(uint64_t seed, uint64_t k)
{
    uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

/* rows [row0,row0+n) of the column-major-filled n_total x p matrix -> out (ld) */
void orc_fill_uniform(double *out, int64_t n, int64_t p, int64_t ld, int64_t row0, int64_t n_total,
                      uint64_t seed)
{
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < p; ++j)
        for (int64_t i = 0; i < n; ++i)
            out[(size_t)i + (size_t)j * (size_t)ld] =
                sm64_u01(seed, (uint64_t)(row0 + i) + (uint64_t)j * (uint64_t)n_total);
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int t)
{
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

/* ---- dominant left singular vector: one-sided (Hestenes) Jacobi on a copy of K ----
 * src/plskern.jl:150-155, src/plsnipals.jl:72-77 (LAPACK dgesdd there; sign free, F3).
 * K is p x q column-major (ld = p).  Sign rule: largest-|.| entry of w positive. */
static void dominant_left_sv(const double *K, int64_t p, int64_t q, double *w, double *work /* p*q */)
{
    if (q == 1) {
        double s = 0.0;
        for (int64_t j = 0; j < p; ++j) s += K[j] * K[j];
        s = sqrt(s);
        for (int64_t j = 0; j < p; ++j) w[j] = K[j] / s;
        return;
    }
    double *A = work;
    memcpy(A, K, sizeof(double) * (size_t)p * (size_t)q);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int64_t a = 0; a < q - 1; ++a)
            for (int64_t b = a + 1; b < q; ++b) {
                double *ca = A + a * p, *cb = A + b * p;
                double aa = 0, bb = 0, ab = 0;
                for (int64_t j = 0; j < p; ++j) { aa += ca[j] * ca[j]; bb += cb[j] * cb[j]; ab += ca[j] * cb[j]; }
                if (ab == 0.0 || fabs(ab) <= 1e-300) continue;
                double rel = fabs(ab) / sqrt(aa * bb);
                if (rel > off) off = rel;
                if (rel < 1e-17) continue;
                double zeta = (bb - aa) / (2.0 * ab);
                double tn = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + tn * tn), sn = cs * tn;
                for (int64_t j = 0; j < p; ++j) {
                    double x = ca[j], y = cb[j];
                    ca[j] = cs * x - sn * y;
                    cb[j] = sn * x + cs * y;
                }
            }
        if (off < 1e-15) break;
    }
    int64_t best = 0; double bestn = -1.0;
    for (int64_t k = 0; k < q; ++k) {
        double s = 0; const double *c = A + k * p;
        for (int64_t j = 0; j < p; ++j) s += c[j] * c[j];
        if (s > bestn) { bestn = s; best = k; }
    }
    double nr = sqrt(bestn), big = 0.0; const double *c = A + best * p;
    for (int64_t j = 0; j < p; ++j) if (fabs(c[j]) > fabs(big)) big = c[j];
    double sg = (big < 0 ? -1.0 : 1.0) / nr;
    for (int64_t j = 0; j < p; ++j) w[j] = c[j] * sg;
}

/* Dot-product kernels with 8 interleaved partial sums combined in a fixed order: the result does not depend on the
 * thread count, and the compiler may keep the 8 lanes in SIMD registers without -ffast-math (a single running sum is a
 * 4-cycle dependent chain per element: ~6 GB/s per core, which made this port slower than the OpenBLAS dgemv it
 * stands in for when few cores are available). */
static inline double dot8(const double *a, const double *b, int64_t n)
{
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int l = 0; l < 8; ++l) s[l] += a[i + l] * b[i + l];
    for (int l = 0; i < n; ++i, ++l) s[l] += a[i] * b[i];
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}
static inline double dot8w(const double *a, const double *d, const double *b, int64_t n)   /* sum a_i * (d_i * b_i) */
{
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int l = 0; l < 8; ++l) s[l] += a[i + l] * (d[i + l] * b[i + l]);
    for (int l = 0; i < n; ++i, ++l) s[l] += a[i] * (d[i] * b[i]);
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}
static inline double wvar8(const double *a, const double *d, double m, int64_t n)   /* sum d_i (a_i - m)^2 */
{
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int l = 0; l < 8; ++l) { double e = a[i + l] - m; s[l] += d[i + l] * e * e; }
    for (int l = 0; i < n; ++i, ++l) { double e = a[i] - m; s[l] += d[i] * e * e; }
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

/* ---- preamble: src/plskern.jl:114-130 == src/plsnipals.jl:39-56 ----
 * mweight (utility.jl:715-723), colmean (:195), colstd/colvar two-pass (:264,:314-323),
 * center! (:76-81) / cscale! (:482-487).  X, Y overwritten.  d = normalised weights. */
static void preamble(double *X, int64_t ldx, double *Y, int64_t ldy, const double *w, int64_t n, int64_t p,
                     int64_t q, int scal, double *d, double *xmeans, double *xscales, double *ymeans,
                     double *yscales)
{
    double sw = 0.0;
    if (w) for (int64_t i = 0; i < n; ++i) sw += w[i]; else sw = (double)n;
    for (int64_t i = 0; i < n; ++i) d[i] = (w ? w[i] : 1.0) / sw;
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < p + q; ++j) {
        double *col = j < p ? &XC(0, j) : &YC(0, j - p);
        double m = dot8(d, col, n);
        double s = 1.0;
        if (scal) {
            s = sqrt(wvar8(col, d, m, n));
            for (int64_t i = 0; i < n; ++i) col[i] = (col[i] - m) / s;
        } else {
            for (int64_t i = 0; i < n; ++i) col[i] = col[i] - m;
        }
        if (j < p) { xmeans[j] = m; xscales[j] = s; } else { ymeans[j - p] = m; yscales[j - p] = s; }
    }
}

/* K = X' D Y  (p x q, column-major ld p) — src/plskern.jl:131-132, src/plsnipals.jl:71 */
static void xtdy(const double *X, int64_t ldx, const double *Y, int64_t ldy, const double *d, int64_t n,
                 int64_t p, int64_t q, double *K)
{
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < p; ++j) {
        const double *xc = &XC(0, j);
        for (int64_t k = 0; k < q; ++k) {
            K[j + k * p] = dot8w(xc, d, &YC(0, k), n);
        }
    }
}

/* t = X v  (dgemv-N, src/plskern.jl:162 / src/plsnipals.jl:78); row blocks, columns in order */
static void xv(const double *X, int64_t ldx, int64_t n, int64_t p, const double *v, double *t)
{
    const int64_t RB = 2048;
#pragma omp parallel for schedule(static)
    for (int64_t i0 = 0; i0 < n; i0 += RB) {
        int64_t i1 = i0 + RB < n ? i0 + RB : n;
        for (int64_t i = i0; i < i1; ++i) t[i] = 0.0;
        for (int64_t j = 0; j < p; ++j) {
            const double *xc = &XC(0, j); double vj = v[j];
            for (int64_t i = i0; i < i1; ++i) t[i] += xc[i] * vj;
        }
    }
}

/* z = X' u  (dgemv-T, src/plskern.jl:167 / src/plsnipals.jl:81) */
static void xtu(const double *X, int64_t ldx, int64_t n, int64_t p, const double *u, double *z)
{
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < p; ++j) {
        z[j] = dot8(&XC(0, j), u, n);
    }
}

/* `plskern!` — src/plskern.jl:112-178.  Outputs column-major; T n x nlv (ld n), P/R/W p x nlv, C q x nlv.
 * Returns the clamped nlv. */
int orc_plskern(double *X, int64_t ldx, double *Y, int64_t ldy, const double *w, int64_t n, int64_t p,
                int64_t q, int nlv, int scal, double *T, double *P, double *R, double *W, double *C, double *TT,
                double *xmeans, double *xscales, double *ymeans, double *yscales, double *wnorm)
{
    if (nlv > n) nlv = (int)n;
    if (nlv > p) nlv = (int)p;                                   /* :116 */
    double *d = wnorm;
    preamble(X, ldx, Y, ldy, w, n, p, q, scal, d, xmeans, xscales, ymeans, yscales);
    double *K = (double *)malloc(sizeof(double) * (size_t)p * (size_t)q);
    double *work = (double *)malloc(sizeof(double) * (size_t)p * (size_t)q);
    double *dt = (double *)malloc(sizeof(double) * (size_t)n);
    double *wv = (double *)malloc(sizeof(double) * (size_t)p);
    double *r = (double *)malloc(sizeof(double) * (size_t)p);
    double *zp = (double *)malloc(sizeof(double) * (size_t)p);
    xtdy(X, ldx, Y, ldy, d, n, p, q, K);                         /* :131-132 */
    for (int a = 0; a < nlv; ++a) {                              /* :149-175 */
        dominant_left_sv(K, p, q, wv, work);                     /* :150-155 */
        memcpy(r, wv, sizeof(double) * (size_t)p);
        for (int j = 0; j < a; ++j) {                            /* :156-161 */
            double s = 0.0;
            for (int64_t i = 0; i < p; ++i) s += wv[i] * P[i + (size_t)j * p];
            for (int64_t i = 0; i < p; ++i) r[i] -= s * R[i + (size_t)j * p];
        }
        double *t = T + (size_t)a * (size_t)n;
        xv(X, ldx, n, p, r, t);                                  /* :162 */
        double tt = 0.0;
        for (int64_t i = 0; i < n; ++i) { dt[i] = d[i] * t[i]; tt += t[i] * dt[i]; }   /* :163-164 */
        for (int64_t k = 0; k < q; ++k) {                        /* :165-166 */
            double s = 0.0;
            for (int64_t i = 0; i < p; ++i) s += K[i + k * p] * r[i];
            C[k + (size_t)a * q] = s / tt;
        }
        xtu(X, ldx, n, p, dt, zp);                               /* :167 */
        for (int64_t k = 0; k < q; ++k)                          /* :168 */
            for (int64_t i = 0; i < p; ++i) K[i + k * p] -= zp[i] * C[k + (size_t)a * q];
        for (int64_t i = 0; i < p; ++i) {                        /* :169-173 */
            P[i + (size_t)a * p] = zp[i] / tt;
            W[i + (size_t)a * p] = wv[i];
            R[i + (size_t)a * p] = r[i];
        }
        TT[a] = tt;                                              /* :174 */
    }
    free(K); free(work); free(dt); free(wv); free(r); free(zp);
    return nlv;
}

/* small dense helpers for R = W inv(P'W)  (src/plsnipals.jl:95) */
static int invert(double *A, int m, double *Ainv)              /* Gauss-Jordan, partial pivoting */
{
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Ainv[i + j * m] = (i == j);
    for (int c = 0; c < m; ++c) {
        int piv = c; double best = fabs(A[c + c * m]);
        for (int i = c + 1; i < m; ++i) if (fabs(A[i + c * m]) > best) { best = fabs(A[i + c * m]); piv = i; }
        if (piv != c)
            for (int j = 0; j < m; ++j) {
                double tmp = A[c + j * m]; A[c + j * m] = A[piv + j * m]; A[piv + j * m] = tmp;
                tmp = Ainv[c + j * m]; Ainv[c + j * m] = Ainv[piv + j * m]; Ainv[piv + j * m] = tmp;
            }
        double dd = A[c + c * m];
        for (int j = 0; j < m; ++j) { A[c + j * m] /= dd; Ainv[c + j * m] /= dd; }
        for (int i = 0; i < m; ++i) {
            if (i == c) continue;
            double f = A[i + c * m];
            if (f == 0.0) continue;
            for (int j = 0; j < m; ++j) { A[i + j * m] -= f * A[c + j * m]; Ainv[i + j * m] -= f * Ainv[c + j * m]; }
        }
    }
    return 0;
}

/* `plsnipals!` — src/plsnipals.jl:37-97.  X, Y end centred and deflated. */
int orc_plsnipals(double *X, int64_t ldx, double *Y, int64_t ldy, const double *w, int64_t n, int64_t p,
                  int64_t q, int nlv, int scal, double *T, double *P, double *R, double *W, double *C,
                  double *TT, double *xmeans, double *xscales, double *ymeans, double *yscales, double *wnorm)
{
    if (nlv > n) nlv = (int)n;
    if (nlv > p) nlv = (int)p;                                   /* :41 */
    double *d = wnorm;
    preamble(X, ldx, Y, ldy, w, n, p, q, scal, d, xmeans, xscales, ymeans, yscales);
    double *K = (double *)malloc(sizeof(double) * (size_t)p * (size_t)q);
    double *work = (double *)malloc(sizeof(double) * (size_t)p * (size_t)q);
    double *dt = (double *)malloc(sizeof(double) * (size_t)n);
    double *wv = (double *)malloc(sizeof(double) * (size_t)p);
    double *zp = (double *)malloc(sizeof(double) * (size_t)p);
    double *c = (double *)malloc(sizeof(double) * (size_t)q);
    for (int a = 0; a < nlv; ++a) {                              /* :70-94 */
        xtdy(X, ldx, Y, ldy, d, n, p, q, K);                     /* :71 */
        dominant_left_sv(K, p, q, wv, work);                     /* :72-77 */
        double *t = T + (size_t)a * (size_t)n;
        xv(X, ldx, n, p, wv, t);                                 /* :78 */
        double tt = 0.0;
        for (int64_t i = 0; i < n; ++i) { dt[i] = d[i] * t[i]; tt += t[i] * dt[i]; }   /* :79-80 */
        xtu(X, ldx, n, p, dt, zp);                               /* :81 */
        for (int64_t i = 0; i < p; ++i) zp[i] /= tt;             /* :82 */
        xtu(Y, ldy, n, q, dt, c);                                /* :83 */
        for (int64_t k = 0; k < q; ++k) c[k] /= tt;              /* :84 */
#pragma omp parallel for schedule(static)
        for (int64_t j = 0; j < p + q; ++j) {                    /* :86-87 */
            double *col = j < p ? &XC(0, j) : &YC(0, j - p);
            double f = j < p ? zp[j] : c[j - p];
            for (int64_t i = 0; i < n; ++i) col[i] -= t[i] * f;
        }
        for (int64_t i = 0; i < p; ++i) { P[i + (size_t)a * p] = zp[i]; W[i + (size_t)a * p] = wv[i]; }
        for (int64_t k = 0; k < q; ++k) C[k + (size_t)a * q] = c[k];
        TT[a] = tt;
    }
    /* R = W * inv(P' * W)   :95 */
    double *M = (double *)malloc(sizeof(double) * (size_t)nlv * nlv);
    double *Mi = (double *)malloc(sizeof(double) * (size_t)nlv * nlv);
    for (int i = 0; i < nlv; ++i)
        for (int j = 0; j < nlv; ++j) {
            double s = 0.0;
            for (int64_t k = 0; k < p; ++k) s += P[k + (size_t)i * p] * W[k + (size_t)j * p];
            M[i + j * nlv] = s;
        }
    invert(M, nlv, Mi);
    for (int j = 0; j < nlv; ++j)
        for (int64_t k = 0; k < p; ++k) {
            double s = 0.0;
            for (int i = 0; i < nlv; ++i) s += W[k + (size_t)i * p] * Mi[i + j * nlv];
            R[k + (size_t)j * p] = s;
        }
    free(M); free(Mi); free(K); free(work); free(dt); free(wv); free(zp); free(c);
    return nlv;
}
