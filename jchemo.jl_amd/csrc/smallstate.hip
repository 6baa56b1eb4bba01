// K3/K5 — the replicated small state of a fit, one workgroup, everything p x q or smaller, all fp64.
//
//   phase A (finish LV a)   src/plskern.jl:165-166,168-174: c = K'r/tt ; K -= zp c' ; P_a, W_a, R_a, C_a, TT_a
//                           src/plsnipals.jl:82,84,89-93  : zp/tt, c/tt and the column stores
//   phase B (prepare a+1)   src/plskern.jl:150-161 / src/plsnipals.jl:72-77: w = dominant left singular
//                           vector of K (q == 1: K/||K||), r = w - sum_j (w.P_j) R_j
// The reference calls LAPACK dgesdd on the p x q matrix; here: Gram G = K'K (q x q), parallel cyclic
// Jacobi eigen-decomposition in LDS, v = dominant eigenvector, w = K v / ||K v||  (matches dgesdd's U[:,1]
// to <= 3e-13).  Sign rule (the reference's sign is LAPACK's, F3): largest-|.| entry of v > 0.
// Every rank of a multi-GPU fit runs this kernel on identical all-reduced inputs -> bit-identical state.
#include "jch_internal.h"

#define NT 256

// out[k] (k < ncols) = sum_j A[j*lda + k] * x[j], j < rows; column tiles of 64.  scratch >= 256 doubles.
__device__ static void block_matTvec(const double *__restrict__ A, int lda, int rows, int ncols,
                                     const double *__restrict__ x, double *out_lds, double *scratch)
{
    for (int c0 = 0; c0 < ncols; c0 += 64) {
        const int nc = min(64, ncols - c0);
        const int cw = nc <= 16 ? 16 : (nc <= 32 ? 32 : 64);
        const int groups = NT / cw;
        const int k = threadIdx.x % cw, g = threadIdx.x / cw;
        double s = 0.0;
        if (k < nc)
            for (int j = g; j < rows; j += groups) s += A[(size_t)j * lda + c0 + k] * x[j];
        __syncthreads();
        scratch[g * cw + k] = s;
        __syncthreads();
        if (threadIdx.x < nc) {
            double t = 0.0;
            for (int gg = 0; gg < groups; ++gg) t += scratch[gg * cw + threadIdx.x];
            out_lds[c0 + threadIdx.x] = t;
        }
        __syncthreads();
    }
}

// Parallel cyclic Jacobi on the symmetric q x q matrix in A0 (LDS, ld = lda).  On return the eigenvalues are
// on the diagonal of the returned buffer and V holds the eigenvectors (columns).  Two syncs per round.
__device__ static void jacobi_eig(int q, int lda, double *&A0, double *&A1, double *&V0, double *&V1, double *cs,
                                  int *partner, int *flag)
{
    const int m = (q + 1) & ~1;  // even number of players; index q (if odd) is a bye
    const int tid = threadIdx.x;
    for (int e = tid; e < q * q; e += NT) {
        const int i = e / q, j = e % q;
        V0[i * lda + j] = (i == j) ? 1.0 : 0.0;
    }
    if (tid == 0) *flag = 0;
    __syncthreads();
    for (int sweep = 0; sweep < 40; ++sweep) {
        for (int round = 0; round < m - 1; ++round) {
            // ---- step 1: one thread per pair computes its rotation
            if (tid < m / 2) {
                int a, b;
                if (tid == 0) { a = m - 1; b = round; }
                else { a = (round + tid) % (m - 1); b = (round - tid + (m - 1)) % (m - 1); }
                if (a > b) { const int t = a; a = b; b = t; }
                double c = 1.0, s = 0.0;
                if (b < q) {
                    const double app = A0[a * lda + a], aqq = A0[b * lda + b], apq = A0[a * lda + b];
                    if (fabs(apq) > 1e-290 && fabs(apq) > 1e-17 * sqrt(fabs(app * aqq))) {
                        const double theta = (aqq - app) / (2.0 * apq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(t * t + 1.0);
                        s = t * c;
                        *flag = 1;  // benign race: any rotation this sweep sets it
                    }
                    partner[a] = b; partner[b] = a;
                    // role: +1 for the lower index (p), -1 for the higher (q)
                    cs[2 * a] = c; cs[2 * a + 1] = s;
                    cs[2 * b] = c; cs[2 * b + 1] = -s;
                } else if (a < q) {  // bye
                    partner[a] = a;
                    cs[2 * a] = 1.0; cs[2 * a + 1] = 0.0;
                }
            }
            __syncthreads();
            // ---- step 2: A1 = J' A0 J, V1 = V0 J   (for index i with partner i': Jcol_i = c e_i + s_i e_i',
            //      with s_i = +s for the lower index (new_p = c*x_p - s*x_q) -> encoded so that
            //      new_i = c * x_i - sgn * s * x_partner, sgn carried in cs[2i+1])
            for (int e = tid; e < q * q; e += NT) {
                const int i = e / q, j = e % q;
                const int ip = partner[i], jp = partner[j];
                const double ci = cs[2 * i], si = cs[2 * i + 1], cj = cs[2 * j], sj = cs[2 * j + 1];
                // row op on rows (i, ip) evaluated at columns j and jp
                const double rij = ci * A0[i * lda + j] - si * A0[ip * lda + j];
                const double rijp = ci * A0[i * lda + jp] - si * A0[ip * lda + jp];
                A1[i * lda + j] = cj * rij - sj * rijp;
                V1[i * lda + j] = cj * V0[i * lda + j] - sj * V0[i * lda + jp];
            }
            __syncthreads();
            double *t = A0; A0 = A1; A1 = t;
            t = V0; V0 = V1; V1 = t;
        }
        const int any = *flag;
        __syncthreads();
        if (tid == 0) *flag = 0;
        __syncthreads();
        if (!any) break;
    }
}

struct lv_args {
    jch_small s;
    int p, q, qpad, ldr, a, nlv, algo, do_a, do_b;   // algo: 0 plskern, 1 plsnipals, 2 plssimp, 4 plswold (phase B only)
    int maxit;      // plswold
    double tol;
    double *ws;     // global workspace (round 4: no limit on q or nlv): dots [nlv], then for q > 64 the eigen-solver's matrices
                    // A0, A1, V0, V1 [4 q (q + 1)], cs [2 (q + 2)], partner / flag (ints), and plswold's four q-vectors
};

__global__ __launch_bounds__(NT) void k_lv_update(lv_args g)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int p = g.p, q = g.q, qpad = g.qpad, ldr = g.ldr, a = g.a, tid = threadIdx.x;
    const int lda = q + 1;
    const int qv = (q + 63) & ~63;
    double *scratch = lds;                 // [256]
    double *vec = scratch + 256;           // [qv]   c / v / dots
    double *dots = g.ws;                   // [nlv]  global (any nlv)
    // the eigen-solver's q x q matrices: LDS up to q = 64, the global workspace beyond (same code, generic pointers)
    const bool eig_lds = q <= JCH_MAXQ;
    double *A0 = eig_lds ? vec + qv : g.ws + ((g.nlv + 7) & ~7);
    double *A1 = A0 + q * lda, *V0 = A1 + q * lda, *V1 = V0 + q * lda;
    double *cs = V1 + q * lda;             // [2*(q+2)]
    int *partner = reinterpret_cast<int *>(cs + 2 * (q + 2));  // [q+2]
    int *flag = partner + (q + 2);
    double *wvec = reinterpret_cast<double *>(partner + ((q + 2 + 2 + 1) & ~1));   // plswold: b, a_prev, G b, delta  [4 q]
    double *K = g.s.K;
    const double *zt = g.s.zt;

    // ------------------------------------------------------------------ phase A
    if (g.do_a) {
        const double tt = zt[ldr];
        if (g.algo == 2) {
            // SIMPLS (src/plssimp.jl:64-83; state as in siblings.hip: K = XtY projected on the complement of the loadings,
            // V = s.W an orthonormal basis of their span).  s.w serves as the work vector (phase B rewrites it).
            block_matTvec(K, qpad, p, q, g.s.r, vec, scratch);  // vec[k] = (K' r)_k  (== XtY' r: r is orthogonal to V)
            if (tid < q) g.s.C[(size_t)a * q + tid] = vec[tid] / tt;
            double *tv = g.s.w, *V = g.s.W;
            for (int j = tid; j < p; j += NT) {
                const double pj = zt[j] / tt;
                g.s.P[(size_t)a * p + j] = pj;
                g.s.R[(size_t)a * p + j] = g.s.r[j];
                tv[j] = pj;
            }
            __threadfence_block();
            __syncthreads();
            for (int pass = 0; pass < 2; ++pass) {   // v = P_a - V (V'P_a), twice
                const int lane = tid & 63, wv = tid >> 6;
                for (int i = wv; i < a; i += NT / 64) {
                    double s3 = 0.0;
                    for (int j = lane; j < p; j += 64) s3 += V[(size_t)i * p + j] * tv[j];
                    s3 = jch_wave_sum(s3);
                    if (lane == 0) dots[i] = s3;
                }
                __syncthreads();
                for (int j = tid; j < p; j += NT) {
                    double acc = 0.0;
                    for (int i = 0; i < a; ++i) acc += dots[i] * V[(size_t)i * p + j];
                    tv[j] -= acc;
                }
                __threadfence_block();
                __syncthreads();
            }
            double ss = 0.0;
            for (int j = tid; j < p; j += NT) ss += tv[j] * tv[j];
            const double inv = 1.0 / sqrt(jch_block_sum<NT>(ss, scratch));
            for (int j = tid; j < p; j += NT) V[(size_t)a * p + j] = tv[j] * inv;
            __threadfence_block();
            __syncthreads();
            block_matTvec(K, qpad, p, q, V + (size_t)a * p, vec, scratch);   // vec = K'v
            for (int e = tid; e < p * q; e += NT) {
                const int j = e / q, k = e % q;
                K[(size_t)j * qpad + k] -= V[(size_t)a * p + j] * vec[k];
            }
        } else if (g.algo == 0) {
            block_matTvec(K, qpad, p, q, g.s.r, vec, scratch);  // vec[k] = (K' r)_k
            if (tid < q) {
                vec[tid] = vec[tid] / tt;
                g.s.C[(size_t)a * q + tid] = vec[tid];
            }
            __syncthreads();
            for (int e = tid; e < p * q; e += NT) {
                const int j = e / q, k = e % q;
                K[(size_t)j * qpad + k] -= zt[j] * vec[k];
            }
            for (int j = tid; j < p; j += NT) {
                g.s.P[(size_t)a * p + j] = zt[j] / tt;
                g.s.W[(size_t)a * p + j] = g.s.w[j];
                g.s.R[(size_t)a * p + j] = g.s.r[j];
            }
        } else {
            for (int j = tid; j < ldr; j += NT) {
                const double z = j < p ? zt[j] / tt : 0.0;
                g.s.zpc[j] = z;
                if (j < p) {
                    g.s.P[(size_t)a * p + j] = z;
                    g.s.W[(size_t)a * p + j] = g.s.w[j];
                }
            }
            for (int k = tid; k < qpad; k += NT) {
                const double c = k < q ? zt[ldr + 1 + k] / tt : 0.0;
                g.s.zpc[ldr + k] = c;
                if (k < q) g.s.C[(size_t)a * q + k] = c;
            }
            if (g.s.variant == 3)     // OPT-IN one-pass NIPALS (JCH_NIPALS_ONE_PASS): K_{a+1} = K_a - zp_raw c_raw' / tt
                for (int e = tid; e < p * q; e += NT) {
                    const int j = e / q, k = e % q;
                    K[(size_t)j * qpad + k] -= zt[j] * (zt[ldr + 1 + k] / tt);
                }
        }
        if (tid == 0) g.s.TT[a] = tt;
        __syncthreads();
        __threadfence_block();
    }
    if (!g.do_b) return;

    // ------------------------------------------------------------------ phase B: next w, r
    const int anext = g.do_a ? a + 1 : a;  // number of finished LVs (columns of P/R valid)
    double ssq = 0.0;
    if (q == 1) {
        if (g.algo == 4 && tid == 0 && g.s.niter) g.s.niter[a] = g.maxit >= 2 ? (0.0 < g.tol ? 2.0 : (double)g.maxit) : 1.0;
        for (int j = tid; j < p; j += NT) {
            const double v = K[(size_t)j * qpad];
            g.s.w[j] = v;
            ssq += v * v;
        }
    } else {
        // Gram G = K'K (q x q): the nent = q(q+1)/2 upper entries, rows of K split over NT/EW groups
        const int nent = q * (q + 1) / 2;
        const int EW = nent <= 64 ? 64 : (nent <= 128 ? 128 : 256);
        const int G = NT / EW, el = tid % EW, gr = tid / EW;
        for (int e0 = 0; e0 < nent; e0 += EW) {
            int e = e0 + el, k1 = 0;
            double s = 0.0;
            const bool act = e < nent;
            if (act) {
                while (e >= q - k1) { e -= q - k1; ++k1; }
                const int k2 = k1 + e;
                for (int j = gr; j < p; j += G) s += K[(size_t)j * qpad + k1] * K[(size_t)j * qpad + k2];
            }
            __syncthreads();
            scratch[gr * EW + el] = s;
            __syncthreads();
            if (act && gr == 0) {
                double t = 0.0;
                for (int gg = 0; gg < G; ++gg) t += scratch[gg * EW + el];
                const int k2 = k1 + e;
                A0[k1 * lda + k2] = t;
                A0[k2 * lda + k1] = t;
            }
        }
        __syncthreads();
        if (g.algo == 4) {
            // plswold: the inner loop of src/plswold.jl:79-92 as a power iteration on G = K'K (see k_wold_b, siblings.hip);
            // one wave, lane l owns the components l, l + 64, ... (q <= 64: one each — the arithmetic of the earlier one-per-lane form)
            if (tid < 64) {
                const int lane = tid;
                double *vb = wvec, *va = wvec + q, *vg = va + q, *vd = vg + q;
                auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
                for (int c = lane; c < q; c += 64) { vb[c] = c == 0 ? 1.0 : 0.0; va[c] = 0.0; }
                int k = 1;
                for (;;) {
                    wsync();
                    double pn2 = 0.0, pg2 = 0.0;
                    for (int c = lane; c < q; c += 64) {
                        double gb = 0.0;
                        for (int kk = 0; kk < q; ++kk) gb += A0[c * lda + kk] * vb[kk];
                        vg[c] = gb;
                        pn2 += vb[c] * gb; pg2 += gb * gb;
                    }
                    const double n2 = jch_wave_sum(pn2), g2 = jch_wave_sum(pg2);
                    bool stop = k >= g.maxit;
                    if (k >= 2) {
                        for (int c = lane; c < q; c += 64) vd[c] = vb[c] / sqrt(n2) - va[c];
                        wsync();
                        double pd = 0.0;
                        for (int c = lane; c < q; c += 64) {
                            double gd = 0.0;
                            for (int kk = 0; kk < q; ++kk) gd += A0[c * lda + kk] * vd[kk];
                            pd += vd[c] * gd;
                        }
                        const double dif = jch_wave_sum(pd);
                        if (dif < g.tol) stop = true;
                    }
                    if (stop) {
                        for (int c = lane; c < q; c += 64) vec[c] = vb[c] / sqrt(n2);
                        break;
                    }
                    wsync();
                    for (int c = lane; c < q; c += 64) { va[c] = vb[c] / sqrt(n2); vb[c] = vg[c] / sqrt(g2); }
                    ++k;
                }
                if (lane == 0 && g.s.niter) g.s.niter[a] = (double)k;
            }
            __syncthreads();
        } else {
        jacobi_eig(q, lda, A0, A1, V0, V1, cs, partner, flag);
        if (tid == 0) {
            int best = 0;
            for (int k = 1; k < q; ++k)
                if (A0[k * lda + k] > A0[best * lda + best]) best = k;
            double big = 0.0;
            for (int k = 0; k < q; ++k)
                if (fabs(V0[k * lda + best]) > fabs(big)) big = V0[k * lda + best];
            const double sg = big < 0.0 ? -1.0 : 1.0;
            for (int k = 0; k < q; ++k) vec[k] = sg * V0[k * lda + best];
        }
        __syncthreads();
        }
        for (int j = tid; j < p; j += NT) {
            double wv = 0.0;
            for (int k = 0; k < q; ++k) wv += K[(size_t)j * qpad + k] * vec[k];
            g.s.w[j] = wv;
            ssq += wv * wv;
        }
    }
    const double nrm = sqrt(jch_block_sum<NT>(ssq, scratch));
    const bool plain = (g.algo != 0 || anext == 0);
    for (int j = tid; j < ldr; j += NT) {  // each thread re-reads only its own stores
        const double wv = j < p ? g.s.w[j] / nrm : 0.0;
        g.s.w[j] = wv;
        if (plain) g.s.r[j] = wv;
    }
    if (plain) return;
    // dots[i] = w . P_i   (one wave per finished LV, lanes stride the vector)
    __syncthreads();
    {
        const int lane = tid & 63, wv = tid >> 6;
        for (int i = wv; i < anext; i += NT / 64) {
            const double *Pi = g.s.P + (size_t)i * p;
            double s = 0.0;
            for (int j = lane; j < p; j += 64) s += g.s.w[j] * Pi[j];
            s = jch_wave_sum(s);
            if (lane == 0) dots[i] = s;
        }
    }
    __syncthreads();
    for (int j = tid; j < ldr; j += NT) {
        double rj = 0.0;
        if (j < p) {
            rj = g.s.w[j];
            for (int l = 0; l < anext; ++l) rj -= dots[l] * g.s.R[(size_t)l * p + j];
        }
        g.s.r[j] = rj;
    }
}

int32_t jch_launch_lv_update(jch_ctx *ctx, const jch_small &s, int p, int q, int qpad, int ldr, int a, int nlv, int algo,
                             int nslice, int ldz, bool fast, bool fuse_p2p, const double *bf_src, int bf_ld, int bf_ldr, double tol, int maxit)
{
    // a encodes the phase:  a == -1           -> phase B only (first w, r)
    //                       a >= 0, a < nlv   -> phase A for LV a, then phase B unless it was the last LV
    // plsnipals splits A and B around the deflation pass: a |= 0x40000000 -> phase A only,
    //                                                      a |= 0x20000000 -> phase B only with `a` LVs finished.
    lv_args g;
    g.s = s; g.p = p; g.q = q; g.qpad = qpad; g.ldr = ldr; g.nlv = nlv; g.algo = algo; g.tol = tol; g.maxit = maxit < 1 ? 1 : maxit;
    const int flags = a < 0 ? 0 : (a & 0x60000000);
    const int aa = a < 0 ? -1 : (a & 0x1fffffff);
    if (aa < 0) { g.a = 0; g.do_a = 0; g.do_b = 1; }
    else if (flags & 0x40000000) { g.a = aa; g.do_a = 1; g.do_b = 0; }
    else if (flags & 0x20000000) { g.a = aa; g.do_a = 0; g.do_b = 1; }
    else { g.a = aa; g.do_a = 1; g.do_b = (aa + 1 < nlv) ? 1 : 0; }
    if (fast && algo >= 2) return jch_fail(ctx, JCH_EINVAL, "internal: plssimp / plswold use their own fast kernels (siblings.hip)");
    if (fast) return jch_launch_lv_update_fast(ctx, s, p, q, qpad, ldr, g.a, nlv, algo, g.do_a, g.do_b, nslice, ldz, fuse_p2p, bf_src, bf_ld, bf_ldr);
    if (fuse_p2p || bf_src) return jch_fail(ctx, JCH_EINVAL, "internal: the fused inbox all-reduce / bf16 fix-up need the fast small-state kernel");
    if (nslice != 1) return jch_fail(ctx, JCH_EINVAL, "internal: generic small-state kernel needs a single zt slice");
    const int lda = q + 1, qv = (q + 63) & ~63;
    const size_t eig = sizeof(double) * (4 * (size_t)q * lda + 2 * (q + 2)) + sizeof(int) * ((q + 2 + 2 + 1) & ~1) + sizeof(double) * 4 * (size_t)q;
    const size_t lds = sizeof(double) * (256 + qv) + (q <= JCH_MAXQ ? eig : 0);
    JCH_TRY(jch_reserve(ctx, ctx->lvws, sizeof(double) * ((size_t)((nlv + 7) & ~7)) + eig + 256));
    g.ws = (double *)ctx->lvws.ptr;
    static jch_per_device_once attr_once;
    if (!attr_once.done(ctx->device)) {
        JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_lv_update, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.mark(ctx->device);
    }
    hipLaunchKernelGGL(k_lv_update, dim3(1), dim3(NT), lds, ctx->stream, g);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

// R = W inv(P'W)   (src/plsnipals.jl:95) — once per plsnipals fit, three small launches:
//   k_nipals_M    M = P'W (nlv x nlv), one wave per entry, many blocks
//   k_nipals_inv  Gauss-Jordan with partial pivoting, single workgroup (matrices in global scratch)
//   k_nipals_Rmul R = W * Mi, one thread per output
struct nipR_args {
    const double *P, *W;
    double *R, *M, *Mi;
    int p, nlv;
};
__global__ __launch_bounds__(NT) void k_nipals_M(nipR_args g)
{
    const int m = g.nlv, p = g.p, lane = threadIdx.x & 63;
    const int e = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (e >= m * m) return;
    const int i = e / m, j = e % m;
    double s0 = 0.0, s1 = 0.0;
    int k = lane;
    for (; k + 64 < p; k += 128) {
        s0 += g.P[(size_t)i * p + k] * g.W[(size_t)j * p + k];
        s1 += g.P[(size_t)i * p + k + 64] * g.W[(size_t)j * p + k + 64];
    }
    for (; k < p; k += 64) s0 += g.P[(size_t)i * p + k] * g.W[(size_t)j * p + k];
    const double s = jch_wave_sum(s0 + s1);
    if (lane == 0) { g.M[e] = s; g.Mi[e] = (i == j) ? 1.0 : 0.0; }
}
// LDS = true (2 m^2 doubles fit: m <= 90): the whole elimination runs on LDS copies — barriers only; the global-memory version
// orders its phases with block-scope fences (round 1 used device-scope ones: each wrote back / invalidated the L2).
template <bool LDS>
__global__ __launch_bounds__(NT) void k_nipals_inv(nipR_args g)
{
    extern __shared__ __attribute__((aligned(16))) double inv_lds[];
    __shared__ int piv_s;
    const int m = g.nlv, tid = threadIdx.x;
    double *M = LDS ? inv_lds : g.M, *Mi = LDS ? inv_lds + m * m : g.Mi;
    if (LDS) {
        for (int e = tid; e < m * m; e += NT) { M[e] = g.M[e]; Mi[e] = g.Mi[e]; }
        __syncthreads();
    }
    const int jl = tid % 64, il = tid / 64;             // (compile-time divisors) column lane / row group of the update
    for (int c = 0; c < m; ++c) {
        if (tid == 0) {
            int piv = c; double best = fabs(M[c * m + c]);
            for (int i = c + 1; i < m; ++i)
                if (fabs(M[i * m + c]) > best) { best = fabs(M[i * m + c]); piv = i; }
            piv_s = piv;
        }
        __syncthreads();
        const int piv = piv_s;
        if (piv != c)
            for (int j = tid; j < m; j += NT) {
                double t = M[c * m + j]; M[c * m + j] = M[piv * m + j]; M[piv * m + j] = t;
                t = Mi[c * m + j]; Mi[c * m + j] = Mi[piv * m + j]; Mi[piv * m + j] = t;
            }
        if (!LDS) __threadfence_block();
        __syncthreads();
        const double dd = M[c * m + c];
        __syncthreads();
        for (int j = tid; j < m; j += NT) { M[c * m + j] /= dd; Mi[c * m + j] /= dd; }
        if (!LDS) __threadfence_block();
        __syncthreads();
        for (int i = il; i < m; i += NT / 64) {
            if (i == c) continue;
            const double f = M[i * m + c];
            if (f == 0.0) continue;
            // column c of M (the multipliers f) is left untouched in this step and zeroed in the next
            for (int j = jl; j < m; j += 64) {
                if (j != c) M[i * m + j] -= f * M[c * m + j];
                Mi[i * m + j] -= f * Mi[c * m + j];
            }
        }
        if (!LDS) __threadfence_block();
        __syncthreads();
        for (int i = tid; i < m; i += NT)
            if (i != c) M[i * m + c] = 0.0;
        if (!LDS) __threadfence_block();
        __syncthreads();
    }
    if (LDS)
        for (int e = tid; e < m * m; e += NT) g.Mi[e] = Mi[e];
}
__global__ __launch_bounds__(NT) void k_nipals_Rmul(nipR_args g)
{
    const int m = g.nlv, p = g.p;
    const int e = blockIdx.x * NT + threadIdx.x;     // R[j][k] = sum_i W[i][k] * Mi[i][j]   (stored [lv][p])
    if (e >= m * p) return;
    const int j = e / p, k = e - j * p;
    double s = 0.0;
    for (int i = 0; i < m; ++i) s += g.W[(size_t)i * p + k] * g.Mi[i * m + j];
    g.R[(size_t)j * p + k] = s;
}

int32_t jch_launch_nipals_R(jch_ctx *ctx, const jch_small &s, int p, int nlv)
{
    JCH_TRY(jch_reserve(ctx, ctx->gemm_b, sizeof(double) * 2 * (size_t)nlv * nlv));
    nipR_args g;
    g.P = s.P; g.W = s.W; g.R = s.R; g.p = p; g.nlv = nlv;
    g.M = (double *)ctx->gemm_b.ptr;
    g.Mi = g.M + (size_t)nlv * nlv;
    hipLaunchKernelGGL(k_nipals_M, dim3((nlv * nlv + NT / 64 - 1) / (NT / 64)), dim3(NT), 0, ctx->stream, g);
    const size_t inv_lds = sizeof(double) * 2 * (size_t)nlv * nlv;
    if (inv_lds <= 140 * 1024) {
        static jch_per_device_once attr;
        if (!attr.done(ctx->device)) {
            JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_nipals_inv<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));   // (+ the static pivot word)
            attr.mark(ctx->device);
        }
        hipLaunchKernelGGL(k_nipals_inv<true>, dim3(1), dim3(NT), inv_lds, ctx->stream, g);
    } else {
        hipLaunchKernelGGL(k_nipals_inv<false>, dim3(1), dim3(NT), 0, ctx->stream, g);
    }
    hipLaunchKernelGGL(k_nipals_Rmul, dim3((nlv * p + NT - 1) / NT), dim3(NT), 0, ctx->stream, g);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}
