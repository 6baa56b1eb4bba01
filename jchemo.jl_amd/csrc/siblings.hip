// Small-state kernels of the sibling PLS algorithms.  They all reuse the row kernels of the hot
// path (fused sweep, deflation) and differ only in the replicated p x q state handled between two sweeps:
//   plssimp  (src/plssimp.jl:28-88)  k_lv_update_simp : XtY projected on the complement of the loadings, r = w
//   plsrosa  (src/plsrosa.jl:32-96)  k_rosa_orthw     : W re-orthonormalised (:77-79) after a plskern-shaped fit
//                                    k_ydeflate_all   : `plsrosa!` hands back Y - T C' (:87)
//   plswold  (src/plswold.jl:36-111) k_wold_b         : inner power iteration (:79-92) on the q x q Gram matrix
// One workgroup each, K in LDS, fp64; q <= 16 and p x q inside LDS (the envelope of the fast small-state path).
#include <stdlib.h>

#include "jch_internal.h"
#include "lv_device.h"

// K (global, [p][16]) -> LDS [p][ldk]; columns >= QP dropped (they are zero)
template <int QP>
__device__ __forceinline__ void load_k_lds(const double *__restrict__ K, double *Kl, int p, int tid)
{
    constexpr int ldk = QP | 1;
    const int tot = p * 16;
    for (int base = 0; base < tot; base += FT * 16) {
        double kr[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) kr[i] = K[min(base + tid + FT * i, tot - 1)];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = base + tid + FT * i;
            if (e < tot && (e & 15) < QP) Kl[(e >> 4) * ldk + (e & 15)] = kr[i];
        }
    }
}

// G0 (QP x QP, ld lda, zero-padded, pre-zeroed) = K'K from the LDS copy of K.  All FT threads; ends with a barrier.
template <int QP>
__device__ __forceinline__ void gram_lds(const double *Kl, int p, int q, double *G0, double *scratch)
{
    constexpr int ldk = QP | 1, lda = QP + 2;
    const int tid = threadIdx.x;
    const int nent = q * (q + 1) / 2;
    const int el = tid & 63, gr = tid >> 6;
    for (int e0 = 0; e0 < nent; e0 += 64) {
        int e = e0 + el, k1 = 0;
        double s = 0.0;
        const bool act = e < nent;
        if (act) {
            while (e >= q - k1) { e -= q - k1; ++k1; }
            const int k2 = k1 + e;
            for (int j = gr; j < p; j += 8 * (FT / 64)) {
                double x1[8], x2[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int jj = min(j + u * (FT / 64), p - 1);
                    x1[u] = Kl[jj * ldk + k1];
                    x2[u] = Kl[jj * ldk + k2];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) s += (j + u * (FT / 64) < p) ? x1[u] * x2[u] : 0.0;
            }
        }
        __syncthreads();
        scratch[gr * 64 + el] = s;
        __syncthreads();
        if (act && gr == 0) {
            double t = 0.0;
#pragma unroll
            for (int gg = 0; gg < FT / 64; ++gg) t += scratch[gg * 64 + el];
            const int k2 = k1 + e;
            G0[k1 * lda + k2] = t;
            G0[k2 * lda + k1] = t;
        }
    }
    __syncthreads();
}

// w = K v / ||K v|| for the QP-vector v in LDS; written to both s.w and s.r (no r-recursion in these algorithms)
template <int QP>
__device__ __forceinline__ void write_w_from_v(const double *Kl, const double *vl, int p, int ldr, double *scratch,
                                               double *w_out, double *r_out, double *rs_out = nullptr, const double *scl = nullptr)
{
    constexpr int ldk = QP | 1;
    const int tid = threadIdx.x;
    double wr[JCH_SWEEP_MAXP / FT];
    double ssq = 0.0;
#pragma unroll
    for (int it = 0; it < JCH_SWEEP_MAXP / FT; ++it) {
        const int j = min(tid + it * FT, p - 1);
        double wv_ = 0.0;
#pragma unroll
        for (int k = 0; k < QP; ++k) wv_ += Kl[j * ldk + k] * vl[k];
        if (tid + it * FT >= p) wv_ = 0.0;
        wr[it] = wv_;
        ssq += wv_ * wv_;
    }
    const double inv = 1.0 / sqrt(jch_block_sum<FT>(ssq, scratch));
#pragma unroll
    for (int it = 0; it < JCH_SWEEP_MAXP / FT; ++it) {
        const int j = tid + it * FT;
        if (j < ldr) {
            const double wn = j < p ? wr[it] * inv : 0.0;
            w_out[j] = wn;
            r_out[j] = wn;
            if (rs_out) rs_out[j] = j < p ? wn / scl[j] : 0.0;
        }
    }
}

// ------------------------------------------------------------------------------------------- SIMPLS
// State: K = (I - V V') XtY with V an orthonormal basis of span(P[:, 0..a)) (the reference rebuilds the projector
// from P every LV, src/plssimp.jl:68-69; nested projectors give the same matrix, so K is deflated by ONE new
// direction per LV).  s.W holds V during the fit (SIMPLS has no W; the caller reports R in its place, :85-87).
//   phase A (after the sweep of LV a): c = K'r / tt (== XtY'r / tt because r is orthogonal to V, :75-76),
//            P_a = zp / tt, v = P_a - V (V'P_a) twice (re-orthogonalised Gram-Schmidt), K <- K - v (v'K)
//   phase B: r = dominant left singular vector of K (:71, also when q == 1)
template <int QP>
__global__ __launch_bounds__(FT) void k_lv_update_simp(lvf_args g)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int p = g.p, q = g.q, ldr = g.ldr, a = g.a, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    constexpr int ldk = QP | 1, lda = QP + 2;
    double *Kl = lds;                              // [p][ldk]
    double *ztl = Kl + (size_t)p * ldk;            // [ldr + 18]
    double *rl = ztl + (ldr + 18);                 // [ldr]
    double *pl = rl + ldr;                         // [ldr]   P_a, then v
    double *scratch = pl + ldr;                    // [2 FT]
    double *cl = scratch + 2 * FT;                 // [16]
    double *vl = cl + 16;                          // [16]
    double *ul = vl + 16;                          // [nlv]
    double *G0 = ul + ((g.nlv + 1) & ~1), *A0 = G0 + QP * lda, *A1 = A0 + QP * lda, *V0 = A1 + QP * lda, *V1 = V0 + QP * lda;
    double *csl = V1 + QP * lda;                   // [2 (QP + 2)]
    double *K = g.s.K, *V = g.s.W;
    load_k_lds<QP>(K, Kl, p, tid);
    for (int e = tid; e < 5 * QP * lda; e += FT) G0[e] = 0.0;
    if (tid < 32) cl[tid] = 0.0;
    if (g.do_a) {
        for (int c = tid; c < ldr + 1 + (g.raw_mu ? 1 : 0); c += FT) {
            double s = 0.0;
            if (g.nslice == 1) s = g.s.zt[c];
            else
                for (int sl = 0; sl < JCH_ZT_SLICES; ++sl) s += g.s.zt[(size_t)sl * g.ldz + c];
            ztl[c] = s;
        }
        for (int j = tid; j < ldr; j += FT) rl[j] = g.s.r[j];
    }
    __syncthreads();
    if (g.do_a && g.raw_mu) {   // raw mode (fit.hip): the sweep ran on rows minus the pivot; zp = zp_raw - (mu - pivot) * st
        const double st_ = ztl[ldr + 1];
        if (g.s.rs) { for (int j = tid; j < p; j += FT) ztl[j] = (ztl[j] - g.s.mshift[j] * st_) / g.s.scl[j]; }
        else for (int j = tid; j < p; j += FT) ztl[j] -= g.s.mshift[j] * st_;
        __syncthreads();
    }
    if (g.do_a) {
        const double tt = ztl[ldr];
        {
            const int k = tid & 15, gr = tid >> 4;
            scratch[gr * 16 + k] = k < QP ? kcol_dot(Kl, ldk, k, rl, gr, FT / 16, p) : 0.0;
        }
        for (int j = tid; j < ldr; j += FT) {
            const double z = j < p ? ztl[j] / tt : 0.0;
            pl[j] = z;
            if (j < p) {
                g.s.P[(size_t)a * p + j] = z;
                g.s.R[(size_t)a * p + j] = rl[j];
            }
        }
        __syncthreads();
        if (tid < 16) {
            double t = 0.0;
#pragma unroll
            for (int gg = 0; gg < FT / 16; ++gg) t += scratch[gg * 16 + tid];
            t = tid < q ? t / tt : 0.0;
            if (tid < q) g.s.C[(size_t)a * q + tid] = t;
        }
        if (tid == 0) g.s.TT[a] = tt;
        for (int pass = 0; pass < 2; ++pass) {   // v = P_a - V (V'P_a), twice
            for (int i = wv; i < a; i += FT / 64) {
                double s = 0.0;
                for (int j = lane; j < p; j += 64) s += V[(size_t)i * p + j] * pl[j];
                s = jch_wave_sum(s);
                if (lane == 0) ul[i] = s;
            }
            __syncthreads();
            for (int j = tid; j < p; j += FT) {
                double acc = 0.0;
                for (int i = 0; i < a; ++i) acc += ul[i] * V[(size_t)i * p + j];
                pl[j] -= acc;
            }
            __syncthreads();
        }
        double ss = 0.0;
        for (int j = tid; j < p; j += FT) ss += pl[j] * pl[j];
        const double inv = 1.0 / sqrt(jch_block_sum<FT>(ss, scratch));
        for (int j = tid; j < p; j += FT) {
            const double v = pl[j] * inv;
            pl[j] = v;
            V[(size_t)a * p + j] = v;
        }
        __syncthreads();
        {   // gk = v'K
            const int k = tid & 15, gr = tid >> 4;
            scratch[gr * 16 + k] = k < QP ? kcol_dot(Kl, ldk, k, pl, gr, FT / 16, p) : 0.0;
        }
        __syncthreads();
        if (tid < 16) {
            double t = 0.0;
#pragma unroll
            for (int gg = 0; gg < FT / 16; ++gg) t += scratch[gg * 16 + tid];
            cl[tid] = tid < q ? t : 0.0;
        }
        __syncthreads();
        for (int j = tid; j < p; j += FT) {   // K <- K - v (v'K)
            const double v = pl[j];
#pragma unroll
            for (int k = 0; k < QP; ++k) {
                const double kv = Kl[j * ldk + k] - v * cl[k];
                Kl[j * ldk + k] = kv;
                K[(size_t)j * 16 + k] = kv;
            }
        }
        __syncthreads();
    }
    if (!g.do_b) return;
    if (q > 1) {
        gram_lds<QP>(Kl, p, q, G0, scratch);
        if (wv == 0) {
            bool solved;
            if constexpr (QP == 16) solved = dominant_by_squaring_mfma16(q, lda, G0, vl, nullptr);
            else solved = dominant_by_squaring<QP>(q, lda, G0, A0, A1, vl, nullptr);
            if (!solved) {
                for (int e = lane; e < QP * lda; e += 64) A0[e] = G0[e];
                wavesync();
                jacobi_wave(q, lda, A0, A1, V0, V1, csl, vl, nullptr);
            }
        }
    } else if (tid == 0) {
        vl[0] = 1.0;
    }
    __syncthreads();
    write_w_from_v<QP>(Kl, vl, p, ldr, scratch, g.s.w, g.s.r, g.s.rs, g.s.scl);
}

// ------------------------------------------------------------------------------------------- Wold NIPALS
// Phase B of plswold: the inner loop of src/plswold.jl:79-92 only ever touches X and Y through K = X'DY
// (wx ~ X'ty = K wy, wy ~ Y'tx = K'wx), so it is a power iteration that can run on the q x q Gram matrix G = K'K:
//   b_1 = e_1 (ty = Y[:, 1], :75);   wx_k = K b_k / n_k,  n_k^2 = b_k'G b_k;   b_{k+1} = G b_k / ||G b_k||
//   dif_k = ||wx_k - wx_{k-1}||^2 = d'G d  with  d = b_k / n_k - b_{k-1} / n_{k-1}          (:87)
// stop after pass k when dif_k < tol (k >= 2: the first check compares with `rand(p)`, :78, and never passes) or
// k == maxit (:88-91).  One wave, lane j < QP owns component j.  Output: w = wx_k (sign as iterated, no sign rule),
// niter[a] = k (:93).
template <int QP>
__global__ __launch_bounds__(FT) void k_wold_b(lvf_args g)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int p = g.p, q = g.q, ldr = g.ldr, a = g.a, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    constexpr int ldk = QP | 1, lda = QP + 2;
    double *Kl = lds;                              // [p][ldk]
    double *scratch = Kl + (size_t)p * ldk;        // [2 FT]
    double *vl = scratch + 2 * FT;                 // [16]  b / n
    double *bl = vl + 16;                          // [16]
    double *G0 = bl + 16;                          // [QP][lda]
    load_k_lds<QP>(g.s.K, Kl, p, tid);
    for (int e = tid; e < QP * lda; e += FT) G0[e] = 0.0;
    if (tid < 32) vl[tid] = 0.0;
    __syncthreads();
    int iters = 1;
    if (q > 1) {
        gram_lds<QP>(Kl, p, q, G0, scratch);
        if (wv == 0) {
            const int j = lane < QP ? lane : 0;
            double grow[QP];
#pragma unroll
            for (int k = 0; k < QP; ++k) grow[k] = G0[j * lda + k];
            double b = (lane == 0) ? 1.0 : 0.0, aprev = 0.0;   // aprev: component j of b_{k-1} / n_{k-1}
            int k = 1;
            for (;;) {
                if (lane < QP) bl[lane] = b;
                wavesync();
                double gb = 0.0;
#pragma unroll
                for (int kk = 0; kk < QP; ++kk) gb += grow[kk] * bl[kk];       // (G b)_j
                wavesync();
                double n2 = (lane < q) ? b * gb : 0.0, g2 = (lane < q) ? gb * gb : 0.0;
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { n2 += __shfl_xor(n2, o, 16); g2 += __shfl_xor(g2, o, 16); }
                const double acur = b / sqrt(n2);
                bool stop = k >= g.maxit;
                if (k >= 2) {
                    const double d = acur - aprev;
                    if (lane < QP) bl[lane] = d;
                    wavesync();
                    double gd = 0.0;
#pragma unroll
                    for (int kk = 0; kk < QP; ++kk) gd += grow[kk] * bl[kk];
                    wavesync();
                    double dif = (lane < q) ? d * gd : 0.0;
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) dif += __shfl_xor(dif, o, 16);
                    dif = __shfl(dif, 0, 64);
                    if (dif < g.tol) stop = true;
                }
                if (stop) {
                    if (lane < QP) vl[lane] = lane < q ? acur : 0.0;
                    break;
                }
                aprev = acur;
                b = gb / sqrt(g2);
                ++k;
            }
            iters = k;
            if (lane == 0 && g.s.niter) g.s.niter[a] = (double)k;
        }
    } else {
        // q == 1: wx = K / ||K|| from the first pass on; the second pass sees dif = 0 (src/plswold.jl:87-91)
        if (tid == 0) {
            vl[0] = 1.0;
            if (g.s.niter) g.s.niter[a] = g.maxit >= 2 ? (0.0 < g.tol ? 2.0 : (double)g.maxit) : 1.0;
        }
    }
    (void)iters;
    __syncthreads();
    write_w_from_v<QP>(Kl, vl, p, ldr, scratch, g.s.w, g.s.r);
}

// ------------------------------------------------------------------------------------------- ROSA
// W <- columns re-orthonormalised in order: w_a = w_a - Z (Z'w_a), Z = the already treated columns, then normalised
// (src/plsrosa.jl:77-79).  W is [nlv][p]; one workgroup, sequential over a (PLS weights are orthogonal up to
// rounding, so this only removes the accumulated rounding drift — but it is what the reference returns).
__global__ __launch_bounds__(FT) void k_rosa_orthw(double *W, int p, int nlv)
{
    __shared__ double ul[1024];
    __shared__ double scratch[FT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int a = 1; a < nlv; ++a) {
        double *wa = W + (size_t)a * p;
        for (int i0 = 0; i0 < a; i0 += 1024) {
            const int cnt = min(a - i0, 1024);
            for (int i = wv; i < cnt; i += FT / 64) {
                double s = 0.0;
                for (int j = lane; j < p; j += 64) s += W[(size_t)(i0 + i) * p + j] * wa[j];
                s = jch_wave_sum(s);
                if (lane == 0) ul[i] = s;
            }
            __syncthreads();
            // all dots of a chunk use the SAME (not yet updated) w_a only when a <= 1024; beyond that the chunks form a
            // block Gram-Schmidt, which is at least as accurate
            for (int j = tid; j < p; j += FT) {
                double acc = 0.0;
                for (int i = 0; i < cnt; ++i) acc += ul[i] * W[(size_t)(i0 + i) * p + j];
                wa[j] -= acc;
            }
            __threadfence_block();
            __syncthreads();
        }
        double ss = 0.0;
        for (int j = tid; j < p; j += FT) ss += wa[j] * wa[j];
        const double inv = 1.0 / sqrt(jch_block_sum<FT>(ss, scratch));
        for (int j = tid; j < p; j += FT) wa[j] *= inv;
        __threadfence_block();
        __syncthreads();
    }
}

// Yc (column-major n x q, centred/scaled) <- Yc - T C'   (`plsrosa!` returns the deflated Y, src/plsrosa.jl:87)
__global__ __launch_bounds__(256) void k_ydeflate_all(double *__restrict__ Yc, int64_t ldy, const double *__restrict__ T,
                                                      int64_t n, const double *__restrict__ C, int q, int nlv)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        for (int k0 = 0; k0 < q; k0 += 8) {
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int a = 0; a < nlv; ++a) {
                const double t = T[(size_t)a * (size_t)n + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] += (k0 + u < q) ? t * C[(size_t)a * q + k0 + u] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + u < q) Yc[(size_t)i + (size_t)(k0 + u) * (size_t)ldy] -= acc[u];
        }
    }
}

// ------------------------------------------------------------------------------------------- launchers
static size_t sib_lds_bytes(int p, int q, int ldr, int nlv)
{
    const int QP = q <= 1 ? 1 : (q <= 2 ? 2 : (q <= 4 ? 4 : (q <= 8 ? 8 : 16)));
    const int ldk = QP | 1, lda = QP + 2;
    return sizeof(double) * ((size_t)p * ldk + (ldr + 18) + 2 * (size_t)ldr + 2 * FT + 32 + ((nlv + 1) & ~1) + 5 * (size_t)QP * lda +
                             2 * (QP + 2) + 8);
}

bool jch_sibling_supported(int p, int q, int ldr, int nlv)
{
    return q <= 16 && p <= JCH_SWEEP_MAXP && nlv <= 1024 && sib_lds_bytes(p, q, ldr, nlv) <= 150 * 1024;
}

template <typename F>
static int32_t set_lds_attr(jch_ctx *ctx, F f)
{
    JCH_HIP(ctx, hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return JCH_OK;
}

int32_t jch_launch_lv_update_simp(jch_ctx *ctx, const jch_small &s, int p, int q, int ldr, int a, int nlv, int nslice, int ldz)
{
    lvf_args g{};
    g.s = s; g.p = p; g.q = q; g.qpad = 16; g.ldr = ldr; g.nlv = nlv; g.algo = 2; g.nslice = nslice; g.ldz = ldz;
    g.raw_mu = s.variant == 2;
    if (a < 0) { g.a = 0; g.do_a = 0; g.do_b = 1; }
    else { g.a = a; g.do_a = 1; g.do_b = a + 1 < nlv ? 1 : 0; }
    const size_t lds = sib_lds_bytes(p, q, ldr, nlv);
    static jch_per_device_once attr_once;
    if (!attr_once.done(ctx->device)) {
        JCH_TRY(set_lds_attr(ctx, k_lv_update_simp<1>)); JCH_TRY(set_lds_attr(ctx, k_lv_update_simp<2>));
        JCH_TRY(set_lds_attr(ctx, k_lv_update_simp<4>)); JCH_TRY(set_lds_attr(ctx, k_lv_update_simp<8>));
        JCH_TRY(set_lds_attr(ctx, k_lv_update_simp<16>));
        attr_once.mark(ctx->device);
    }
    if (q <= 1) hipLaunchKernelGGL(k_lv_update_simp<1>, dim3(1), dim3(FT), lds, ctx->stream, g);
    else if (q <= 2) hipLaunchKernelGGL(k_lv_update_simp<2>, dim3(1), dim3(FT), lds, ctx->stream, g);
    else if (q <= 4) hipLaunchKernelGGL(k_lv_update_simp<4>, dim3(1), dim3(FT), lds, ctx->stream, g);
    else if (q <= 8) hipLaunchKernelGGL(k_lv_update_simp<8>, dim3(1), dim3(FT), lds, ctx->stream, g);
    else hipLaunchKernelGGL(k_lv_update_simp<16>, dim3(1), dim3(FT), lds, ctx->stream, g);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

int32_t jch_launch_wold_b(jch_ctx *ctx, const jch_small &s, int p, int q, int ldr, int a, int nlv, double tol, int maxit)
{
    lvf_args g{};
    g.s = s; g.p = p; g.q = q; g.qpad = 16; g.ldr = ldr; g.nlv = nlv; g.algo = 4; g.a = a; g.do_b = 1;
    g.tol = tol; g.maxit = maxit < 1 ? 1 : maxit;
    const size_t lds = sib_lds_bytes(p, q, ldr, nlv);
    static jch_per_device_once attr_once;
    if (!attr_once.done(ctx->device)) {
        JCH_TRY(set_lds_attr(ctx, k_wold_b<1>)); JCH_TRY(set_lds_attr(ctx, k_wold_b<2>)); JCH_TRY(set_lds_attr(ctx, k_wold_b<4>));
        JCH_TRY(set_lds_attr(ctx, k_wold_b<8>)); JCH_TRY(set_lds_attr(ctx, k_wold_b<16>));
        attr_once.mark(ctx->device);
    }
    if (q <= 1) hipLaunchKernelGGL(k_wold_b<1>, dim3(1), dim3(FT), lds, ctx->stream, g);
    else if (q <= 2) hipLaunchKernelGGL(k_wold_b<2>, dim3(1), dim3(FT), lds, ctx->stream, g);
    else if (q <= 4) hipLaunchKernelGGL(k_wold_b<4>, dim3(1), dim3(FT), lds, ctx->stream, g);
    else if (q <= 8) hipLaunchKernelGGL(k_wold_b<8>, dim3(1), dim3(FT), lds, ctx->stream, g);
    else hipLaunchKernelGGL(k_wold_b<16>, dim3(1), dim3(FT), lds, ctx->stream, g);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

int32_t jch_launch_rosa_orthw(jch_ctx *ctx, double *W, int p, int nlv)
{
    hipLaunchKernelGGL(k_rosa_orthw, dim3(1), dim3(FT), 0, ctx->stream, W, p, nlv);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

int32_t jch_launch_ydeflate_all(jch_ctx *ctx, double *Yc, int64_t ldy, const double *T, int64_t n, const double *C, int q, int nlv)
{
    int64_t nb = (n + 255) / 256;
    if (nb > ctx->cus * 8) nb = ctx->cus * 8;
    hipLaunchKernelGGL(k_ydeflate_all, dim3((unsigned)nb), dim3(256), 0, ctx->stream, Yc, ldy, T, n, C, q, nlv);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}
