// OPT-IN variant "kernel algorithm #2" of plskern.  Not in the reference (Jchemo implements
// Dayal & MacGregor's improved kernel algorithm #1, src/plskern.jl:35-36 cites the paper that holds both): the p x p
// Gram matrix G = X'DX is formed ONCE, after which the per-LV quantities that algorithm #1 gets from a sweep over X
// come from G:    zp = X'D X r = G r ,   tt = t'D t = r' G r ,   and all scores at the end as T = X R.
// X is then read 2 + 1 + 1 + 1 times in total (means, centre/XtY, Gram, scores) instead of 3 + nlv, and the LV loop
// needs NO pass over X and NO collective: with row sharding only G (p x p) is all-reduced, once.
// Same results up to rounding (the sums are associated differently); therefore opt-in (desc->reserved = 1), the
// default and the headline benchmark stay on algorithm #1.
//
//   k_syrk        G partials on v_mfma_f64_16x16x4_f64: 128 x 128 output tile per workgroup (upper tile pairs only),
//                 rows split over workgroups (split-K), operands staged row-major -> LDS with register prefetch
//   k_syrk_reduce fixed-order sum over the row splits + symmetric fill
//   k_gmatvec     zp = G r (one wave per row of G), per LV
//   k_scores      T = Xr R  (n x nlv) on MFMA f64, row-major tile through LDS
// Bound: MFMA f64 (n p^2 flop at 78.6 TF) for k_syrk; HBM for k_scores.
// Measured (cfg2): k_syrk 7.4 ms = 44 TF executed (56 % of peak).  An XCD-aware block mapping (all tile pairs of one row
// split on one XCD, to share the slabs through that L2) was tried and made no difference (7.8 ms): not L2-bound.
#include <stdlib.h>

#include <algorithm>

#include "jch_internal.h"

typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef double v4f64 __attribute__((ext_vector_type(4)));

#define SY_LD 144   // LDS row stride (doubles): 128 + 16 -> consecutive k-rows land 32 banks apart (conflict-free b64 reads)
#define SY_KB 32    // rows per staged chunk

__global__ __launch_bounds__(256, 2) void k_syrk(const double *__restrict__ Xr, int64_t n, int p, int ldr,
                                                 const double *__restrict__ dw, double *__restrict__ Gpart, int nsplit,
                                                 int nblk /*column blocks of 128*/)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *At = lds;                      // [SY_KB][SY_LD]  d_r * X[r][i-block]
    double *Bt = lds + SY_KB * SY_LD;      // [SY_KB][SY_LD]  X[r][j-block]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // upper-triangular tile pair from the linear pair index
    int pr = blockIdx.y, bi = 0;
    while (pr >= nblk - bi) { pr -= nblk - bi; ++bi; }
    const int bj = bi + pr;
    const bool diag = bi == bj;
    const int64_t per = ((n + nsplit - 1) / nsplit + SY_KB - 1) / SY_KB * SY_KB;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = std::min<int64_t>(n, r0 + per);
    const int ci = 128 * bi + 2 * lane, cj = 128 * bj + 2 * lane;
    const bool vi = ci < ldr, vj = cj < ldr;
    v4f64 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = v4f64{0.0, 0.0, 0.0, 0.0};
    v2f64 va[8], vb[8];
    double dv[8];
    auto prefetch = [&](int64_t rb) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int64_t row = rb + wv * 8 + it;
            const bool live = row < r1;
            va[it] = (live && vi) ? __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(Xr + (size_t)row * ldr + ci)) : v2f64{0.0, 0.0};
            if (!diag) vb[it] = (live && vj) ? __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(Xr + (size_t)row * ldr + cj)) : v2f64{0.0, 0.0};
            dv[it] = live ? dw[row] : 0.0;
        }
    };
    const int qi = wv >> 1, qj = wv & 1;
    if (r0 < r1) prefetch(r0);
    for (int64_t rb = r0; rb < r1; rb += SY_KB) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = wv * 8 + it;
            const v2f64 sa = v2f64{va[it].x * dv[it], va[it].y * dv[it]};
            *reinterpret_cast<v2f64 *>(At + row * SY_LD + 2 * lane) = sa;
            *reinterpret_cast<v2f64 *>(Bt + row * SY_LD + 2 * lane) = diag ? va[it] : vb[it];
        }
        __syncthreads();
        if (rb + SY_KB < r1) prefetch(rb + SY_KB);
#pragma unroll
        for (int kk = 0; kk < SY_KB / 4; ++kk) {
            const int krow = 4 * kk + (lane >> 4);
            double a[4], b[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                a[m] = At[krow * SY_LD + 64 * qi + 16 * m + (lane & 15)];
                b[m] = Bt[krow * SY_LD + 64 * qj + 16 * m + (lane & 15)];
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
        }
        __syncthreads();
    }
    // D[m][n]: n = lane & 15, m = (lane >> 4) + 4 reg
    double *gp = Gpart + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (128 * 128);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 4; ++nj)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int m = 64 * qi + 16 * mi + (lane >> 4) + 4 * reg, nn = 64 * qj + 16 * nj + (lane & 15);
                gp[m * 128 + nn] = acc[mi][nj][reg];
            }
}

// G[i][j] (ld = ldg) = sum over splits, both triangles
__global__ __launch_bounds__(256) void k_syrk_reduce(const double *__restrict__ Gpart, int nsplit, int npairs, int nblk, int p,
                                                     double *__restrict__ G, int ldg)
{
    int pr = blockIdx.y, bi = 0;
    while (pr >= nblk - bi) { pr -= nblk - bi; ++bi; }
    const int bj = bi + pr;
    const int e = blockIdx.x * 256 + threadIdx.x;   // element of the 128 x 128 tile
    const int m = e >> 7, nn = e & 127;
    const int i = 128 * bi + m, j = 128 * bj + nn;
    if (i >= p || j >= p) return;
    double s = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) s += Gpart[((size_t)sp * npairs + blockIdx.y) * (128 * 128) + e];
    G[(size_t)i * ldg + j] = s;
    if (bi != bj) G[(size_t)j * ldg + i] = s;
}

// zp = G r : one wave per row of G; out[i] (i < p), pad entries zero
__global__ __launch_bounds__(256) void k_gmatvec(const double *__restrict__ G, int ldg, int p, int ldr, const double *__restrict__ r,
                                                 double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= ldr) return;
    double s = 0.0;
    if (i < p) {
        const double *gr = G + (size_t)i * ldg;
        double s0 = 0.0, s1 = 0.0;
        int j = lane;
        for (; j + 64 < p; j += 128) { s0 += gr[j] * r[j]; s1 += gr[j + 64] * r[j + 64]; }
        for (; j < p; j += 64) s0 += gr[j] * r[j];
        s = jch_wave_sum(s0 + s1);
    }
    if (lane == 0) out[i] = s;
}

// T (n x nlv, column-major ld n) = Xr (n x ldr row-major) * Rm', Rm = R stored [lv][p] (== Julia's p x nlv)
__global__ __launch_bounds__(256) void k_scores(const double *__restrict__ Xr, int64_t n, int p, int ldr, const double *__restrict__ Rm,
                                                int nlv, double *__restrict__ T)
{
    __shared__ __attribute__((aligned(16))) double xt[64 * 66];
    __shared__ double bl[64 * 33];   // [col within tile][lv], lv padded to 32 (+1)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lv0 = blockIdx.y * 32;
    const int nl = std::min(32, nlv - lv0);
    const int64_t nchunks = (n + 63) / 64;
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t i0 = c * 64;
        v4f64 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        for (int j0 = 0; j0 < ldr; j0 += 64) {
            __syncthreads();
            // X tile: 64 rows x 64 cols, 32 lanes x 16 B per row, 2 rows per wave-instruction
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = 2 * (wv + 4 * k) + (lane >> 5), col = 2 * (lane & 31);
                const int64_t i = i0 + row;
                v2f64 v = {0.0, 0.0};
                if (i < n && j0 + col < ldr) v = __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(Xr + (size_t)i * ldr + j0 + col));
                *reinterpret_cast<v2f64 *>(xt + row * 66 + col) = v;
            }
            for (int e = tid; e < 64 * 32; e += 256) {
                const int col = e >> 5, l = e & 31;
                bl[col * 33 + l] = (j0 + col < p && l < nl) ? Rm[(size_t)(lv0 + l) * p + j0 + col] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int kc = 4 * kk + (lane >> 4);
                const double a = xt[(16 * wv + (lane & 15)) * 66 + kc];     // A[m = row][k = col]
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bl[kc * 33 + (lane & 15)], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bl[kc * 33 + 16 + (lane & 15)], acc1, 0, 0, 0);
            }
        }
        // D[m = row (lane>>4) + 4 reg][n = lv lane&15]
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t i = i0 + 16 * wv + (lane >> 4) + 4 * reg;
            const int l0 = lane & 15;
            if (i < n) {
                if (l0 < nl) T[(size_t)i + (size_t)(lv0 + l0) * (size_t)n] = acc0[reg];
                if (16 + l0 < nl) T[(size_t)i + (size_t)(lv0 + 16 + l0) * (size_t)n] = acc1[reg];
            }
        }
    }
}

// ---------------------------------------------------------------- launchers
int32_t jch_launch_syrk(jch_ctx *ctx, const double *Xr, int64_t n, int p, int ldr, const double *d, double *G, int ldg)
{
    const int nblk = (ldr + 127) / 128, npairs = nblk * (nblk + 1) / 2;
    int nsplit = std::max(1, (ctx->cus * 2) / npairs);
    if ((int64_t)nsplit * SY_KB > n) nsplit = (int)std::max<int64_t>(1, n / SY_KB);
    JCH_TRY(jch_reserve(ctx, ctx->kpart, sizeof(double) * (size_t)nsplit * npairs * 128 * 128));   // (free after the prologue)
    double *Gpart = (double *)ctx->kpart.ptr;
    const size_t lds = sizeof(double) * 2 * SY_KB * SY_LD;
    static jch_per_device_once attr;
    if (!attr.done(ctx->device)) { JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_syrk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr.mark(ctx->device); }
    (void)jch_ev(ctx);
    hipLaunchKernelGGL(k_syrk, dim3(nsplit, npairs), dim3(256), lds, ctx->stream, Xr, n, p, ldr, d, Gpart, nsplit, nblk);
    (void)jch_ev(ctx);
    hipLaunchKernelGGL(k_syrk_reduce, dim3(64, npairs), dim3(256), 0, ctx->stream, Gpart, nsplit, npairs, nblk, p, G, ldg);
    JCH_TRY(jch_allreduce_f64(ctx, G, (size_t)p * ldg));   // the ONLY n-dependent collective of the LV phase
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

int32_t jch_launch_gmatvec(jch_ctx *ctx, const double *G, int ldg, int p, int ldr, const double *r, double *out)
{
    hipLaunchKernelGGL(k_gmatvec, dim3((ldr + 3) / 4), dim3(256), 0, ctx->stream, G, ldg, p, ldr, r, out);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

int32_t jch_launch_scores(jch_ctx *ctx, const double *Xr, int64_t n, int p, int ldr, const double *Rm, int nlv, double *T)
{
    const int64_t nchunks = (n + 63) / 64;
    const int lvt = (nlv + 31) / 32;
    const int nbx = (int)std::max<int64_t>(1, std::min<int64_t>(nchunks, (int64_t)ctx->cus * 4));
    hipLaunchKernelGGL(k_scores, dim3(nbx, lvt), dim3(256), 0, ctx->stream, Xr, n, p, ldr, Rm, nlv, T);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}
