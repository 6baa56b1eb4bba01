// Sufficient statistics of prediction scores (gridscorelv / gridcvlv).  One pass over the
// prediction matrix gives, per prediction column c (= nlv level a * q + y column k) and over the rows selected by
// `mask`:  sum e, sum e^2, sum y e  (e = y - pred), and per y column sum y, sum y^2, and the row count — enough for
// msep / rmsep / ssr / bias / r2 / cor2 (src/scores.jl:25-32,54-62,155-158,190-196,268,426-429) without bringing
// n-sized data back to the host.  Deterministic two-stage reduction.
#include <algorithm>
#include <vector>

#include "jch_internal.h"

__global__ __launch_bounds__(256) void k_score_sums(const double *__restrict__ Pred, int64_t m, int64_t ldp, const double *__restrict__ Y,
                                                    int q, int64_t ldy, const double *__restrict__ mask, double *__restrict__ part)
{
    __shared__ double sc[4];
    const int c = blockIdx.y, k = c % q;
    const double *pc = Pred + (size_t)c * (size_t)ldp, *yc = Y + (size_t)k * (size_t)ldy;
    double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
        const double w = mask ? mask[i] : 1.0;
        if (w != 0.0) {
            const double y = yc[i], e = y - pc[i];
            s[0] += w * e; s[1] += w * e * e; s[2] += w * y * e; s[3] += w * y; s[4] += w * y * y; s[5] += w;
        }
    }
    double *out = part + ((size_t)blockIdx.x * gridDim.y + c) * 6;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double t = jch_block_sum<256>(s[j], sc);
        if (threadIdx.x == 0) out[j] = t;
    }
}

__global__ __launch_bounds__(256) void k_score_reduce(const double *__restrict__ part, int nbx, int ncol, double *__restrict__ out)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ncol * 6) return;
    double s = 0.0;
    for (int b = 0; b < nbx; ++b) s += part[(size_t)b * ncol * 6 + e];
    out[e] = s;
}

// Pred m x ncol (ncol = levels * q, level-major) and Y m x q, mask m (may be NULL) [loc]; sums: ncol x 6 HOST.
// With a communicator the sums cover all ranks' rows.
extern "C" int32_t jch_score_sums(jch_ctx *ctx, int32_t loc, const double *Pred, int64_t m, int64_t ncol, int64_t ldp,
                                  const double *Y, int64_t q, int64_t ldy, const double *mask, double *sums)
{
    if (!ctx) return JCH_EINVAL;
    if (!Pred || !Y || !sums || m < 1 || ncol < 1 || q < 1 || ncol % q != 0 || ldp < m || ldy < m)
        return jch_fail(ctx, JCH_EINVAL, "jch_score_sums: bad arguments (ncol must be a multiple of q)");
    if (loc != JCH_LOC_HOST && loc != JCH_LOC_DEVICE) return jch_fail(ctx, JCH_EINVAL, "jch_score_sums: bad loc");
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    const double *dP = Pred, *dY = Y, *dM = mask;
    int64_t ldpd = ldp, ldyd = ldy;
    if (loc == JCH_LOC_HOST) {
        JCH_TRY(jch_reserve(ctx, ctx->xq, sizeof(double) * ((size_t)m * ncol + (size_t)m * q + (size_t)m)));
        double *b = (double *)ctx->xq.ptr;
        if (ldp == m) JCH_HIP(ctx, hipMemcpyAsync(b, Pred, sizeof(double) * (size_t)m * ncol, hipMemcpyHostToDevice, ctx->stream));
        else JCH_HIP(ctx, hipMemcpy2DAsync(b, sizeof(double) * m, Pred, sizeof(double) * ldp, sizeof(double) * m, ncol, hipMemcpyHostToDevice, ctx->stream));
        dP = b; ldpd = m; b += (size_t)m * ncol;
        if (ldy == m) JCH_HIP(ctx, hipMemcpyAsync(b, Y, sizeof(double) * (size_t)m * q, hipMemcpyHostToDevice, ctx->stream));
        else JCH_HIP(ctx, hipMemcpy2DAsync(b, sizeof(double) * m, Y, sizeof(double) * ldy, sizeof(double) * m, q, hipMemcpyHostToDevice, ctx->stream));
        dY = b; ldyd = m; b += (size_t)m * q;
        if (mask) { JCH_HIP(ctx, hipMemcpyAsync(b, mask, sizeof(double) * (size_t)m, hipMemcpyHostToDevice, ctx->stream)); dM = b; }
    }
    const int nbx = (int)std::max<int64_t>(1, std::min<int64_t>((m + 255) / 256, std::max<int64_t>(1, (int64_t)ctx->cus * 8 / ncol)));
    JCH_TRY(jch_reserve(ctx, ctx->colpart, sizeof(double) * ((size_t)nbx * ncol * 6 + (size_t)ncol * 6) + 4096));
    double *part = (double *)ctx->colpart.ptr, *out = part + (size_t)nbx * ncol * 6;
    hipLaunchKernelGGL(k_score_sums, dim3(nbx, (unsigned)ncol), dim3(256), 0, ctx->stream, dP, m, ldpd, dY, (int)q, ldyd, dM, part);
    hipLaunchKernelGGL(k_score_reduce, dim3((unsigned)((ncol * 6 + 255) / 256)), dim3(256), 0, ctx->stream, part, nbx, (int)ncol, out);
    JCH_TRY(jch_allreduce_f64(ctx, out, (size_t)ncol * 6));
    JCH_HIP(ctx, hipGetLastError());
    JCH_HIP(ctx, hipMemcpyAsync(sums, out, sizeof(double) * (size_t)ncol * 6, hipMemcpyDeviceToHost, ctx->stream));
    JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JCH_OK;
}

// ---- the same statistics for EVERY nlv of a range straight from the scores (round 4, second half): the predictions of a fold
// of gridcvlv / of gridscorelv for nlv = lo..hi are running sums over the score columns, pred_a = ymeans + sum_{l <= a} t_l (c_l .*
// yscales)' (src/plskern.jl:207-217, :226-238 on the rows' scores), so the (hi - lo + 1) q prediction columns never have to exist:
// a thread walks its rows' score columns once, carries the running prediction of ONE response and adds e = y - pred to the
// level's three sums.  Reads m x k scores (once per response slice: L2 / Infinity Cache hits after the first) + Y + mask instead
// of writing and re-reading m x (hi - lo + 1) q predictions (2.08 GB twice at cfg2).  (Measured and dropped, round 4: a wave per
// response with the four waves of a block on the same 64 rows — the score columns from the CU's L1 for three of them, wave sums
// instead of block sums — needs 324 registers for 16-column load batches, runs one wave per SIMD and takes 1.18 ms against 0.58
// for a fold of cfg2.)  Levels beyond the fit's k LVs repeat level k
// (the reference clamps, src/plskern.jl:228).  Same two-stage fixed-order reduction as k_score_sums; part laid out [nbx][ncol][6].
#define SLV_LE 32     // levels per launch (accumulators in registers: 3 per level)
__global__ __launch_bounds__(256) void k_score_sums_lv(const double *__restrict__ T, int64_t m, int64_t ldt, int kfit, const double *__restrict__ Cs,
                                                       const double *__restrict__ y0, const double *__restrict__ Y, int q, int64_t ldy,
                                                       const double *__restrict__ mask, int lo, int le, int ncol, int col0, double *__restrict__ part)
{
    __shared__ double sc[4];
    const int k = blockIdx.y;
    const double *yc = Y + (size_t)k * (size_t)ldy;
    double s[SLV_LE][3];
#pragma unroll
    for (int a = 0; a < SLV_LE; ++a) { s[a][0] = 0.0; s[a][1] = 0.0; s[a][2] = 0.0; }
    double sy = 0.0, syy = 0.0, sw = 0.0;
    const double ym = y0[k];
    const int npre = min(lo - 1, kfit);                    // score columns already inside the first level of this launch
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
        const double w = mask ? mask[i] : 1.0;
        if (w == 0.0) continue;                            // (as k_score_sums: an unselected row may hold anything)
        const double y = yc[i];
        sy += w * y; syy += w * y * y; sw += w;
        double pred = ym;
        for (int a = 1; a <= npre; ++a) pred += T[i + (int64_t)(a - 1) * ldt] * Cs[(size_t)(a - 1) * q + k];
        // eight score columns at a time: unconditional loads (clamped column, the term dropped by a select), then the eight levels
#pragma unroll
        for (int u0 = 0; u0 < SLV_LE; u0 += 8) {
            if (u0 >= le) continue;                        // block-uniform
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int a = lo + u0 + u;
                t[u] = T[i + (int64_t)max(0, min(a, kfit) - 1) * ldt];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int a = lo + u0 + u;                 // level of slot u0 + u
                const bool term = a >= 1 && a <= kfit, live = u0 + u < le;
                const double c = Cs[(size_t)max(0, min(a, kfit) - 1) * q + k];
                pred += term ? t[u] * c : 0.0;
                const double e = live ? y - pred : 0.0;
                s[u0 + u][0] += w * e; s[u0 + u][1] += w * e * e; s[u0 + u][2] += w * y * e;
            }
        }
    }
    const double t3 = jch_block_sum<256>(sy, sc), t4 = jch_block_sum<256>(syy, sc), t5 = jch_block_sum<256>(sw, sc);
#pragma unroll
    for (int u = 0; u < SLV_LE; ++u) {
        if (u >= le) continue;                             // block-uniform
        const double t0 = jch_block_sum<256>(s[u][0], sc), t1 = jch_block_sum<256>(s[u][1], sc), t2 = jch_block_sum<256>(s[u][2], sc);
        if (threadIdx.x == 0) {
            double *out = part + ((size_t)blockIdx.x * ncol + col0 + (size_t)u * q + k) * 6;
            out[0] = t0; out[1] = t1; out[2] = t2; out[3] = t3; out[4] = t4; out[5] = t5;
        }
    }
}

// T m x kfit (the rows' scores: `transform(fm, X)`, or the T of a fit whose held-out rows carried weight zero), Y m x q, mask m (may be
// NULL) [loc]; C q x kfit (ld q), ymeans / yscales (q; NULL = 0 / 1) HOST; sums: (nlv_hi - nlv_lo + 1) q x 6 HOST, level-major like
// jch_score_sums on the predictions for nlv_lo..nlv_hi.  With a communicator the sums cover all ranks' rows.
extern "C" int32_t jch_score_sums_lv(jch_ctx *ctx, int32_t loc, const double *T, int64_t m, int64_t kfit, int64_t ldt, const double *C,
                                     const double *ymeans, const double *yscales, const double *Y, int64_t q, int64_t ldy, const double *mask,
                                     int32_t nlv_lo, int32_t nlv_hi, double *sums)
{
    if (!ctx) return JCH_EINVAL;
    if (!Y || !sums || m < 1 || q < 1 || kfit < 0 || (kfit > 0 && (!T || !C || ldt < m)) || ldy < m || nlv_lo < 0 || nlv_hi < nlv_lo)
        return jch_fail(ctx, JCH_EINVAL, "jch_score_sums_lv: bad arguments");
    if (loc != JCH_LOC_HOST && loc != JCH_LOC_DEVICE) return jch_fail(ctx, JCH_EINVAL, "jch_score_sums_lv: bad loc");
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    const int le_all = nlv_hi - nlv_lo + 1;
    const int64_t ncol = (int64_t)le_all * q;
    const double *dT = kfit > 0 ? T : Y, *dY = Y, *dM = mask;
    int64_t ldtd = kfit > 0 ? ldt : ldy, ldyd = ldy;
    if (loc == JCH_LOC_HOST) {
        JCH_TRY(jch_reserve(ctx, ctx->xq, sizeof(double) * ((size_t)m * kfit + (size_t)m * q + (size_t)m)));
        double *b = (double *)ctx->xq.ptr;
        if (kfit > 0) {
            if (ldt == m) JCH_HIP(ctx, hipMemcpyAsync(b, T, sizeof(double) * (size_t)m * kfit, hipMemcpyHostToDevice, ctx->stream));
            else JCH_HIP(ctx, hipMemcpy2DAsync(b, sizeof(double) * m, T, sizeof(double) * ldt, sizeof(double) * m, kfit, hipMemcpyHostToDevice, ctx->stream));
        }
        dT = b; ldtd = m; b += (size_t)m * kfit;
        if (kfit == 0) dT = b;                             // (no score column is ever used; the kernel's clamped loads stay in range)
        if (ldy == m) JCH_HIP(ctx, hipMemcpyAsync(b, Y, sizeof(double) * (size_t)m * q, hipMemcpyHostToDevice, ctx->stream));
        else JCH_HIP(ctx, hipMemcpy2DAsync(b, sizeof(double) * m, Y, sizeof(double) * ldy, sizeof(double) * m, q, hipMemcpyHostToDevice, ctx->stream));
        dY = b; ldyd = m; b += (size_t)m * q;
        if (mask) { JCH_HIP(ctx, hipMemcpyAsync(b, mask, sizeof(double) * (size_t)m, hipMemcpyHostToDevice, ctx->stream)); dM = b; }
    }
    // small constants: Cs[l][k] = C[k][l] yscales[k], ymeans
    std::vector<double> hc((size_t)kfit * q + q + 8, 0.0);
    for (int64_t l = 0; l < kfit; ++l)
        for (int64_t k = 0; k < q; ++k) hc[(size_t)l * q + k] = C[k + (size_t)l * q] * (yscales ? yscales[k] : 1.0);
    for (int64_t k = 0; k < q; ++k) hc[(size_t)kfit * q + k] = ymeans ? ymeans[k] : 0.0;
    const int nbx = (int)std::max<int64_t>(1, std::min<int64_t>((m + 255) / 256, std::max<int64_t>(1, (int64_t)ctx->cus * 8 / q)));
    JCH_TRY(jch_reserve(ctx, ctx->colpart, sizeof(double) * ((size_t)nbx * ncol * 6 + (size_t)ncol * 6 + hc.size()) + 4096));
    double *part = (double *)ctx->colpart.ptr, *out = part + (size_t)nbx * ncol * 6, *dC = out + (size_t)ncol * 6;
    JCH_HIP(ctx, hipMemcpyAsync(dC, hc.data(), sizeof(double) * hc.size(), hipMemcpyHostToDevice, ctx->stream));
    for (int l0 = 0; l0 < le_all; l0 += SLV_LE)
        hipLaunchKernelGGL(k_score_sums_lv, dim3(nbx, (unsigned)q), dim3(256), 0, ctx->stream, dT, m, ldtd, (int)kfit, dC, dC + (size_t)kfit * q, dY, (int)q,
                           ldyd, dM, nlv_lo + l0, std::min(SLV_LE, le_all - l0), (int)ncol, l0 * (int)q, part);
    hipLaunchKernelGGL(k_score_reduce, dim3((unsigned)((ncol * 6 + 255) / 256)), dim3(256), 0, ctx->stream, part, nbx, (int)ncol, out);
    JCH_TRY(jch_allreduce_f64(ctx, out, (size_t)ncol * 6));
    JCH_HIP(ctx, hipGetLastError());
    JCH_HIP(ctx, hipMemcpyAsync(sums, out, sizeof(double) * (size_t)ncol * 6, hipMemcpyDeviceToHost, ctx->stream));
    JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (also keeps `hc` alive until its upload is done)
    return JCH_OK;
}
