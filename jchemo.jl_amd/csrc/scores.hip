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
