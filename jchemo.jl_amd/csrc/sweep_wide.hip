// Fallback of the per-LV sweep for rows wider than the register-resident kernel holds (p > 2048; e.g. dkplsr's
// plskern! on an n x n Gram matrix, src/dkplsr.jl:122).  Two passes over X per LV — exactly the reference's own
// schedule (src/plskern.jl:162 then :167):
//   pass 1  k_rowdot   t = X r, tt = t'Dt (and c_raw = Y'Dt for plsnipals), wave per row, any width
//   pass 2  k_colacc   zp[c0 .. c0+2048) = sum_i d_i t_i x_i[c0 ..], one launch per 2048-column panel
// Same deterministic two-stage reductions as sweep.hip.  Algorithmic bytes per LV: 2 n ld 8.
#include <algorithm>

#include "jch_internal.h"

typedef double v2f64 __attribute__((ext_vector_type(2)));

template <bool NIPALS>
__global__ __launch_bounds__(256) void k_rowdot(const double *__restrict__ Xr, int64_t n, int ldr, const double *__restrict__ dw,
                                                const double *__restrict__ rvec, const double *__restrict__ Yr, int qpad,
                                                double *__restrict__ tcol, double *__restrict__ part, int ldpart)
{
    __shared__ double sc[4 * 65];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double tt = 0.0, cacc = 0.0;
    const int nv = ldr / 2;   // ldr is even
    for (int64_t row = (int64_t)blockIdx.x * 4 + wv; row < n; row += (int64_t)gridDim.x * 4) {
        const v2f64 *rp = reinterpret_cast<const v2f64 *>(Xr + (size_t)row * (size_t)ldr);
        const v2f64 *rr = reinterpret_cast<const v2f64 *>(rvec);
        double s0 = 0.0, s1 = 0.0;
        int c = lane;
        for (; c + 192 < nv; c += 256) {
            const v2f64 a0 = __builtin_nontemporal_load(rp + c), a1 = __builtin_nontemporal_load(rp + c + 64),
                        a2 = __builtin_nontemporal_load(rp + c + 128), a3 = __builtin_nontemporal_load(rp + c + 192);
            const v2f64 b0 = rr[c], b1 = rr[c + 64], b2 = rr[c + 128], b3 = rr[c + 192];
            s0 += a0.x * b0.x + a0.y * b0.y; s1 += a1.x * b1.x + a1.y * b1.y;
            s0 += a2.x * b2.x + a2.y * b2.y; s1 += a3.x * b3.x + a3.y * b3.y;
        }
        for (; c < nv; c += 64) { const v2f64 a = rp[c], b = rr[c]; s0 += a.x * b.x + a.y * b.y; }
        const double t = jch_wave_sum(s0 + s1);
        const double dt = dw[row] * t;
        tt += dt * t;
        if (NIPALS) cacc += dt * (lane < qpad ? Yr[(size_t)row * qpad + lane] : 0.0);
        if (lane == 0) tcol[row] = t;
    }
    if (lane == 0) sc[wv * 65] = tt;
    if (NIPALS) sc[wv * 65 + 1 + lane] = cacc;
    __syncthreads();
    double *prow = part + (size_t)blockIdx.x * ldpart;
    if (threadIdx.x == 0) prow[0] = ((sc[0] + sc[65]) + sc[130]) + sc[195];
    if (NIPALS && threadIdx.x < qpad) {
        const int k = 1 + threadIdx.x;
        prow[k] = ((sc[k] + sc[65 + k]) + sc[130 + k]) + sc[195 + k];
    }
}

// zp over one panel of <= 2048 columns starting at column c0 (t given)
__global__ __launch_bounds__(256) void k_colacc(const double *__restrict__ Xr, int64_t n, int ldr, int c0, int width,
                                                const double *__restrict__ dw, const double *__restrict__ tcol,
                                                double *__restrict__ part, int ldpart)
{
    extern __shared__ __attribute__((aligned(16))) double red[];  // [4][2048]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v2f64 zp[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) zp[k] = v2f64{0.0, 0.0};
    for (int64_t row = (int64_t)blockIdx.x * 4 + wv; row < n; row += (int64_t)gridDim.x * 4) {
        const v2f64 *rp = reinterpret_cast<const v2f64 *>(Xr + (size_t)row * (size_t)ldr + c0) + lane;
        v2f64 x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = (2 * lane + 128 * k < width) ? __builtin_nontemporal_load(rp + 64 * k) : v2f64{0.0, 0.0};
        const double dt = dw[row] * tcol[row];
#pragma unroll
        for (int k = 0; k < 16; ++k) { zp[k].x += dt * x[k].x; zp[k].y += dt * x[k].y; }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) *reinterpret_cast<v2f64 *>(red + wv * 2048 + 2 * lane + 128 * k) = zp[k];
    __syncthreads();
    double *prow = part + (size_t)blockIdx.x * ldpart;
    for (int c = threadIdx.x; c < width; c += 256) prow[c] = ((red[c] + red[2048 + c]) + red[4096 + c]) + red[6144 + c];
}

__global__ __launch_bounds__(1024) void k_reduce_cols(const double *__restrict__ part, int nb, int ldpart, int m, double *__restrict__ out)
{
    __shared__ double sc[16][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0.0;
    if (c < m)
        for (int b = g; b < nb; b += 16) s += part[(size_t)b * ldpart + c];
    sc[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < m) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sc[k][cl];
        out[c] = t;
    }
}

int32_t jch_launch_sweep_wide(jch_ctx *ctx, const double *Xr, int64_t n, int ldr, const double *d, const double *rvec,
                              const double *Yr, int qpad, bool nipals, double *tcol, double *zt)
{
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, (int64_t)ctx->cus * 4));
    const int ldp1 = 72, ldp2 = 2048;
    JCH_TRY(jch_reserve(ctx, ctx->part, sizeof(double) * (size_t)nb * ldp2));
    double *part = (double *)ctx->part.ptr;
    static jch_per_device_once attr;
    if (!attr.done(ctx->device)) { JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_colacc, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr.mark(ctx->device); }
    (void)jch_ev(ctx);
    if (nipals) hipLaunchKernelGGL(k_rowdot<true>, dim3(nb), dim3(256), 0, ctx->stream, Xr, n, ldr, d, rvec, Yr, qpad, tcol, part, ldp1);
    else hipLaunchKernelGGL(k_rowdot<false>, dim3(nb), dim3(256), 0, ctx->stream, Xr, n, ldr, d, rvec, Yr, qpad, tcol, part, ldp1);
    const int m1 = 1 + (nipals ? qpad : 0);
    hipLaunchKernelGGL(k_reduce_cols, dim3((m1 + 63) / 64), dim3(1024), 0, ctx->stream, part, nb, ldp1, m1, zt + ldr);   // tt, c_raw
    const int nb2 = (int)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, (int64_t)ctx->cus * 2));
    for (int c0 = 0; c0 < ldr; c0 += 2048) {
        const int width = std::min(2048, ldr - c0);
        hipLaunchKernelGGL(k_colacc, dim3(nb2), dim3(256), sizeof(double) * 4 * 2048, ctx->stream, Xr, n, ldr, c0, width, d, tcol, part, ldp2);
        hipLaunchKernelGGL(k_reduce_cols, dim3((width + 63) / 64), dim3(1024), 0, ctx->stream, part, nb2, ldp2, width, zt + c0);
    }
    (void)jch_ev(ctx);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

// c_raw = Y'Dt for a response block wider than the NIPALS sweep's one-lane-per-response layout (qpad > 64; round 4): the column
// accumulation of k_colacc on the row-major Yr [n][qpad], panels of 2048 columns.  out [qpad].
int32_t jch_launch_ytdt(jch_ctx *ctx, const double *Yr, int64_t n, int qpad, const double *d, const double *tcol, double *out)
{
    const int nb2 = (int)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, (int64_t)ctx->cus * 2));
    const int ldp2 = 2048;
    JCH_TRY(jch_reserve(ctx, ctx->part, sizeof(double) * (size_t)nb2 * ldp2));
    double *part = (double *)ctx->part.ptr;
    static jch_per_device_once attr;
    if (!attr.done(ctx->device)) { JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_colacc, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr.mark(ctx->device); }
    for (int c0 = 0; c0 < qpad; c0 += 2048) {
        const int width = std::min(2048, qpad - c0);
        hipLaunchKernelGGL(k_colacc, dim3(nb2), dim3(256), sizeof(double) * 4 * 2048, ctx->stream, Yr, n, qpad, c0, width, d, tcol, part, ldp2);
        hipLaunchKernelGGL(k_reduce_cols, dim3((width + 63) / 64), dim3(1024), 0, ctx->stream, part, nb2, ldp2, width, out + c0);
    }
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}
