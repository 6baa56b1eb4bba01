// K4 — the fused per-latent-variable sweep: ONE read of X per LV.
//
// Replaces the two dgemv of the reference loop (src/plskern.jl:162 `t = X r` and :167 `zp = X' D t`, with
// :163-164 `tt = t' D t` and the T column store :170; src/plsnipals.jl:78-83 for the NIPALS variant, which
// also needs c_raw = Y' D t).  The reference reads X twice per LV; t_i = x_i . r is row-local, so tt and
// zp = sum_i d_i t_i x_i are accumulated while the row is still in registers.
//
// Layout: Xr is the ROW-major working copy (ld = ldr, even; pad column zero).  One wave owns R consecutive
// rows at a time; lane l holds columns {2l, 2l+1} + 128k (k < KC) of each row as double2 (16-B loads, one
// wave-instruction = 1 KiB contiguous).  Phase 1: per-lane partial dot with the r fragment (registers),
// 64-lane butterfly -> t_i in every lane.  Phase 2: zp fragment += (d_i t_i) * x fragment.  No LDS in the
// loop; LDS only for the 4-wave combine at the end.  Per-block partials are written to `part` and summed in
// a fixed order by k_reduce_part: bit-reproducible run to run (no float atomics).
//
// Bound: HBM.  Algorithmic bytes per launch = n*ldr*8 (X) + 16 n (weights read + T column write).
#include "jch_internal.h"

typedef double v2f64 __attribute__((ext_vector_type(2)));

template <int KC, int R, bool NIPALS>
__global__ __launch_bounds__(256) void k_sweep(const double *__restrict__ Xr, int64_t n, int ldr,
                                               const double *__restrict__ dw, const double *__restrict__ rvec,
                                               const double *__restrict__ Yr, int qpad, double *__restrict__ tcol,
                                               double *__restrict__ part, int ldpart)
{
    extern __shared__ __attribute__((aligned(16))) double red[];  // [4][KC*128] + [4] tt + [4][64] c
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v2f64 rf[KC], zp[KC];
    bool in[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int col = 2 * lane + 128 * k;
        in[k] = col < ldr;
        rf[k] = in[k] ? *reinterpret_cast<const v2f64 *>(rvec + col) : v2f64{0.0, 0.0};
        zp[k] = v2f64{0.0, 0.0};
    }
    double tt = 0.0, cacc = 0.0;
    const int64_t ngroups = (n + R - 1) / R;
    const int64_t gstride = (int64_t)gridDim.x * 4;
    for (int64_t g = (int64_t)blockIdx.x * 4 + wv; g < ngroups; g += gstride) {
        const int64_t row0 = g * R;
        v2f64 x[R][KC];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const bool live = row0 + rr < n;  // wave-uniform
            const v2f64 *rp = reinterpret_cast<const v2f64 *>(Xr + (size_t)(row0 + rr) * (size_t)ldr) + lane;
#pragma unroll
            for (int k = 0; k < KC; ++k) x[rr][k] = (live && in[k]) ? rp[64 * k] : v2f64{0.0, 0.0};
        }
        double tsel = 0.0;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < KC; ++k) s += x[rr][k].x * rf[k].x + x[rr][k].y * rf[k].y;
            const double t = jch_wave_sum(s);
            const bool live = row0 + rr < n;
            const double dt = live ? dw[row0 + rr] * t : 0.0;
            tt += dt * t;
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                zp[k].x += dt * x[rr][k].x;
                zp[k].y += dt * x[rr][k].y;
            }
            if (NIPALS) {
                const double yv = (live && lane < qpad) ? Yr[(size_t)(row0 + rr) * qpad + lane] : 0.0;
                cacc += dt * yv;
            }
            if (lane == rr) tsel = t;
        }
        if (lane < R && row0 + lane < n) tcol[row0 + lane] = tsel;
    }
    // ---- combine the 4 waves of the block in wave order, then one partial row per block
    double *zred = red;                    // [4][KC*128]
    double *tred = red + 4 * KC * 128;     // [4]
    double *cred = tred + 4;               // [4][64]
#pragma unroll
    for (int k = 0; k < KC; ++k)
        *reinterpret_cast<v2f64 *>(zred + wv * (KC * 128) + 2 * lane + 128 * k) = zp[k];
    if (lane == 0) tred[wv] = tt;
    if (NIPALS) cred[wv * 64 + lane] = cacc;
    __syncthreads();
    double *prow = part + (size_t)blockIdx.x * ldpart;
    for (int c = threadIdx.x; c < ldr; c += 256)
        prow[c] = ((zred[c] + zred[KC * 128 + c]) + zred[2 * KC * 128 + c]) + zred[3 * KC * 128 + c];
    if (threadIdx.x == 0) prow[ldr] = ((tred[0] + tred[1]) + tred[2]) + tred[3];
    if (NIPALS && threadIdx.x < qpad)
        prow[ldr + 1 + threadIdx.x] =
            ((cred[threadIdx.x] + cred[64 + threadIdx.x]) + cred[128 + threadIdx.x]) + cred[192 + threadIdx.x];
}

// zt[c] = sum over blocks of part[b][c], fixed order: 4 interleaved streams per column, then combined.
__global__ __launch_bounds__(256) void k_reduce_part(const double *__restrict__ part, int nb, int ldpart, int m,
                                                     double *__restrict__ zt)
{
    __shared__ double sc[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0.0;
    if (c < m)
        for (int b = g; b < nb; b += 4) s += part[(size_t)b * ldpart + c];
    sc[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < m) zt[c] = ((sc[0][cl] + sc[1][cl]) + sc[2][cl]) + sc[3][cl];
}

template <int KC, int R>
static int32_t launch_sweep_t(jch_ctx *ctx, const double *Xr, int64_t n, int ldr, const double *d, const double *rvec,
                              const double *Yr, int qpad, bool nipals, double *tcol, double *zt, int m)
{
    const int64_t ngroups = (n + R - 1) / R;
    const size_t lds = sizeof(double) * (4 * KC * 128 + 4 + 256);
    // persistent-style grid: exactly the blocks the CUs can hold (register/LDS-limited), rows interleaved
    static int occ[2] = {0, 0};
    if (occ[nipals] == 0) {
        int nblk = 0;
        hipError_t e = nipals ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_sweep<KC, R, true>, 256, lds)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_sweep<KC, R, false>, 256, lds);
        occ[nipals] = (e == hipSuccess && nblk > 0) ? nblk : 2;
    }
    int bpc = ctx->sweep_blocks_per_cu > 0 ? ctx->sweep_blocks_per_cu : occ[nipals];
    int64_t nb64 = (ngroups + 3) / 4;
    if (nb64 > (int64_t)ctx->cus * bpc) nb64 = (int64_t)ctx->cus * bpc;
    if (nb64 < 1) nb64 = 1;
    const int nb = (int)nb64;
    const int ldpart = (m + 7) & ~7;
    JCH_TRY(jch_reserve(ctx, ctx->part, sizeof(double) * (size_t)nb * ldpart));
    double *part = (double *)ctx->part.ptr;
    (void)jch_ev(ctx);  // profiling span of the dominant kernel (begin)
    if (nipals)
        hipLaunchKernelGGL((k_sweep<KC, R, true>), dim3(nb), dim3(256), lds, ctx->stream, Xr, n, ldr, d, rvec, Yr, qpad,
                           tcol, part, ldpart);
    else
        hipLaunchKernelGGL((k_sweep<KC, R, false>), dim3(nb), dim3(256), lds, ctx->stream, Xr, n, ldr, d, rvec, Yr, qpad,
                           tcol, part, ldpart);
    (void)jch_ev(ctx);  // (end)
    hipLaunchKernelGGL(k_reduce_part, dim3((m + 63) / 64), dim3(256), 0, ctx->stream, part, nb, ldpart, m, zt);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

int32_t jch_launch_sweep(jch_ctx *ctx, const double *Xr, int64_t n, int p, int ldr, const double *d, const double *rvec,
                         const double *Yr, int qpad, int q_extra, double *tcol, double *zt)
{
    (void)p;
    const bool nip = q_extra > 0;
    const int m = ldr + 1 + (nip ? qpad : 0);
    if (ldr <= 128) return launch_sweep_t<1, 4>(ctx, Xr, n, ldr, d, rvec, Yr, qpad, nip, tcol, zt, m);
    if (ldr <= 256) return launch_sweep_t<2, 4>(ctx, Xr, n, ldr, d, rvec, Yr, qpad, nip, tcol, zt, m);
    if (ldr <= 512) return launch_sweep_t<4, 4>(ctx, Xr, n, ldr, d, rvec, Yr, qpad, nip, tcol, zt, m);
    if (ldr <= 1024) return launch_sweep_t<8, 2>(ctx, Xr, n, ldr, d, rvec, Yr, qpad, nip, tcol, zt, m);
    if (ldr <= 2048) return launch_sweep_t<16, 1>(ctx, Xr, n, ldr, d, rvec, Yr, qpad, nip, tcol, zt, m);
    return jch_fail(ctx, JCH_EINVAL, "fused sweep supports p <= %d (got ld %d)", JCH_SWEEP_MAXP, ldr);
}
