// K6 — NIPALS deflation fused with the next kernel matrix (src/plsnipals.jl:86-87 then :71 of the next LV):
//     X <- X - t zp' ,  Y <- Y - t c' ,  K_next = X' D Y   (from the deflated X, Y)
// One read + one write of the row-major working copy Xr per LV; K_next on v_mfma_f64_16x16x4_f64 exactly as
// in the prologue kernel K2 (prologue.hip).  The reference allocates an n x p outer-product temporary here.
// Bound: HBM, 2*n*ldr*8 bytes per launch.
#include "jch_internal.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));
#define XT_LD 65
#define YT_LD 17

// APPLY: subtract t*zp' / t*c' from what is loaded; STORE: write the deflated X tile back (y group 0 only).
// Yr is never written here (every column tile re-reads it): k_deflate_y runs afterwards.
template <bool APPLY, bool STORE>
__global__ __launch_bounds__(256) void k_deflate_xty(double *__restrict__ Xr, int64_t n, int p, int ldr,
                                                      const double *__restrict__ Yr, int qpad, int q, int yg0,
                                                      const double *__restrict__ d, const double *__restrict__ tcol,
                                                      const double *__restrict__ zpc, double *__restrict__ Kpart,
                                                      int kp_rows)
{
    __shared__ double xt[64 * XT_LD];
    __shared__ double yt[64 * YT_LD];
    __shared__ double tl[64], dl[64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j0 = blockIdx.y * 64;
    const int yg = yg0 + blockIdx.z;
    const int64_t nchunks = (n + 63) / 64;
    const int jcol = j0 + lane;
    const double zpj = (APPLY && jcol < ldr) ? zpc[jcol] : 0.0;
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t i0 = c * 64;
        if (tid < 64) {
            const int64_t i = i0 + tid;
            tl[tid] = (APPLY && i < n) ? tcol[i] : 0.0;
            dl[tid] = i < n ? d[i] : 0.0;
        }
        __syncthreads();
        // Y tile: element e -> (row e>>4, col e&15): row-major Yr, 16 doubles = 128 B per row
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = tid + 256 * k, row = e >> 4, col = e & 15;
            const int yc = yg * 16 + col;
            const int64_t i = i0 + row;
            double v = 0.0;
            if (i < n && yc < q) {
                v = Yr[(size_t)i * qpad + yc];
                if (APPLY) v -= tl[row] * zpc[ldr + yc];
            }
            yt[row * YT_LD + col] = dl[row] * v;
        }
        // X tile: row-major, (row wv+4k, col lane): 512 B per wave-instruction
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int row = wv + 4 * k;
            const int64_t i = i0 + row;
            double v = 0.0;
            if (i < n && jcol < ldr) {
                v = Xr[(size_t)i * ldr + jcol];
                if (APPLY) {
                    v -= tl[row] * zpj;
                    if (STORE) Xr[(size_t)i * ldr + jcol] = v;
                }
            }
            xt[row * XT_LD + lane] = v;
        }
        __syncthreads();
        if (Kpart) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int row = 4 * kk + (lane >> 4);
                const double a = xt[row * XT_LD + 16 * wv + (lane & 15)];
                const double b = yt[row * YT_LD + (lane & 15)];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (Kpart) {
        double *kp = Kpart + ((size_t)blockIdx.x * kp_rows) * qpad;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int j = j0 + 16 * wv + (lane >> 4) + 4 * reg;
            if (j < kp_rows) kp[(size_t)j * qpad + yg * 16 + (lane & 15)] = acc[reg];
        }
    }
}

__global__ __launch_bounds__(256) void k_deflate_y(double *__restrict__ Yr, int64_t n, int qpad, int q,
                                                   const double *__restrict__ tcol, const double *__restrict__ cvec)
{
    const int64_t total = n * qpad;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e / qpad;
        const int k = (int)(e % qpad);
        if (k < q) Yr[e] -= tcol[i] * cvec[k];
    }
}

__global__ __launch_bounds__(256) void k_reduce_kpart2(const double *__restrict__ Kpart, int nbx, int kp_rows, int p, int qpad,
                                                       double *__restrict__ K)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= p * qpad) return;
    const size_t stride = (size_t)kp_rows * qpad;
    double s = 0.0;
    for (int b = 0; b < nbx; ++b) s += Kpart[(size_t)b * stride + e];
    K[e] = s;
}

int32_t jch_launch_deflate(jch_ctx *ctx, double *Xr, int64_t n, int p, int ldr, double *Yr, int qpad, int q,
                           const double *d, const double *tcol, const double *zpc, double *Knext)
{
    const int ptiles = (ldr + 63) / 64, kp_rows = ptiles * 64, ygroups = qpad / 16;
    const int64_t nchunks = (n + 63) / 64;
    int nbx = (ctx->cus * 3 + ptiles - 1) / ptiles;
    if (nbx > nchunks) nbx = (int)(nchunks > 0 ? nchunks : 1);
    double *Kpart = nullptr;
    if (Knext) {
        JCH_TRY(jch_reserve(ctx, ctx->kpart, sizeof(double) * (size_t)nbx * kp_rows * qpad));
        Kpart = (double *)ctx->kpart.ptr;
    }
    (void)jch_ev(ctx);
    // pass A: deflate X in place (+ K columns of y group 0, from y deflated on the fly)
    hipLaunchKernelGGL((k_deflate_xty<true, true>), dim3(nbx, ptiles, 1), dim3(256), 0, ctx->stream, Xr, n, p, ldr, Yr, qpad,
                       q, 0, d, tcol, zpc, Kpart, kp_rows);
    (void)jch_ev(ctx);
    const int nby = (int)std::min<int64_t>((n * qpad + 255) / 256, (int64_t)ctx->cus * 8);
    hipLaunchKernelGGL(k_deflate_y, dim3(nby > 0 ? nby : 1), dim3(256), 0, ctx->stream, Yr, n, qpad, q, tcol, zpc + ldr);
    if (Knext && ygroups > 1)  // pass B: remaining y groups from the already deflated X, Y
        hipLaunchKernelGGL((k_deflate_xty<false, false>), dim3(nbx, ptiles, ygroups - 1), dim3(256), 0, ctx->stream, Xr, n, p,
                           ldr, Yr, qpad, q, 1, d, tcol, zpc, Kpart, kp_rows);
    if (Knext) {
        hipLaunchKernelGGL(k_reduce_kpart2, dim3((p * qpad + 255) / 256), dim3(256), 0, ctx->stream, Kpart, nbx, kp_rows, p,
                           qpad, Knext);
        JCH_TRY(jch_allreduce_f64(ctx, Knext, (size_t)p * qpad));
    }
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}
