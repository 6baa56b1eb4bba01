// K6 — NIPALS deflation fused with the next kernel matrix (src/plsnipals.jl:86-87 then :71 of the next LV):
//     X <- X - t zp' ,  Y <- Y - t c' ,  K_next = X' D Y   (from the deflated X, Y)
// One read + one write of the row-major working copy Xr per LV; K_next on v_mfma_f64_16x16x4_f64 exactly as
// in the prologue kernel K2 (prologue.hip).  The reference allocates an n x p outer-product temporary here.
// Bound: HBM, 2*n*ldr*8 bytes per launch.
#include <stdlib.h>

#include <algorithm>

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

// ---- streaming variant for q <= 4 (cfg4: q = 1): same register-resident row layout as the sweep (sweep.hip) —
// no LDS transpose, 16-B coalesced loads and stores, K_next accumulated like zp.  Per-block partials
// part[b][k * ldr + c] are summed in fixed order by k_reduce_kstream (deterministic).
typedef double v2f64 __attribute__((ext_vector_type(2)));
template <int KC, int R, int Q, bool PF>
__global__ __launch_bounds__(256) void k_deflate_stream(double *__restrict__ Xr, int64_t n, int ldr, double *__restrict__ Yr,
                                                         int qpad, const double *__restrict__ dw,
                                                         const double *__restrict__ tcol, const double *__restrict__ zpc,
                                                         double *__restrict__ part, int ldpart, int plain_stores)
{
    extern __shared__ __attribute__((aligned(16))) double red[];  // [4][Q][KC*128]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v2f64 zpf[KC], kacc[Q][KC];
    bool in[KC];
    double cf[Q];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int col = 2 * lane + 128 * k;
        in[k] = col < ldr;
        zpf[k] = in[k] ? *reinterpret_cast<const v2f64 *>(zpc + col) : v2f64{0.0, 0.0};
#pragma unroll
        for (int y = 0; y < Q; ++y) kacc[y][k] = v2f64{0.0, 0.0};
    }
#pragma unroll
    for (int y = 0; y < Q; ++y) cf[y] = zpc[ldr + y];
    const int64_t ngroups = (n + R - 1) / R;
    const int64_t gstride = (int64_t)gridDim.x * 4;
    // software prefetch: rows, scores and weights of the wave's next group are requested before the current group is
    // rewritten (JCH_DEFLATE_PF=0 restores the plain loop)
    v2f64 xn[R][KC];
    double tn[R], dn_[R];
    auto fetch = [&](int64_t gg) {
        const int64_t r0 = gg * R;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const bool live = r0 + rr < n;
            const v2f64 *rp = reinterpret_cast<const v2f64 *>(Xr + (size_t)(r0 + rr) * (size_t)ldr) + lane;
#pragma unroll
            for (int k = 0; k < KC; ++k)
                xn[rr][k] = (live && in[k]) ? __builtin_nontemporal_load(rp + 64 * k) : v2f64{0.0, 0.0};
            tn[rr] = live ? tcol[r0 + rr] : 0.0;
            dn_[rr] = live ? dw[r0 + rr] : 0.0;
        }
    };
    int64_t g = (int64_t)blockIdx.x * 4 + wv;
    if (g < ngroups) fetch(g);
    for (; g < ngroups; g += gstride) {
        const int64_t row0 = g * R;
        v2f64 x[R][KC];
        double tc[R], dc[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            tc[rr] = tn[rr]; dc[rr] = dn_[rr];
#pragma unroll
            for (int k = 0; k < KC; ++k) x[rr][k] = xn[rr][k];
        }
        if (PF && g + gstride < ngroups) fetch(g + gstride);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int64_t row = row0 + rr;
            if (row < n) {   // wave-uniform
                const double t = tc[rr], dv = dc[rr];
                v2f64 *wp = reinterpret_cast<v2f64 *>(Xr + (size_t)row * (size_t)ldr) + lane;
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    x[rr][k].x -= t * zpf[k].x;
                    x[rr][k].y -= t * zpf[k].y;
                    if (in[k]) { if (plain_stores) wp[64 * k] = x[rr][k]; else __builtin_nontemporal_store(x[rr][k], wp + 64 * k); }
                }
#pragma unroll
                for (int y = 0; y < Q; ++y) {
                    const double yn = Yr[(size_t)row * qpad + y] - t * cf[y];
                    if (lane == y) Yr[(size_t)row * qpad + y] = yn;
                    const double s = dv * yn;
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        kacc[y][k].x += s * x[rr][k].x;
                        kacc[y][k].y += s * x[rr][k].y;
                    }
                }
            }
        }
        if (!PF && g + gstride < ngroups) fetch(g + gstride);
    }
    double *prow = part + (size_t)blockIdx.x * ldpart;
#pragma unroll
    for (int y = 0; y < Q; ++y) {
#pragma unroll
        for (int k = 0; k < KC; ++k)
            *reinterpret_cast<v2f64 *>(red + (size_t)(wv * Q + y) * (KC * 128) + 2 * lane + 128 * k) = kacc[y][k];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < Q * ldr; e += 256) {
        const int y = e / ldr, c = e - y * ldr;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(size_t)(w * Q + y) * (KC * 128) + c];
        prow[e] = s;
    }
}

// K[c][y] = sum_b part[b][y*ldr + c]   (fixed order: 16 interleaved block streams, then combined)
__global__ __launch_bounds__(1024) void k_reduce_kstream(const double *__restrict__ part, int nb, int ldpart, int ldr, int p,
                                                         int Q, int qpad, double *__restrict__ K)
{
    __shared__ double sc[16][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + cl;     // index into [Q][ldr]
    double s = 0.0;
    if (e < Q * ldr)
        for (int b = g; b < nb; b += 16) s += part[(size_t)b * ldpart + e];
    sc[g][cl] = s;
    __syncthreads();
    if (g == 0 && e < Q * ldr) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sc[k][cl];
        const int y = e / ldr, c = e - y * ldr;
        if (c < p) K[(size_t)c * qpad + y] = t;
    }
}

template <int KC, int R, int Q, bool PF>
static int32_t launch_deflate_stream_pf(jch_ctx *ctx, double *Xr, int64_t n, int p, int ldr, double *Yr, int qpad, const double *d,
                                     const double *tcol, const double *zpc, double *Knext)
{
    const size_t lds = sizeof(double) * 4 * Q * KC * 128;
    static int bpc = 0;
    static jch_per_device_once occ_once;
    if (!occ_once.done(ctx->device)) {
        int nblk = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_deflate_stream<KC, R, Q, PF>, 256, lds);
        bpc = (e == hipSuccess && nblk > 0) ? nblk : 2;
        if (lds > 64 * 1024)
            JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_deflate_stream<KC, R, Q, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        occ_once.mark(ctx->device);
    }
    const int64_t ngroups = (n + R - 1) / R;
    // (measurement knobs, read per call: JCH_DEFLATE_BPC blocks per CU instead of the occupancy maximum, JCH_DEFLATE_NT=0 plain stores)
    const char *e_bpc = getenv("JCH_DEFLATE_BPC"), *e_nt = getenv("JCH_DEFLATE_NT");
    const int use_bpc = (e_bpc && atoi(e_bpc) > 0) ? std::min(atoi(e_bpc), bpc) : bpc;
    const int plain_stores = (e_nt && atoi(e_nt) == 0) ? 1 : 0;
    int64_t nb64 = std::min<int64_t>((ngroups + 3) / 4, (int64_t)ctx->cus * use_bpc);
    const int nb = (int)std::max<int64_t>(nb64, 1);
    const int ldpart = (Q * ldr + 7) & ~7;
    JCH_TRY(jch_reserve(ctx, ctx->kpart, sizeof(double) * (size_t)nb * ldpart));
    double *part = (double *)ctx->kpart.ptr;
    (void)jch_ev(ctx);
    hipLaunchKernelGGL((k_deflate_stream<KC, R, Q, PF>), dim3(nb), dim3(256), lds, ctx->stream, Xr, n, ldr, Yr, qpad, d, tcol, zpc,
                       part, ldpart, plain_stores);
    (void)jch_ev(ctx);
    if (Knext) {
        // pad columns of K (y >= q) must stay zero: they are never written here and were zeroed by the prologue
        hipLaunchKernelGGL(k_reduce_kstream, dim3((Q * ldr + 63) / 64), dim3(1024), 0, ctx->stream, part, nb, ldpart, ldr, p, Q,
                           qpad, Knext);
        JCH_TRY(jch_allreduce_f64(ctx, Knext, (size_t)p * qpad));
    }
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

template <int KC, int R, int Q>
static int32_t launch_deflate_stream(jch_ctx *ctx, double *Xr, int64_t n, int p, int ldr, double *Yr, int qpad, const double *d,
                                     const double *tcol, const double *zpc, double *Knext)
{
    static int pf = -1;
    if (pf < 0) { const char *e = getenv("JCH_DEFLATE_PF"); pf = e ? atoi(e) : 1; }
    if (pf) return launch_deflate_stream_pf<KC, R, Q, true>(ctx, Xr, n, p, ldr, Yr, qpad, d, tcol, zpc, Knext);
    return launch_deflate_stream_pf<KC, R, Q, false>(ctx, Xr, n, p, ldr, Yr, qpad, d, tcol, zpc, Knext);
}

// ---- postponed write-back (see k_sweep_lazy, sweep.hip): the same pass as k_deflate_stream, but the rows in memory are
// `npend - 1` deflations behind.  All pending corrections (oldest first; the newest is this LV's) are applied in registers
// with the eager kernel's own expression, K_next is accumulated from the result, Y is deflated eagerly (128 B per row),
// and the rows are stored only when `flush` — every m-th LV.  HBM bytes per LV: n*ldr*8 read + n*ldr*8/m written.
__device__ __forceinline__ double jch_kp_readlane(double v, int srclane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), srclane), __builtin_amdgcn_readlane(__double2loint(v), srclane));
}
template <int KC, int R, int Q>
__global__ __launch_bounds__(256) void k_kpass_lazy(double *__restrict__ Xr, int64_t n, int ldr, double *__restrict__ Yr, int qpad,
                                                    const double *__restrict__ dw, const double *__restrict__ pend_p, int npend,
                                                    const double *__restrict__ tpend, int64_t tstride,
                                                    const double *__restrict__ cvec, int flush,
                                                    double *__restrict__ part, int ldpart)
{
    extern __shared__ __attribute__((aligned(16))) double red[];  // loop: [npend][KC*128]; end: [4][Q][KC*128]
    constexpr int LDP = KC * 128;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v2f64 kacc[Q][KC];
    int coff[KC];
    double cf[Q];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int col = 2 * lane + 128 * k;
        coff[k] = col < ldr ? col : ldr - 2;
#pragma unroll
        for (int y = 0; y < Q; ++y) kacc[y][k] = v2f64{0.0, 0.0};
    }
    const int64_t ngroups = (n + R - 1) / R;
    const int64_t gstride = (int64_t)gridDim.x * 4;
    const int pl = min(lane, npend * R - 1);                            // npend >= 1 here
    const int64_t pk_off = (int64_t)(pl / R) * tstride;
    const int pr = pl % R;
    v2f64 xn[R][KC];
    double dn_[R], tpn = 0.0;
    auto fetch = [&](int64_t gg) {
        const int64_t r0 = gg * R;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int64_t row = r0 + rr < n ? r0 + rr : n - 1;
            const double *rp = Xr + (size_t)row * (size_t)ldr;
#pragma unroll
            for (int k = 0; k < KC; ++k) xn[rr][k] = __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(rp + coff[k]));
            dn_[rr] = dw[row];
        }
        tpn = tpend[pk_off + (r0 + pr < n ? r0 + pr : n - 1)];
    };
    int64_t g = (int64_t)blockIdx.x * 4 + wv;
    if (g < ngroups) fetch(g);
    for (int e = threadIdx.x; e < npend * LDP; e += 256) red[e] = pend_p[e];
#pragma unroll
    for (int y = 0; y < Q; ++y) cf[y] = cvec[y];
    __syncthreads();
    for (; g < ngroups; g += gstride) {
        const int64_t row0 = g * R;
        v2f64 x[R][KC];
        double dc[R];
        const double tpc = tpn;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            dc[rr] = dn_[rr];
#pragma unroll
            for (int k = 0; k < KC; ++k) x[rr][k] = xn[rr][k];
        }
        if (g + gstride < ngroups) fetch(g + gstride);
        double tk[R];
        for (int k = 0; k < npend; ++k) {
#pragma unroll
            for (int rr = 0; rr < R; ++rr) tk[rr] = jch_kp_readlane(tpc, k * R + rr);
            const double *pl_k = red + k * LDP + 2 * lane;
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                const v2f64 pf = *reinterpret_cast<const v2f64 *>(pl_k + 128 * kk);
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    x[rr][kk].x -= tk[rr] * pf.x;
                    x[rr][kk].y -= tk[rr] * pf.y;
                }
            }
        }
        // (tk now holds this LV's scores: the newest pending entry)
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int64_t row = row0 + rr;
            if (row < n) {   // wave-uniform
                if (flush) {
                    double *wp = Xr + (size_t)row * (size_t)ldr;
#pragma unroll
                    for (int k = 0; k < KC; ++k)
                        if (2 * lane + 128 * k < ldr) __builtin_nontemporal_store(x[rr][k], reinterpret_cast<v2f64 *>(wp + 2 * lane + 128 * k));
                }
#pragma unroll
                for (int y = 0; y < Q; ++y) {
                    const double yn = Yr[(size_t)row * qpad + y] - tk[rr] * cf[y];
                    if (lane == y) Yr[(size_t)row * qpad + y] = yn;
                    const double s = dc[rr] * yn;
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        kacc[y][k].x += s * x[rr][k].x;
                        kacc[y][k].y += s * x[rr][k].y;
                    }
                }
            }
        }
    }
    __syncthreads();                            // pending loadings dead: the area becomes the combine buffer
    double *prow = part + (size_t)blockIdx.x * ldpart;
#pragma unroll
    for (int y = 0; y < Q; ++y) {
#pragma unroll
        for (int k = 0; k < KC; ++k)
            *reinterpret_cast<v2f64 *>(red + (size_t)(wv * Q + y) * (KC * 128) + 2 * lane + 128 * k) = kacc[y][k];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < Q * ldr; e += 256) {
        const int y = e / ldr, c = e - y * ldr;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(size_t)(w * Q + y) * (KC * 128) + c];
        prow[e] = s;
    }
}

template <int KC, int R, int Q>
static int32_t launch_kpass_lazy_t(jch_ctx *ctx, double *Xr, int64_t n, int p, int ldr, double *Yr, int qpad, const double *d,
                                   const double *pend_p, int npend, int npend_max, const double *tpend, int64_t tstride,
                                   const double *cvec, bool flush, double *Knext)
{
    const size_t lds_red = sizeof(double) * 4 * Q * KC * 128;
    const size_t lds = std::max(lds_red, sizeof(double) * (size_t)npend_max * KC * 128);
    static int bpc = 0;
    static jch_per_device_once once;
    if (!once.done(ctx->device)) {
        JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_kpass_lazy<KC, R, Q>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int nblk = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_kpass_lazy<KC, R, Q>, 256, lds_red);
        bpc = (e == hipSuccess && nblk > 0) ? nblk : 1;
        once.mark(ctx->device);
    }
    if (lds > 160 * 1024) return jch_fail(ctx, JCH_EINVAL, "internal: lazy NIPALS pass: %zu bytes of LDS", lds);
    const char *e_bpc = getenv("JCH_DEFLATE_BPC");
    int use_bpc = (e_bpc && atoi(e_bpc) > 0) ? std::min(atoi(e_bpc), bpc) : bpc;
    use_bpc = std::max(1, std::min<int>(use_bpc, (int)((160 * 1024) / lds)));
    const int64_t ngroups = (n + R - 1) / R;
    const int nb = (int)std::max<int64_t>(std::min<int64_t>((ngroups + 3) / 4, (int64_t)ctx->cus * use_bpc), 1);
    const int ldpart = (Q * ldr + 7) & ~7;
    JCH_TRY(jch_reserve(ctx, ctx->kpart, sizeof(double) * (size_t)nb * ldpart));
    double *part = (double *)ctx->kpart.ptr;
    (void)jch_ev(ctx);
    hipLaunchKernelGGL((k_kpass_lazy<KC, R, Q>), dim3(nb), dim3(256), lds, ctx->stream, Xr, n, ldr, Yr, qpad, d, pend_p, npend, tpend,
                       tstride, cvec, flush ? 1 : 0, part, ldpart);
    (void)jch_ev(ctx);
    if (Knext) {
        hipLaunchKernelGGL(k_reduce_kstream, dim3((Q * ldr + 63) / 64), dim3(1024), 0, ctx->stream, part, nb, ldpart, ldr, p, Q, qpad, Knext);
        JCH_TRY(jch_allreduce_f64(ctx, Knext, (size_t)p * qpad));
    }
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

// ---- the same postponed write-back for 4 < q <= 16 (and q = 3, 4 at the widest rows): the MFMA tile kernel above with up
// to MP pending corrections — lane `jcol` keeps its MP loadings in registers, the scores of the 64-row chunk sit in LDS.
// Y is read with this LV's step applied on the fly and rewritten by k_deflate_y afterwards, as in the eager path.
#define JCH_TILE_MP 8
__global__ __launch_bounds__(256) void k_kpass_tile_lazy(double *__restrict__ Xr, int64_t n, int ldr, const double *__restrict__ Yr, int qpad,
                                                         int q, const double *__restrict__ d, const double *__restrict__ pend_p, int ldp,
                                                         int npend, const double *__restrict__ tpend, int64_t tstride,
                                                         const double *__restrict__ cvec, int flush, double *__restrict__ Kpart, int kp_rows)
{
    __shared__ double xt[64 * XT_LD];
    __shared__ double yt[64 * YT_LD];
    __shared__ double tl[JCH_TILE_MP][64], dl[64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j0 = blockIdx.y * 64;
    const int64_t nchunks = (n + 63) / 64;
    const int jcol = j0 + lane;
    double zpj[JCH_TILE_MP];
#pragma unroll
    for (int k = 0; k < JCH_TILE_MP; ++k) zpj[k] = (k < npend && jcol < ldr) ? pend_p[(size_t)k * ldp + jcol] : 0.0;
    const int knew = npend - 1;
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t i0 = c * 64;
        {   // scores of the pending LVs and the weights of this chunk: thread (k = wv + 4 j, row = lane)
            const int64_t i = i0 + lane;
#pragma unroll
            for (int k = wv; k < JCH_TILE_MP; k += 4) tl[k][lane] = (k < npend && i < n) ? tpend[(size_t)k * tstride + i] : 0.0;
            if (wv == 3) dl[lane] = i < n ? d[i] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = tid + 256 * k, row = e >> 4, col = e & 15;
            const int64_t i = i0 + row;
            double v = 0.0;
            if (i < n && col < q) v = Yr[(size_t)i * qpad + col] - tl[knew][row] * cvec[col];
            yt[row * YT_LD + col] = dl[row] * v;
        }
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int row = wv + 4 * k;
            const int64_t i = i0 + row;
            double v = 0.0;
            if (i < n && jcol < ldr) {
                v = Xr[(size_t)i * ldr + jcol];
#pragma unroll
                for (int kk = 0; kk < JCH_TILE_MP; ++kk)
                    if (kk < npend) v -= tl[kk][row] * zpj[kk];     // wave-uniform; oldest first
                if (flush) Xr[(size_t)i * ldr + jcol] = v;
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
            if (j < kp_rows) kp[(size_t)j * qpad + (lane & 15)] = acc[reg];
        }
    }
}

// K[e] = sum_b Kpart[b][e], e < m (fixed order: 16 interleaved block streams, then combined)
__global__ __launch_bounds__(1024) void k_reduce_kpart3(const double *__restrict__ Kpart, int nb, int stride, int m, double *__restrict__ K)
{
    __shared__ double sc[16][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + cl;
    double s = 0.0;
    if (e < m)
        for (int b = g; b < nb; b += 16) s += Kpart[(size_t)b * stride + e];
    sc[g][cl] = s;
    __syncthreads();
    if (g == 0 && e < m) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sc[k][cl];
        K[e] = t;
    }
}

// ---- the MFMA pass without the LDS transpose (4 < q <= 16, p <= 1024).  v_mfma_f64_16x16x4 takes A[m = lane & 15][k = lane >> 4]:
// with k = row and m = column, lane l can load its operand straight from the row-major copy — a 16-B load of row
// i0 + (l >> 4), columns c, c + 1 with c = 32 s + 2 (l & 15) feeds TWO products (the even and the odd columns of the
// 32-column span s are two 16-row "strips" of K); 16 lanes read 256 contiguous bytes, one instruction 4 rows.  The four
// waves of a block share the rows of a chunk and split its spans (s = wave + 4 u), so the B operand d_i * y_i (with this
// LV's Y step applied) and the pending scores are staged ONCE per chunk in LDS, double-buffered: one barrier per chunk,
// nothing else.  Pending loadings of the lane's 2 NS columns sit in registers.  X is prefetched one chunk ahead.
// NW (round 4): waves per block.  NW = 8 with half the spans per wave (NS = 2 at p <= 512) keeps a wave's state inside 256
// registers, so TWO waves share a SIMD: one wave's products and pending corrections run while the other waits for its rows (with
// one wave per SIMD at 337 registers 56 % of the wave cycles were issue stalls and nothing else was resident to use them).
template <int NS, int STEPS, int MP, int MINB, int NW>
__global__ __launch_bounds__(64 * NW, MINB) void k_kpass_mfma_lazy(double *__restrict__ Xr, int64_t n, int ldr, double *__restrict__ Yr, int q,
                                                         const double *__restrict__ dw, const double *__restrict__ pend_p, int ldp, int npend,
                                                         const double *__restrict__ tpend, int64_t tstride, const double *__restrict__ cvec,
                                                         int flush, double *__restrict__ Kpart, int kp_rows)
{
    constexpr int RB = 4 * STEPS;                       // rows per chunk
    __shared__ double bt[2][RB][16];                    // d_i * (y_i - t_i c')
    __shared__ double tl[2][MP][RB];                    // pending scores of the chunk's rows
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    int coff[NS];
    bool in[NS];
    double pk[MP][NS][2];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int col = 32 * (wv + NW * u) + 2 * c16;
        in[u] = col < ldr;
        coff[u] = in[u] ? col : ldr - 2;
#pragma unroll
        for (int k = 0; k < MP; ++k) {
            pk[k][u][0] = (k < npend && in[u]) ? pend_p[(size_t)k * ldp + col] : 0.0;
            pk[k][u][1] = (k < npend && in[u]) ? pend_p[(size_t)k * ldp + col + 1] : 0.0;
        }
    }
    v4f64 acc[NS][2];
#pragma unroll
    for (int u = 0; u < NS; ++u) { acc[u][0] = v4f64{0.0, 0.0, 0.0, 0.0}; acc[u][1] = v4f64{0.0, 0.0, 0.0, 0.0}; }
    const int64_t nchunks = (n + RB - 1) / RB;
    const int knew = npend - 1;
    // staging roles: thread (row = tid >> 4, y = tid & 15) builds the B entry, thread (k = tid >> 4, row = tid & 15) a score
    // (with NW = 8 the second 256 threads have no staging role: their srow / tk fall outside RB / npend <= MP <= 16 rows ... 15)
    const int srow = tid < 256 ? tid >> 4 : RB, sy = tid & 15;
    const double cy = sy < q ? cvec[sy] : 0.0;
    const int tk = tid < 256 ? tid >> 4 : MP, trow = tid & 15;
    // X is prefetched one whole chunk ahead.  (Rotating two buffers and re-requesting every consumed step for chunk c + 2 G
    // — 1.5 chunks in flight — was measured SLOWER for the read-only pass, 968 against 800 us at cfg2 shape: the loads then
    // queue behind each step's eight 64-cycle MFMAs instead of going out ahead of all of them.)
    v2f64 xn[STEPS][NS];
    double s_y = 0.0, s_t = 0.0, s_d = 0.0, s_tk = 0.0;
    int64_t s_row = -1;
    auto fetch = [&](int64_t c) {
        const int64_t i0 = c * RB;
        // the staging operands FIRST: vmcnt retires in order, so waiting for them at the end of the chunk must not mean
        // waiting for the whole prefetched chunk behind which they would otherwise queue
        if (srow < RB) {
            const int64_t r = i0 + srow;
            const bool live = r < n;
            s_row = live ? r : -1;
            s_y = live ? Yr[(size_t)r * 16 + sy] : 0.0;
            s_t = live ? tpend[(size_t)knew * tstride + r] : 0.0;
            s_d = live ? dw[r] : 0.0;
        }
        if (tk < npend && trow < RB) {
            const int64_t r = i0 + trow;
            s_tk = r < n ? tpend[(size_t)tk * tstride + r] : 0.0;
        }
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int64_t r = i0 + 4 * st + g;
            const double *rp = Xr + (size_t)(r < n ? r : n - 1) * (size_t)ldr;
#pragma unroll
            for (int u = 0; u < NS; ++u) xn[st][u] = __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(rp + coff[u]));
        }
    };
    // (every Y element of a chunk is read by exactly one thread of exactly one block: that thread also stores the deflated value)
    auto stage = [&](int buf) {
        if (srow < RB) {
            const double yd = s_y - s_t * cy;
            bt[buf][srow][sy] = s_d * yd;
            if (s_row >= 0 && sy < q) Yr[(size_t)s_row * 16 + sy] = yd;
        }
        if (tk < npend && trow < RB) tl[buf][tk][trow] = s_tk;
    };
    int64_t c = blockIdx.x;
    int cur = 0;
    if (c < nchunks) { fetch(c); stage(0); }
    __syncthreads();
    for (; c < nchunks; c += gridDim.x) {
        const int64_t i0 = c * RB;
        v2f64 x[STEPS][NS];
#pragma unroll
        for (int st = 0; st < STEPS; ++st)
#pragma unroll
            for (int u = 0; u < NS; ++u) x[st][u] = xn[st][u];
        const bool more = c + gridDim.x < nchunks;
        if (more) fetch(c + gridDim.x);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int lr = 4 * st + g;
            const int64_t r = i0 + lr;
#pragma unroll
            for (int k = 0; k < MP; ++k) {
                if (k < npend) {                        // block-uniform; oldest first
                    const double t = tl[cur][k][lr];
#pragma unroll
                    for (int u = 0; u < NS; ++u) {
                        x[st][u].x -= t * pk[k][u][0];
                        x[st][u].y -= t * pk[k][u][1];
                    }
                }
            }
            const double b = bt[cur][lr][c16];
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[st][u].x, b, acc[u][0], 0, 0, 0);
                acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[st][u].y, b, acc[u][1], 0, 0, 0);
            }
            if (flush && r < n) {
                double *wp = Xr + (size_t)r * (size_t)ldr;
#pragma unroll
                for (int u = 0; u < NS; ++u)
                    if (in[u]) __builtin_nontemporal_store(x[st][u], reinterpret_cast<v2f64 *>(wp + coff[u]));
            }
        }
        if (more) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    if (Kpart) {   // D[m = 4 reg + g][n = c16]: K row = column 32 s + 2 m + h of X, K column = y
        double *kp = Kpart + (size_t)blockIdx.x * kp_rows * 16;
#pragma unroll
        for (int u = 0; u < NS; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int j = 32 * (wv + NW * u) + 2 * (4 * reg + g) + h;
                    if (j < kp_rows) kp[(size_t)j * 16 + c16] = acc[u][h][reg];
                }
    }
}

template <int NS, int STEPS, int MP, int MINB = 1, int NW = 4>
static int32_t launch_kpass_mfma_lazy_t(jch_ctx *ctx, double *Xr, int64_t n, int p, int ldr, double *Yr, int q, const double *d,
                                        const double *pend_p, int npend, const double *tpend, int64_t tstride, const double *cvec,
                                        bool flush, double *Knext)
{
    constexpr int RB = 4 * STEPS;
    if (npend > MP) return jch_fail(ctx, JCH_EINVAL, "internal: lazy MFMA pass: %d pending corrections, at most %d", npend, MP);
    const int64_t nchunks = (n + RB - 1) / RB;
    const char *e_bpc = getenv("JCH_DEFLATE_BPC");
    const int bpc = (e_bpc && atoi(e_bpc) > 0) ? atoi(e_bpc) : MINB;
    const int nb = (int)std::max<int64_t>(std::min<int64_t>(nchunks, (int64_t)ctx->cus * bpc), 1);
    const int kp_rows = ldr;
    JCH_TRY(jch_reserve(ctx, ctx->kpart, sizeof(double) * (size_t)nb * kp_rows * 16));
    double *Kpart = (double *)ctx->kpart.ptr;
    (void)jch_ev(ctx);
    hipLaunchKernelGGL((k_kpass_mfma_lazy<NS, STEPS, MP, MINB, NW>), dim3(nb), dim3(64 * NW), 0, ctx->stream, Xr, n, ldr, Yr, q, d, pend_p,
                       jch_nipals_lazy_pitch(ldr), npend, tpend, tstride, cvec, flush ? 1 : 0, Knext ? Kpart : nullptr, kp_rows);
    (void)jch_ev(ctx);
    if (Knext) {
        hipLaunchKernelGGL(k_reduce_kpart3, dim3((p * 16 + 63) / 64), dim3(1024), 0, ctx->stream, Kpart, nb, kp_rows * 16, p * 16, Knext);
        JCH_TRY(jch_allreduce_f64(ctx, Knext, (size_t)p * 16));
    }
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

static bool lazy_stream_shape(int ldr, int q) { return q >= 1 && q <= 4 && ldr <= 2048 && !(ldr > 1024 && q > 2); }

static int32_t launch_kpass_tile_lazy(jch_ctx *ctx, double *Xr, int64_t n, int p, int ldr, double *Yr, int qpad, int q, const double *d,
                                      const double *pend_p, int npend, const double *tpend, int64_t tstride, const double *cvec, bool flush,
                                      double *Knext)
{
    const int ptiles = (ldr + 63) / 64, kp_rows = ptiles * 64;
    const int64_t nchunks = (n + 63) / 64;
    int nbx = (ctx->cus * 3 + ptiles - 1) / ptiles;
    if (nbx > nchunks) nbx = (int)(nchunks > 0 ? nchunks : 1);
    double *Kpart = nullptr;
    if (Knext) {
        JCH_TRY(jch_reserve(ctx, ctx->kpart, sizeof(double) * (size_t)nbx * kp_rows * qpad));
        Kpart = (double *)ctx->kpart.ptr;
    }
    (void)jch_ev(ctx);
    hipLaunchKernelGGL(k_kpass_tile_lazy, dim3(nbx, ptiles, 1), dim3(256), 0, ctx->stream, Xr, n, ldr, Yr, qpad, q, d, pend_p,
                       jch_nipals_lazy_pitch(ldr), npend, tpend, tstride, cvec, flush ? 1 : 0, Kpart, kp_rows);
    (void)jch_ev(ctx);
    const int nby = (int)std::min<int64_t>((n * qpad + 255) / 256, (int64_t)ctx->cus * 8);
    hipLaunchKernelGGL(k_deflate_y, dim3(nby > 0 ? nby : 1), dim3(256), 0, ctx->stream, Yr, n, qpad, q, tpend + (size_t)(npend - 1) * tstride, cvec);
    if (Knext) {
        hipLaunchKernelGGL(k_reduce_kpart2, dim3((p * qpad + 255) / 256), dim3(256), 0, ctx->stream, Kpart, nbx, kp_rows, p, qpad, Knext);
        JCH_TRY(jch_allreduce_f64(ctx, Knext, (size_t)p * qpad));
    }
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

int32_t jch_launch_kpass_lazy(jch_ctx *ctx, double *Xr, int64_t n, int p, int ldr, double *Yr, int qpad, int q, const double *d,
                              const double *pend_p, int npend, int npend_max, const double *tpend, int64_t tstride,
                              const double *cvec, bool flush, double *Knext)
{
    if (npend < 1 || npend > npend_max || npend_max > jch_nipals_lazy_capacity(ldr, q))
        return jch_fail(ctx, JCH_EINVAL, "internal: lazy NIPALS pass: bad pending count");
    if (!lazy_stream_shape(ldr, q)) {
        if (qpad == 16 && ldr <= 1024 && !getenv("JCH_KPASS_TILE")) {   // (JCH_KPASS_TILE=1: the LDS-transposing tile kernel instead)
#define JCH_KM(NS, STEPS, MP, ...) return launch_kpass_mfma_lazy_t<NS, STEPS, MP, ##__VA_ARGS__>(ctx, Xr, n, p, ldr, Yr, q, d, pend_p, npend, tpend, tstride, cvec, flush, Knext)
            if (ldr <= 128) JCH_KM(1, 4, 6);
            if (ldr <= 256) JCH_KM(2, 4, 6);
            // (measured at cfg2 shape, read-only pass: 16-row chunks 800-845 us; 32-row chunks with 2 loadings 802; 8-row chunks
            // 914, with two blocks per CU and 3 loadings 745 — but rewriting every 3rd LV instead of every 6th costs as much;
            // 4-row chunks with two blocks per CU 896)
            // round 4, measured and NOT the default (JCH_KPASS_NW=8): eight waves per block, two per SIMD, half the spans per wave
            // (200 registers instead of 337).  plsnipals q = 10 at cfg2 shape: 548.5 LV/s against 550.8 with four waves (pass 1.682
            // against 1.675 ms per LV), plswold 523.7 against 534.9 — a second resident wave does not fill the issue stalls: the f64
            // products and the vector work of BOTH waves queue for the same SIMD (the pipe does not overlap them), and
            // the bytes in flight per CU are unchanged.
            const char *e_nw = getenv("JCH_KPASS_NW");
            const bool nw8 = e_nw && atoi(e_nw) == 8;
            if (ldr <= 512) { if (nw8) JCH_KM(2, 4, 6, 1, 8); JCH_KM(4, 4, 6); }
            if (nw8) JCH_KM(4, 2, 3, 1, 8);
            JCH_KM(8, 2, 3);
#undef JCH_KM
        }
        return launch_kpass_tile_lazy(ctx, Xr, n, p, ldr, Yr, qpad, q, d, pend_p, npend, tpend, tstride, cvec, flush, Knext);
    }
#define JCH_KL(KC, R) do { \
        if (q == 1) return launch_kpass_lazy_t<KC, R, 1>(ctx, Xr, n, p, ldr, Yr, qpad, d, pend_p, npend, npend_max, tpend, tstride, cvec, flush, Knext); \
        if (q == 2) return launch_kpass_lazy_t<KC, R, 2>(ctx, Xr, n, p, ldr, Yr, qpad, d, pend_p, npend, npend_max, tpend, tstride, cvec, flush, Knext); \
        return launch_kpass_lazy_t<KC, R, 4>(ctx, Xr, n, p, ldr, Yr, qpad, d, pend_p, npend, npend_max, tpend, tstride, cvec, flush, Knext); } while (0)
    if (ldr <= 128) JCH_KL(1, 4);
    if (ldr <= 256) JCH_KL(2, 4);
    if (ldr <= 512) JCH_KL(4, 2);
    if (ldr <= 1024) JCH_KL(8, 1);
    if (q == 1) return launch_kpass_lazy_t<16, 2, 1>(ctx, Xr, n, p, ldr, Yr, qpad, d, pend_p, npend, npend_max, tpend, tstride, cvec, flush, Knext);
    return launch_kpass_lazy_t<16, 1, 2>(ctx, Xr, n, p, ldr, Yr, qpad, d, pend_p, npend, npend_max, tpend, tstride, cvec, flush, Knext);
#undef JCH_KL
}

int32_t jch_launch_deflate(jch_ctx *ctx, double *Xr, int64_t n, int p, int ldr, double *Yr, int qpad, int q,
                           const double *d, const double *tcol, const double *zpc, double *Knext)
{
    if (q <= 4 && !getenv("JCH_DEFLATE_TILE")) {
#define JCH_DS(KC, R) do { \
        if (q == 1) return launch_deflate_stream<KC, R, 1>(ctx, Xr, n, p, ldr, Yr, qpad, d, tcol, zpc, Knext); \
        if (q == 2) return launch_deflate_stream<KC, R, 2>(ctx, Xr, n, p, ldr, Yr, qpad, d, tcol, zpc, Knext); \
        return launch_deflate_stream<KC, R, 4>(ctx, Xr, n, p, ldr, Yr, qpad, d, tcol, zpc, Knext); } while (0)
        if (ldr <= 128) JCH_DS(1, 4);
        if (ldr <= 256) JCH_DS(2, 4);
        if (ldr <= 512) JCH_DS(4, 2);
        if (ldr <= 1024) JCH_DS(8, 1);
        if (ldr <= 2048) {   // Q x KC accumulators: keep registers in check at the widest rows
            if (q == 1) {
                static int r2 = -1;
                if (r2 < 0) { const char *e = getenv("JCH_DEFLATE_R2"); r2 = e ? atoi(e) : 1; }   // two rows per wave-iteration: 77.85 -> 77.2 ms per 10 LVs at cfg4 shape
                if (r2) return launch_deflate_stream<16, 2, 1>(ctx, Xr, n, p, ldr, Yr, qpad, d, tcol, zpc, Knext);
                return launch_deflate_stream<16, 1, 1>(ctx, Xr, n, p, ldr, Yr, qpad, d, tcol, zpc, Knext);
            }
            if (q == 2) return launch_deflate_stream<16, 1, 2>(ctx, Xr, n, p, ldr, Yr, qpad, d, tcol, zpc, Knext);
        }
#undef JCH_DS
    }
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
