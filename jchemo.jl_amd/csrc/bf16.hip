// bf16 STORAGE mode of plskern (BASELINE.json configs[2]: X/Y stored bf16, rows sharded over GPUs).
//
// The reference is Float64 only; this mode's contract is "the Float64 algorithm applied to the
// bf16-rounded inputs" with fp32 row arithmetic in the sweep and fp64 everywhere else:
//   * X, Y arrive as bf16 column-major (device-resident).  Means / stds / XtY are computed in fp64 from the exact
//     bf16 values (K1/K2 variants below) — identical to the f64 path on the rounded inputs.
//   * the row-major working copy keeps the RAW bf16 values (centring a bf16 value would round it again, 2^-9
//     relative); centring and scaling are applied algebraically in the sweep:
//         t_i  = sum_j x_ij * rt_j - off,        rt_j = r_j / s_j,  off = sum_j m_j rt_j
//         zp_j = (sum_i d_i t_i x_ij - m_j * sum_i d_i t_i) / s_j
//   * the sweep accumulates in fp32 inside a wave (<= a few hundred rows per wave), converts to fp64 for the
//     cross-wave / cross-block / cross-GPU reductions; all p x q state stays fp64 (H1: the small singular-value gaps
//     amplify any perturbation of XtY).
// Bytes per LV: n * ld * 2 (+ 16 n): 4x less than f64.
#include <stdlib.h>

#include <algorithm>

#include "jch_internal.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ float bflo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// ---------------------------------------------------------------- K1 (bf16 input)
template <bool VAR>
__global__ __launch_bounds__(256) void k_moments_bf16(const bf16_t *__restrict__ Xc, int64_t ldx, const bf16_t *__restrict__ Yc,
                                                       int64_t ldy, const double *__restrict__ d, int64_t n, int p, int q,
                                                       int64_t chunk, const double *__restrict__ means,
                                                       double *__restrict__ colpart)
{
    __shared__ double sc[4];
    const int j = blockIdx.x;
    const bf16_t *col = j < p ? Xc + (size_t)j * (size_t)ldx : Yc + (size_t)(j - p) * (size_t)ldy;
    const int64_t i0 = (int64_t)blockIdx.y * chunk;
    const int64_t i1 = i0 + chunk < n ? i0 + chunk : n;
    const double m = VAR ? means[j] : 0.0;
    double s0 = 0.0, s1 = 0.0;
    int64_t i = i0 + threadIdx.x;
    for (; i + 256 < i1; i += 512) {
        double a0 = (double)bf2f(col[i]), a1 = (double)bf2f(col[i + 256]);
        if (VAR) { a0 -= m; a1 -= m; a0 *= a0; a1 *= a1; }
        s0 += d[i] * a0; s1 += d[i + 256] * a1;
    }
    for (; i < i1; i += 256) {
        double a0 = (double)bf2f(col[i]);
        if (VAR) { a0 -= m; a0 *= a0; }
        s0 += d[i] * a0;
    }
    const double s = jch_block_sum<256>(s0 + s1, sc);
    if (threadIdx.x == 0) colpart[(size_t)blockIdx.y * (size_t)(p + q) + j] = s;
}

__global__ __launch_bounds__(256) void k_colreduce_b(const double *__restrict__ colpart, int S, int m, double *__restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    double s = 0.0;
    for (int k = 0; k < S; ++k) s += colpart[(size_t)k * m + j];
    out[j] = s;
}
__global__ __launch_bounds__(256) void k_fill_b(double *__restrict__ v, int m, double c)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < m) v[j] = c;
}
__global__ __launch_bounds__(256) void k_sqrt_b(double *__restrict__ v, int m)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < m) v[j] = sqrt(v[j]);
}

static int32_t launch_moments_bf16(jch_ctx *ctx, const bf16_t *Xc, int64_t ldx, const bf16_t *Yc, int64_t ldy, const double *d,
                                   int64_t n, int p, int q, const double *means, double *out)
{
    const int m = p + q;
    int S = std::min(64, std::max(1, (ctx->cus * 8 + m - 1) / m));
    int64_t chunk = ((n + S - 1) / S + 255) / 256 * 256;
    if (chunk < 256) chunk = 256;
    S = (int)std::max<int64_t>(1, (n + chunk - 1) / chunk);
    JCH_TRY(jch_reserve(ctx, ctx->colpart, sizeof(double) * ((size_t)S * m + 4096)));
    double *colpart = (double *)ctx->colpart.ptr;
    if (means) hipLaunchKernelGGL(k_moments_bf16<true>, dim3(m, S), dim3(256), 0, ctx->stream, Xc, ldx, Yc, ldy, d, n, p, q, chunk, means, colpart);
    else hipLaunchKernelGGL(k_moments_bf16<false>, dim3(m, S), dim3(256), 0, ctx->stream, Xc, ldx, Yc, ldy, d, n, p, q, chunk, means, colpart);
    hipLaunchKernelGGL(k_colreduce_b, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, colpart, S, m, out);
    JCH_TRY(jch_allreduce_f64(ctx, out, (size_t)m));
    if (means) hipLaunchKernelGGL(k_sqrt_b, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, out, m);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

// ---------------------------------------------------------------- K2 (bf16 input): XtY in fp64 on the matrix cores
// from the exact bf16 values (centred/scaled in fp64), raw bf16 row-major copy (ld = ldr, multiple of 8, pad 0).
#define XT_LD 65
#define YT_LD 17
template <bool SCAL>
__global__ __launch_bounds__(256) void k_center_xty_bf16(const bf16_t *__restrict__ Xc, int64_t ldx, const bf16_t *__restrict__ Yc,
                                                          int64_t ldy, const double *__restrict__ d, int64_t n, int p, int q,
                                                          const double *__restrict__ mom, const double *__restrict__ scl,
                                                          bf16_t *__restrict__ Xr, int ldr, double *__restrict__ Yr, int qpad,
                                                          double *__restrict__ Kpart, int kp_rows)
{
    __shared__ double xt[64 * XT_LD];
    __shared__ double yt[64 * YT_LD];
    __shared__ bf16_t xraw[64 * 66];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j0 = blockIdx.y * 64, yg = blockIdx.z;
    const int64_t nchunks = (n + 63) / 64;
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t i0 = c * 64;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = tid + 256 * k, row = e & 63, col = e >> 6;
            const int yc = yg * 16 + col;
            const int64_t i = i0 + row;
            double v = 0.0, dv = 0.0;
            if (i < n && yc < q) {
                v = (double)bf2f(Yc[(size_t)i + (size_t)yc * (size_t)ldy]) - mom[p + yc];
                if (SCAL) v /= scl[p + yc];
                dv = d[i];
            }
            if (blockIdx.y == 0 && i < n) Yr[(size_t)i * qpad + yc] = v;
            yt[row * YT_LD + col] = dv * v;
        }
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int col = wv + 4 * k, j = j0 + col;
            const int64_t i = i0 + lane;
            double v = 0.0;
            bf16_t raw = 0;
            if (i < n && j < p) {
                raw = Xc[(size_t)i + (size_t)j * (size_t)ldx];
                v = (double)bf2f(raw) - mom[j];
                if (SCAL) v /= scl[j];
            }
            xt[lane * XT_LD + col] = v;
            xraw[lane * 66 + col] = raw;
        }
        __syncthreads();
        if (yg == 0) {
#pragma unroll 4
            for (int k = 0; k < 16; ++k) {
                const int row = wv + 4 * k, j = j0 + lane;
                const int64_t i = i0 + row;
                if (i < n && j < ldr) Xr[(size_t)i * ldr + j] = xraw[row * 66 + lane];
            }
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int row = 4 * kk + (lane >> 4);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xt[row * XT_LD + 16 * wv + (lane & 15)], yt[row * YT_LD + (lane & 15)], acc,
                                                       0, 0, 0);
        }
        __syncthreads();
    }
    double *kp = Kpart + ((size_t)blockIdx.x * kp_rows) * qpad;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int j = j0 + 16 * wv + (lane >> 4) + 4 * reg;
        if (j < kp_rows) kp[(size_t)j * qpad + yg * 16 + (lane & 15)] = acc[reg];
    }
}

__global__ __launch_bounds__(256) void k_reduce_kpart_b(const double *__restrict__ Kpart, int nbx, int kp_rows, int p, int qpad,
                                                        double *__restrict__ K)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= p * qpad) return;
    const size_t stride = (size_t)kp_rows * qpad;
    double s = 0.0;
    for (int b = 0; b < nbx; ++b) s += Kpart[(size_t)b * stride + e];
    K[e] = s;
}

// Sum R per-lane partials s[0..R) over the 64 lanes of a wave and return all R totals in every lane.
// Halving butterfly: at each of the first log2(R) steps a lane keeps one half of its values and ships the other half
// to its partner (R/2 + R/4 + ... + 1 shuffles instead of 6 R), three plain butterfly steps finish the remaining
// lane bits, v_readlane broadcasts the totals.  Fixed order -> deterministic.
template <int R>
__device__ __forceinline__ void wave_sum_rows(float (&s)[R], float (&t)[R])
{
    static_assert(R == 2 || R == 4 || R == 8, "R");
    const int lane = threadIdx.x & 63;
    float cur[R];
#pragma unroll
    for (int i = 0; i < R; ++i) cur[i] = s[i];
    int width = R;
    int bit = 32;
#pragma unroll
    for (int st = 0; st < 3; ++st) {
        if (width > 1) {
            const int half = width / 2;
            const bool hi = (lane & bit) != 0;
#pragma unroll
            for (int i = 0; i < R / 2; ++i) {
                if (i < half) {
                    const float mine = hi ? cur[half + i] : cur[i];
                    const float other = hi ? cur[i] : cur[half + i];
                    cur[i] = mine + __shfl_xor(other, bit, 64);
                }
            }
            width = half;
            bit >>= 1;
        }
    }
    float c = cur[0];
    for (int o = bit; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    // row r lives in the lanes whose top log2(R) bits spell r (bit 5 = most significant)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int src = 0;
        if (R == 8) src = ((r >> 2) & 1) * 32 + ((r >> 1) & 1) * 16 + (r & 1) * 8;
        if (R == 4) src = ((r >> 1) & 1) * 32 + (r & 1) * 16;
        if (R == 2) src = (r & 1) * 32;
        t[r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), src));
    }
}

// ---------------------------------------------------------------- K4 (bf16 storage): fused sweep, fp32 row arithmetic
// lane l owns columns 8l..8l+7 (+512k): one 16-B load per row chunk.  rt, off: see the header comment.
template <int KC, int R>
__global__ __launch_bounds__(256) void k_sweep_bf16(const bf16_t *__restrict__ Xr, int64_t n, int ldr, const double *__restrict__ dw,
                                                    const float *__restrict__ rt, const double *__restrict__ offp,
                                                    double *__restrict__ tcol, double *__restrict__ part, int ldpart)
{
    extern __shared__ __attribute__((aligned(16))) double red[];  // [4][KC*512] + [8]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float rf[KC][8], zp[KC][8];
    bool in[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int col = 8 * lane + 512 * k;
        in[k] = col < ldr;
#pragma unroll
        for (int e = 0; e < 8; ++e) { rf[k][e] = in[k] ? rt[col + e] : 0.f; zp[k][e] = 0.f; }
    }
    const float off = (float)offp[0];
    double tt = 0.0, st = 0.0;
    const int64_t ngroups = (n + R - 1) / R;
    const int64_t gstride = (int64_t)gridDim.x * 4;
    for (int64_t g = (int64_t)blockIdx.x * 4 + wv; g < ngroups; g += gstride) {
        const int64_t row0 = g * R;
        v4u32 x[R][KC];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const bool live = row0 + rr < n;
            const v4u32 *rp = reinterpret_cast<const v4u32 *>(Xr + (size_t)(row0 + rr) * (size_t)ldr) + lane;
#pragma unroll
            for (int k = 0; k < KC; ++k) x[rr][k] = (live && in[k]) ? __builtin_nontemporal_load(rp + 64 * k) : v4u32{0u, 0u, 0u, 0u};
        }
        double tsel = 0.0;
        float sp[R], tr[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < KC; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) s += bflo(x[rr][k][e]) * rf[k][2 * e] + bfhi(x[rr][k][e]) * rf[k][2 * e + 1];
            sp[rr] = s;
        }
        wave_sum_rows<R>(sp, tr);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const bool live = row0 + rr < n;
            const float t = tr[rr] - off;
            const float dtf = live ? (float)dw[row0 + rr] * t : 0.f;
            tt += (double)dtf * (double)t;
            st += (double)dtf;
#pragma unroll
            for (int k = 0; k < KC; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    zp[k][2 * e] += dtf * bflo(x[rr][k][e]);
                    zp[k][2 * e + 1] += dtf * bfhi(x[rr][k][e]);
                }
            if (lane == rr) tsel = (double)t;
        }
        if (lane < R && row0 + lane < n) tcol[row0 + lane] = tsel;
    }
    double *zred = red;                 // [4][KC*512]
    double *tred = red + 4 * KC * 512;  // [8]
#pragma unroll
    for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) zred[wv * (KC * 512) + 8 * lane + 512 * k + e] = (double)zp[k][e];
    if (lane == 0) { tred[wv] = tt; tred[4 + wv] = st; }
    __syncthreads();
    double *prow = part + (size_t)blockIdx.x * ldpart;
    for (int c = threadIdx.x; c < ldr; c += 256)
        prow[c] = ((zred[c] + zred[KC * 512 + c]) + zred[2 * KC * 512 + c]) + zred[3 * KC * 512 + c];
    if (threadIdx.x == 0) {
        prow[ldr] = ((tred[0] + tred[1]) + tred[2]) + tred[3];
        prow[ldr + 1] = ((tred[4] + tred[5]) + tred[6]) + tred[7];
    }
}

__global__ __launch_bounds__(1024) void k_reduce_part_b(const double *__restrict__ part, int nb, int ldpart, int m,
                                                        double *__restrict__ zt)
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
        zt[c] = t;
    }
}

// zp_j <- (zp_raw_j - m_j * st) / s_j ;  slot [ldz_tt] <- tt      (after the cross-GPU all-reduce)
__global__ __launch_bounds__(256) void k_bf16_fix_zt(double *__restrict__ zt, int ldr_b, int p, int ldr_small,
                                                     const double *__restrict__ mom, const double *__restrict__ scl,
                                                     double *__restrict__ zt_small)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const double tt = zt[ldr_b], st = zt[ldr_b + 1];
    if (j < ldr_small) zt_small[j] = j < p ? (zt[j] - mom[j] * st) / scl[j] : 0.0;
    if (j == 0) zt_small[ldr_small] = tt;
}

// rt_j = r_j / s_j (fp32), off = sum_j m_j * rt_j (fp64, from the fp32-rounded rt so that the sweep's row sums and the
// offset use the same coefficients)
__global__ __launch_bounds__(256) void k_bf16_make_rt(const double *__restrict__ r, int p, int ldr_b, const double *__restrict__ mom,
                                                      const double *__restrict__ scl, float *__restrict__ rt, double *__restrict__ off)
{
    __shared__ double sc[4];
    double s = 0.0;
    for (int j = threadIdx.x; j < ldr_b; j += 256) {
        float v = 0.f;
        if (j < p) { v = (float)(r[j] / scl[j]); s += mom[j] * (double)v; }
        rt[j] = v;
    }
    s = jch_block_sum<256>(s, sc);
    if (threadIdx.x == 0) off[0] = s;
}

template <int KC, int R>
static int32_t launch_sweep_bf16_t(jch_ctx *ctx, const bf16_t *Xr, int64_t n, int ldr_b, const double *d, const float *rt,
                                   const double *off, double *tcol, double *zt_raw)
{
    const size_t lds = sizeof(double) * (4 * KC * 512 + 8);
    static int bpc = 0;
    if (bpc == 0) {
        int nblk = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_sweep_bf16<KC, R>, 256, lds);
        bpc = (e == hipSuccess && nblk > 0) ? nblk : 2;
        if (lds > 64 * 1024)
            JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_sweep_bf16<KC, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const int64_t ngroups = (n + R - 1) / R;
    static int bpc_env = -1;
    if (bpc_env < 0) { const char *e = getenv("JCH_BF16_BPC"); bpc_env = e ? atoi(e) : 0; }
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((ngroups + 3) / 4, (int64_t)ctx->cus * (bpc_env > 0 ? bpc_env : bpc)));
    const int m = ldr_b + 2, ldpart = (m + 7) & ~7;
    JCH_TRY(jch_reserve(ctx, ctx->part, sizeof(double) * (size_t)nb * ldpart));
    double *part = (double *)ctx->part.ptr;
    (void)jch_ev(ctx);
    hipLaunchKernelGGL((k_sweep_bf16<KC, R>), dim3(nb), dim3(256), lds, ctx->stream, Xr, n, ldr_b, d, rt, off, tcol, part, ldpart);
    (void)jch_ev(ctx);
    JCH_TRY(jch_launch_reduce_rows(ctx, part, nb, ldpart, m, zt_raw));
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

// ---------------------------------------------------------------- orchestration (called from fit.hip)
int32_t jch_fit_plskern_bf16(jch_ctx *ctx, const jch_pls_desc &d, const void *Xv, int64_t ldx, const void *Yv, int64_t ldy,
                             const double *wdev, double *dn, double *Tdev, jch_small &s, int ldr_small, int qpad, int ldz,
                             bool fast, int *nlv_out)
{
    const int64_t n = d.n;
    const int p = (int)d.p, q = (int)d.q;
    const bf16_t *Xc = (const bf16_t *)Xv, *Yc = (const bf16_t *)Yv;
    const int ldr_b = (p + 7) & ~7;                    // bf16 row stride: 16-B aligned rows
    JCH_TRY(jch_reserve(ctx, ctx->xr, sizeof(bf16_t) * (size_t)n * ldr_b));
    JCH_TRY(jch_reserve(ctx, ctx->yr, sizeof(double) * (size_t)n * qpad));
    bf16_t *Xr = (bf16_t *)ctx->xr.ptr;
    double *Yr = (double *)ctx->yr.ptr;
    JCH_TRY(jch_reserve(ctx, ctx->gemm_b, sizeof(double) * ((size_t)ldr_b + 16) + sizeof(float) * (size_t)ldr_b + 64));
    double *zt_raw = (double *)ctx->gemm_b.ptr;        // [ldr_b + 2] (+pad)
    float *rt = (float *)(zt_raw + ((ldr_b + 2 + 7) & ~7));
    double *off = zt_raw + ldr_b + 4;
    // ---- prologue
    JCH_TRY(jch_launch_weights(ctx, wdev, n, dn, s.hdr));
    int64_t n_total = n;
    if (ctx->nranks > 1 && n < std::min<int64_t>(p, d.nlv)) {   // see fit.hip: larger shards never need the global count
        double hdr_h[2];
        JCH_HIP(ctx, hipMemcpyAsync(hdr_h, s.hdr, sizeof hdr_h, hipMemcpyDeviceToHost, ctx->stream));
        JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        n_total = (int64_t)(hdr_h[1] + 0.5);
    }
    const int nlv = (int)std::min<int64_t>(std::min<int64_t>(n_total, p), d.nlv);
    JCH_TRY(launch_moments_bf16(ctx, Xc, ldx, Yc, ldy, dn, n, p, q, nullptr, s.mom));
    if (d.scal) JCH_TRY(launch_moments_bf16(ctx, Xc, ldx, Yc, ldy, dn, n, p, q, s.mom, s.scl));
    else hipLaunchKernelGGL(k_fill_b, dim3((p + q + 255) / 256), dim3(256), 0, ctx->stream, s.scl, p + q, 1.0);
    {
        const int ptiles = (ldr_b + 63) / 64, kp_rows = ptiles * 64, ygroups = qpad / 16;
        const int64_t nchunks = (n + 63) / 64;
        int nbx = std::max(1, (ctx->cus * 3 + ptiles * ygroups - 1) / (ptiles * ygroups));
        if (nbx > nchunks) nbx = (int)std::max<int64_t>(nchunks, 1);
        JCH_TRY(jch_reserve(ctx, ctx->kpart, sizeof(double) * (size_t)nbx * kp_rows * qpad));
        double *Kpart = (double *)ctx->kpart.ptr;
        dim3 grid(nbx, ptiles, ygroups);
        if (d.scal) hipLaunchKernelGGL(k_center_xty_bf16<true>, grid, dim3(256), 0, ctx->stream, Xc, ldx, Yc, ldy, dn, n, p, q, s.mom,
                                       s.scl, Xr, ldr_b, Yr, qpad, Kpart, kp_rows);
        else hipLaunchKernelGGL(k_center_xty_bf16<false>, grid, dim3(256), 0, ctx->stream, Xc, ldx, Yc, ldy, dn, n, p, q, s.mom,
                                s.scl, Xr, ldr_b, Yr, qpad, Kpart, kp_rows);
        hipLaunchKernelGGL(k_reduce_kpart_b, dim3((p * qpad + 255) / 256), dim3(256), 0, ctx->stream, Kpart, nbx, kp_rows, p, qpad, s.K);
        JCH_TRY(jch_allreduce_f64(ctx, s.K, (size_t)p * qpad));
    }
    (void)jch_ev(ctx);  // end of prologue
    ctx->ev_mark = ctx->ev_used;  // (begin, end) event pairs of the sweeps start here
    // ---- LV loop
    JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr_small, -1, nlv, 0, 1, ldz, fast));
    for (int a = 0; a < nlv; ++a) {
        hipLaunchKernelGGL(k_bf16_make_rt, dim3(1), dim3(256), 0, ctx->stream, s.r, p, ldr_b, s.mom, s.scl, rt, off);
        double *tcol = Tdev + (size_t)a * (size_t)n;
        static int rsel = -1;
        if (rsel < 0) { const char *e = getenv("JCH_BF16_R"); rsel = e ? atoi(e) : 2; }   // measured at n = 1e6, p = 500: R = 8 / 4 / 2 -> 4.0 / 4.85 / 5.2 TB/s
        if (ldr_b <= 512 && rsel == 4) JCH_TRY((launch_sweep_bf16_t<1, 4>(ctx, Xr, n, ldr_b, dn, rt, off, tcol, zt_raw)));
        else if (ldr_b <= 512 && rsel == 2) JCH_TRY((launch_sweep_bf16_t<1, 2>(ctx, Xr, n, ldr_b, dn, rt, off, tcol, zt_raw)));
        else if (ldr_b <= 512) JCH_TRY((launch_sweep_bf16_t<1, 8>(ctx, Xr, n, ldr_b, dn, rt, off, tcol, zt_raw)));
        else if (ldr_b <= 1024) JCH_TRY((launch_sweep_bf16_t<2, 4>(ctx, Xr, n, ldr_b, dn, rt, off, tcol, zt_raw)));
        else JCH_TRY((launch_sweep_bf16_t<4, 2>(ctx, Xr, n, ldr_b, dn, rt, off, tcol, zt_raw)));
        JCH_TRY(jch_allreduce_f64(ctx, zt_raw, (size_t)ldr_b + 2));   // ONE collective per LV: [zp_raw, tt, st]
        hipLaunchKernelGGL(k_bf16_fix_zt, dim3((ldr_small + 255) / 256), dim3(256), 0, ctx->stream, zt_raw, ldr_b, p, ldr_small, s.mom,
                           s.scl, s.zt);
        JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr_small, a, nlv, 0, 1, ldz, fast));
    }
    JCH_HIP(ctx, hipGetLastError());
    *nlv_out = nlv;
    return JCH_OK;
}
