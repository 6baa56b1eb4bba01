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
#include <type_traits>

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

// 16-byte variant of K1: a thread owns 8 consecutive rows (one 16-B load per column, the 8 weights stay in registers)
// and walks over MCG columns, so the weight vector is read once per MCG columns instead of once per column (it is 4x
// the bytes of a bf16 column) and every global access is 16 B.  Needs 16-B aligned bases and leading dimensions % 8.
template <bool VAR, int MCG>
__global__ __launch_bounds__(256) void k_moments_bf16_v8(const bf16_t *__restrict__ Xc, int64_t ldx, const bf16_t *__restrict__ Yc,
                                                          int64_t ldy, const double *__restrict__ d, int64_t n, int p, int q,
                                                          int64_t chunk, const double *__restrict__ means,
                                                          double *__restrict__ colpart)
{
    __shared__ double sc[4];
    typedef double v2f64_ __attribute__((ext_vector_type(2)));
    const int j0 = blockIdx.x * MCG;
    const int m = p + q;
    const int64_t i0 = (int64_t)blockIdx.y * chunk;                 // chunk is a multiple of 8 * 256
    const int64_t i1 = i0 + chunk < n ? i0 + chunk : n;
    double mj[MCG], acc[MCG];
#pragma unroll
    for (int c = 0; c < MCG; ++c) { mj[c] = (VAR && j0 + c < m) ? means[j0 + c] : 0.0; acc[c] = 0.0; }
    for (int64_t i = i0 + 8 * (int64_t)threadIdx.x; i < i1; i += 8 * 256) {
        double dv[8];
        if (i + 8 <= n) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const v2f64_ t = *reinterpret_cast<const v2f64_ *>(d + i + 2 * k);
                dv[2 * k] = t.x; dv[2 * k + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) dv[k] = i + k < n ? d[i + k] : 0.0;
        }
#pragma unroll
        for (int c = 0; c < MCG; ++c) {
            const int j = j0 + c;
            if (j >= m) break;
            const bf16_t *col = j < p ? Xc + (size_t)j * (size_t)ldx : Yc + (size_t)(j - p) * (size_t)ldy;
            float x[8];
            if (i + 8 <= n) {
                const v4u32 w = __builtin_nontemporal_load(reinterpret_cast<const v4u32 *>(col + i));
                x[0] = bflo(w.x); x[1] = bfhi(w.x); x[2] = bflo(w.y); x[3] = bfhi(w.y);
                x[4] = bflo(w.z); x[5] = bfhi(w.z); x[6] = bflo(w.w); x[7] = bfhi(w.w);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = i + k < n ? bf2f(col[i + k]) : 0.0f;
            }
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                double v = (double)x[k];
                if (VAR) { v -= mj[c]; v *= v; }
                a += dv[k] * v;
            }
            acc[c] += a;
        }
    }
#pragma unroll
    for (int c = 0; c < MCG; ++c) {
        const double t = jch_block_sum<256>(acc[c], sc);
        if (threadIdx.x == 0 && j0 + c < m) colpart[(size_t)blockIdx.y * (size_t)m + j0 + c] = t;
    }
}

// out[j] = sum_k colpart[k][j] in a fixed order: 16 interleaved streams of partial rows per column, combined in stream order
// (one thread walking all S rows of a column was latency-bound: 81 us at cfg3 for 2 x 256 threads)
__global__ __launch_bounds__(1024) void k_colreduce_b(const double *__restrict__ colpart, int S, int m, double *__restrict__ out)
{
    __shared__ double sc[16][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + cl;
    double s = 0.0;
    if (j < m)
        for (int k = g; k < S; k += 16) s += colpart[(size_t)k * m + j];
    sc[g][cl] = s;
    __syncthreads();
    if (g == 0 && j < m) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sc[k][cl];
        out[j] = t;
    }
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
    const bool v8 = ldx % 8 == 0 && ldy % 8 == 0 && ((uintptr_t)Xc) % 16 == 0 && ((uintptr_t)Yc) % 16 == 0 && ((uintptr_t)d) % 16 == 0 &&
                    !getenv("JCH_BF16_SCALAR_PROLOGUE");
    if (v8) {
        static int mcg = -1;
        if (mcg < 0) { const char *e = getenv("JCH_BF16_MCG"); mcg = e ? atoi(e) : 8; }   // measured: 16 / 8 / 4 columns per thread -> prologue 1.40 / 1.27 / 1.29 ms at n = 1e6, p = 500
        const int MCGv = mcg == 4 ? 4 : (mcg == 16 ? 16 : 8);
        const int cg = (m + MCGv - 1) / MCGv;
        int S = std::max(1, (ctx->cus * 8 + cg - 1) / cg);
        int64_t chunk = ((n + S - 1) / S + 2047) / 2048 * 2048;      // multiple of 8 rows x 256 threads
        S = (int)std::max<int64_t>(1, (n + chunk - 1) / chunk);
        JCH_TRY(jch_reserve(ctx, ctx->colpart, sizeof(double) * ((size_t)S * m + 4096)));
        double *colpart = (double *)ctx->colpart.ptr;
#define JCH_K1B(V, M) hipLaunchKernelGGL((k_moments_bf16_v8<V, M>), dim3(cg, S), dim3(256), 0, ctx->stream, Xc, ldx, Yc, ldy, d, n, p, q, chunk, means, colpart)
        if (means) { if (MCGv == 4) JCH_K1B(true, 4); else if (MCGv == 8) JCH_K1B(true, 8); else JCH_K1B(true, 16); }
        else { if (MCGv == 4) JCH_K1B(false, 4); else if (MCGv == 8) JCH_K1B(false, 8); else JCH_K1B(false, 16); }
#undef JCH_K1B
        hipLaunchKernelGGL(k_colreduce_b, dim3((m + 63) / 64), dim3(1024), 0, ctx->stream, colpart, S, m, out);
        JCH_TRY(jch_allreduce_f64(ctx, out, (size_t)m));
        if (means) hipLaunchKernelGGL(k_sqrt_b, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, out, m);
        JCH_HIP(ctx, hipGetLastError());
        return JCH_OK;
    }
    int S = std::min(64, std::max(1, (ctx->cus * 8 + m - 1) / m));
    int64_t chunk = ((n + S - 1) / S + 255) / 256 * 256;
    if (chunk < 256) chunk = 256;
    S = (int)std::max<int64_t>(1, (n + chunk - 1) / chunk);
    JCH_TRY(jch_reserve(ctx, ctx->colpart, sizeof(double) * ((size_t)S * m + 4096)));
    double *colpart = (double *)ctx->colpart.ptr;
    if (means) hipLaunchKernelGGL(k_moments_bf16<true>, dim3(m, S), dim3(256), 0, ctx->stream, Xc, ldx, Yc, ldy, d, n, p, q, chunk, means, colpart);
    else hipLaunchKernelGGL(k_moments_bf16<false>, dim3(m, S), dim3(256), 0, ctx->stream, Xc, ldx, Yc, ldy, d, n, p, q, chunk, means, colpart);
    hipLaunchKernelGGL(k_colreduce_b, dim3((m + 63) / 64), dim3(1024), 0, ctx->stream, colpart, S, m, out);
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

// 16-byte variant of K2: a lane loads 8 consecutive rows of one column (16 B), and stores 8 consecutive columns of
// one row of the row-major copy (16 B); same tile, same LDS transposition, same MFMA XtY as above.
#define XR_LD 72
template <bool SCAL>
__global__ __launch_bounds__(256) void k_center_xty_bf16_v8(const bf16_t *__restrict__ Xc, int64_t ldx, const bf16_t *__restrict__ Yc,
                                                             int64_t ldy, const double *__restrict__ d, int64_t n, int p, int q,
                                                             const double *__restrict__ mom, const double *__restrict__ scl,
                                                             bf16_t *__restrict__ Xr, int ldr, double *__restrict__ Yr, int qpad,
                                                             double *__restrict__ Kpart, int kp_rows, int ones_col, int dbg_skip)
{
    __shared__ double yt[64 * YT_LD];
    // The X tile lives in LDS as RAW bf16 only, COLUMN-major [col][XR_LD rows] (9 KB instead of a 33 KB fp64 tile: 8 blocks per
    // CU instead of 3; a 256-row tile — 512-B column runs, 2 blocks per CU — was measured 0.11 ms SLOWER): a lane's 8 consecutive rows of one column go in with one 16-B store, the row-major output gathers 8
    // columns of one row, and the MFMA A-operand is converted to fp64 and centred when it is read (all conflict-free).
    __shared__ __attribute__((aligned(16))) bf16_t xraw[64 * XR_LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j0 = blockIdx.y * 64, yg = blockIdx.z;
    const int64_t nchunks = (n + 63) / 64;
    const int rg = lane & 7, cl = lane >> 3;          // load role: rows 8 rg .. 8 rg + 7 of column (2 wv + k) * 8 + cl
    // MFMA role of this lane: X column j0 + 16 wv + (lane & 15), rows 4 kk + (lane >> 4)
    const int ja = j0 + 16 * wv + (lane & 15);
    const double cma = ja < p ? mom[ja] : 0.0;
    const double csa = (SCAL && ja < p) ? scl[ja] : 1.0;
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    // software pipeline: the global loads of chunk c + gridDim.x are in flight while chunk c is transposed / multiplied
    v4u32 xw[2];
    bf16_t yraw[4];
    double dreg[4];
    auto prefetch = [&](int64_t cc) {
        const int64_t i0 = cc * 64;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = tid + 256 * k, row = e & 63, col = e >> 6;
            const int yc = yg * 16 + col;
            const int64_t i = i0 + row;
            const bool ok = i < n && yc < q;
            yraw[k] = ok ? Yc[(size_t)i + (size_t)yc * (size_t)ldy] : (bf16_t)0;
            dreg[k] = ok ? d[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = j0 + (2 * wv + k) * 8 + cl;
            const int64_t i = i0 + 8 * rg;
            v4u32 w = {0u, 0u, 0u, 0u};
            if (j < p && i + 8 <= n) {
                w = __builtin_nontemporal_load(reinterpret_cast<const v4u32 *>(Xc + (size_t)i + (size_t)j * (size_t)ldx));
            } else if (j < p) {   // row tail: element loads, packed like the vector
                unsigned h[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) h[r] = i + r < n ? (unsigned)Xc[(size_t)(i + r) + (size_t)j * (size_t)ldx] : 0u;
                w.x = h[0] | (h[1] << 16); w.y = h[2] | (h[3] << 16); w.z = h[4] | (h[5] << 16); w.w = h[6] | (h[7] << 16);
            }
            xw[k] = w;
        }
    };
    int64_t c = blockIdx.x;
    if (c < nchunks) prefetch(c);
    for (; c < nchunks; c += gridDim.x) {
        const int64_t i0 = c * 64;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = tid + 256 * k, row = e & 63, col = e >> 6;
            const int yc = yg * 16 + col;
            const int64_t i = i0 + row;
            double v = 0.0, dv = 0.0;
            if (i < n && yc < q) {
                v = (double)bf2f(yraw[k]) - mom[p + yc];
                if (SCAL) v /= scl[p + yc];
                dv = dreg[k];
            }
            if (blockIdx.y == 0 && i < n) Yr[(size_t)i * qpad + yc] = v;
            // raw mode: pad column `ones_col` carries the weights -> column ones_col of X'D[Yc | 1] = sum_i d_i (x_i - pivot)
            yt[row * YT_LD + col] = (yc == ones_col && i < n) ? d[i] : dv * v;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int col = (2 * wv + k) * 8 + cl;
            *reinterpret_cast<v4u32 *>(xraw + col * XR_LD + 8 * rg) = xw[k];
        }
        __syncthreads();
        if (c + gridDim.x < nchunks) prefetch(c + gridDim.x);
        if (yg == 0 && !(dbg_skip & 2)) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int row = (2 * wv + k) * 8 + (lane >> 3), cg = lane & 7;
                const int64_t i = i0 + row;
                const int j = j0 + 8 * cg;
                if (i < n && j < ldr) {
                    unsigned h[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = xraw[(8 * cg + e) * XR_LD + row];
                    const v4u32 w = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
                    __builtin_nontemporal_store(w, reinterpret_cast<v4u32 *>(Xr + (size_t)i * ldr + j));
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int row = 4 * kk + (lane >> 4);
            double a = 0.0;
            if (ja < p && i0 + row < n) {
                a = (double)bf2f(xraw[(16 * wv + (lane & 15)) * XR_LD + row]) - cma;
                if (SCAL) a /= csa;
            }
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, yt[row * YT_LD + (lane & 15)], acc, 0, 0, 0);
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

// ---------------------------------------------------------------- K2pb (round 3): ROW-PANEL prologue for bf16 storage
// The recipe of the f64 k_center_xty_panel (prologue.hip) re-derived for 2-byte elements, where two things change:
//   * a whole TH-row x 512-column tile of raw bf16 fits in LDS (64 x 508 x 2 B = 65 KB), so the row-major copy leaves as
//     COMPLETE rows: the rows of a tile are consecutive in the copy, i.e. one tile = one contiguous 63 KB region written
//     with 16-B stores — no 128-B row pieces straddling two lines any more (the tile kernel's stores were 1.16x the
//     algorithmic bytes: profiles/r02_pmc_bf16_traffic.txt);
//   * the MFMA A-operand needs no LDS at all: lane l loads 8 consecutive rows (16 B) of column 16 wv + (l & 15), rows
//     32 h + 8 (l >> 4) + e, and the product step (h, e) takes element e of load h as A[m = l & 15][k = l >> 4]; the B operand
//     d .* yc is built ONCE per row tile in registers with the same row mapping (k = l >> 4 -> row 32 h + 8 (l >> 4) + e).
//     A wave multiplies exactly the 16 columns it loaded, so nothing but the final transposition goes through LDS and
//     the only block barriers are the two around the row-major store of a tile.
// One block walks the row tiles b, b + G, b + 2G, ... (interleaved, as in the f64 panel); all loads of the NEXT tile (the
// X pieces one by one as their registers are consumed, the Y / weight vectors first) are in flight while the current one
// is converted, multiplied and stored.  Handles FULL tiles only (rows [0, nfull), nfull % TH == 0); the ragged tail
// goes to k_center_xty_bf16_v8.  q <= 16, 16-B aligned columns / weights.
typedef unsigned long long v2u64b __attribute__((ext_vector_type(2)));
// SKIP (measurement only, results then wrong by design): 1 = no products, 2 = no LDS tile / row-major stores
template <int NT, int NH, bool SCAL, int SKIP>
__global__ __launch_bounds__(256, NH <= 2 ? 2 : 1) void k_center_xty_bf16_panel(
    const bf16_t *__restrict__ Xc, int64_t ldx, const bf16_t *__restrict__ Yc, int64_t ldy, const double *__restrict__ d, int64_t nfull,
    int p, int q, const double *__restrict__ mom, const double *__restrict__ scl, bf16_t *__restrict__ Xr, int ldr,
    double *__restrict__ Yr, double *__restrict__ Kpart, int kp_rows, int ones_col)
{
    constexpr int TH = 32 * NH;
    constexpr int NB = TH / 16;                                 // B entries built per thread and row tile
    // LDS row pitch in elements, a compile-time constant (row offsets become instruction immediates): 512 + 4 -> pitch / 2 == 2
    // (mod 4) words: the four row groups of a wave's 2-byte column stores fall into four different 16-word bank windows, and
    // rows stay 8-B aligned for the b64 reads of the store phase.  Columns [wcols, wcols + 4) of a row are a dump area.
    constexpr int pitch = 516;
    extern __shared__ __attribute__((aligned(16))) double bp_lds[];
    double *Bs = bp_lds;                                        // [8 NH][64]: B operand of the tile, one 64-lane row per product step
    double *cm_s = Bs + 8 * NH * 64;                            // [512] column shifts of this column group
    double *cs_s = cm_s + 512;                                  // [512] column divisors (SCAL only)
    bf16_t *xt = reinterpret_cast<bf16_t *>(SCAL ? cs_s + 512 : cs_s);   // [TH][pitch] raw tile, row-major
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = lane >> 4, cidx = lane & 15;
    const int cg0 = blockIdx.y * 512;
    const int wcols = min(ldr - cg0, 512);                      // columns of this group in the copy (multiple of 8)
    const int64_t istep = (int64_t)gridDim.x * TH;
    for (int c = tid; c < 512; c += 256) {
        const int j = cg0 + c;
        cm_s[c] = j < p ? mom[j] : 0.0;
        if (SCAL) cs_s[c] = j < p ? scl[j] : 1.0;
    }
    // ---- B operand d .* yc (and the weights in the ones column), built cooperatively once per row tile: thread -> y column
    //      tid & 15, rows (tid >> 4) + 16 k; lands in Bs in the order the product steps read it (step (h, e), lane 16 g + n
    //      <-> row 32 h + 8 g + e, y column n); the same threads store the centred Y rows (Yr, 512 contiguous bytes a wave)
    const int ycol = tid & 15, yc_ = min(ycol, q - 1);
    const double ym = ycol < q ? mom[p + ycol] : 0.0;
    const double ysd = (SCAL && ycol < q) ? scl[p + ycol] : 1.0;
    bf16_t yraw[NB];
    double draw[NB];
    auto issue_b = [&](int64_t i0) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const size_t row = (size_t)(i0 + (tid >> 4) + 16 * k);
            yraw[k] = Yc[row + (size_t)yc_ * (size_t)ldy];
            draw[k] = d[row];
        }
    };
    auto build_b = [&](int64_t i0) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int row = (tid >> 4) + 16 * k;
            double yv = 0.0;
            if (ycol < q) {
                yv = (double)bf2f(yraw[k]) - ym;
                if (SCAL) yv /= ysd;
            }
            if (blockIdx.y == 0) Yr[(size_t)(i0 + row) * 16 + ycol] = yv;
            // raw mode: the pad column `ones_col` carries the weights -> that column of X'D[Yc | 1] = sum_i d_i (x_i - pivot)
            Bs[(8 * (row >> 5) + (row & 7)) * 64 + 16 * ((row >> 3) & 3) + ycol] = ycol == ones_col ? draw[k] : draw[k] * yv;
        }
    };
    v4f64 acc[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) acc[ct] = v4f64{0.0, 0.0, 0.0, 0.0};
    // X loads in flight: a ring of RING 64-column pieces (piece ct lives in slot ct % RING; when it has been consumed the slot
    // receives piece ct + RING — of the same tile, or of the block's next tile): 4 x 8 KB per block ahead of the arithmetic
    constexpr int RING = NT < 4 ? NT : 4;
    v4u32 R[RING][NH];
    const int jlane = cg0 + 16 * wv + cidx;
    auto issue_piece = [&](int64_t i0, int ct) {                // (columns past p: clamped into the matrix, masked when consumed)
        const size_t coff = (size_t)min(jlane + 64 * ct, p - 1) * (size_t)ldx;
#pragma unroll
        for (int h = 0; h < NH; ++h)
            R[ct % RING][h] = __builtin_nontemporal_load(reinterpret_cast<const v4u32 *>(Xc + (size_t)(i0 + 32 * h + 8 * g) + coff));
    };
    int64_t i0 = (int64_t)blockIdx.x * TH;
    if (i0 < nfull) {
        issue_b(i0);
#pragma unroll
        for (int ct = 0; ct < RING; ++ct) issue_piece(i0, ct);
        build_b(i0);
    }
    __syncthreads();   // cm_s / cs_s / Bs
    auto tile = [&](auto has_next, int64_t i0) {
        if (decltype(has_next)::value) issue_b(i0 + istep);     // (consumed after the products of this tile)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            // straight-line code, no branch per element (a basic block per product kept the scheduler from overlapping the
            // conversions with the matrix pipe): dead columns (>= p: pad columns of the copy, clamped loads of a short last
            // piece) are masked to +0 — their shift is 0, their divisor 1 —, lanes past the group's width store into the
            // dump columns of the LDS row
            const int cc = 64 * ct + 16 * wv + cidx;             // column within the group
            const unsigned lmask = cg0 + cc < p ? 0xffffffffu : 0u;
            const double cm = cm_s[cc & 511];
            const double cs = SCAL ? cs_s[cc & 511] : 1.0;
            bf16_t *xw = xt + 8 * g * pitch + (cc < wcols ? cc : wcols + (cc & 3));
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const unsigned w4[4] = {R[ct % RING][h].x & lmask, R[ct % RING][h].y & lmask, R[ct % RING][h].z & lmask, R[ct % RING][h].w & lmask};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned w = w4[e >> 1];
                    if (!(SKIP & 2)) xw[(32 * h + e) * pitch] = (bf16_t)((e & 1) ? (w >> 16) : w);
                    double a = (double)__uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16)) - cm;
                    if (SCAL) a /= cs;
                    if (!(SKIP & 1)) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs[(8 * h + e) * 64 + lane], acc[ct], 0, 0, 0);
                }
            }
            if (ct + RING < NT) issue_piece(i0, ct + RING);                                      // (folded: the loop is unrolled)
            else if (decltype(has_next)::value) issue_piece(i0 + istep, ct + RING - NT);
        }
        __syncthreads();   // tile complete in LDS; every wave is done reading Bs
        if (decltype(has_next)::value) build_b(i0 + istep);
        // ---- the tile leaves as complete rows: wave wv stores rows wv, wv + 4, ...; 16 B per lane
        if (!(SKIP & 2)) {
            const int c16n = wcols >> 3;
#pragma unroll 4
            for (int row = wv; row < TH; row += 4) {
                for (int c16 = lane; c16 < c16n; c16 += 64) {
                    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(xt + row * pitch + 8 * c16);
                    const v2u64b v = {src[0], src[1]};
                    __builtin_nontemporal_store(v, reinterpret_cast<v2u64b *>(Xr + (size_t)(i0 + row) * (size_t)ldr + cg0 + 8 * c16));
                }
            }
        }
        __syncthreads();   // the tile may be overwritten; Bs of the next tile is published
    };
    for (; i0 + istep < nfull; i0 += istep) tile(std::true_type{}, i0);
    if (i0 < nfull) tile(std::false_type{}, i0);
    // D[m][n]: n = lane & 15 (y column), m = (lane >> 4) + 4 reg (x column within the wave's 16)
    double *kp = Kpart + ((size_t)blockIdx.x * kp_rows) * 16;
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int jj = 64 * ct + 16 * wv + g + 4 * reg;
            if (jj < 512 && cg0 + jj < kp_rows) kp[(size_t)(cg0 + jj) * 16 + cidx] = acc[ct][reg];
        }
    }
}

// ---------------------------------------------------------------- K2pb on the bf16 matrix pipe (round 4)
// UNIT weights, no scaling, q <= 15: the operands of X'[Y | 1] are then the RAW bf16 values themselves — exact bf16 — and the
// product runs on v_mfma_f32_16x16x32_bf16 (16 cycles per 16 x 16 x 32 against 8 x 64 cycles of v_mfma_f64_16x16x4 for the same
// 32 rows: the f64 pipe time of K2pb, 0.21 of its 0.53 ms, disappears).  The 16-B load of 8 consecutive rows of one column IS the
// A operand of a 32-row block (lane l: m = column l & 15, k = rows 8 (l >> 4) ..+7), the same load of a Y column the B operand;
// every bf16 x bf16 product is exact in f32, a block's 32-term sum is taken in f32 inside the instruction and added to an f64
// accumulator right after it, so the only rounding beyond the f64 path's is the f32 sum of 32 exact products per block
// (measured on cfg3-like data at n = 1e6: K within 6e-7, on data with a signal 2e-9 — the mode's budget is 1e-3 / 1e-4).
// Centring happens afterwards in f64 on the p x q sums (k_bf16_m32_fix): K = (X'Y - (X'1) ybar') / n, means = X'1 / n.
// Same tile walk, LDS transposition and row-major stores as k_center_xty_bf16_panel; Yr is not written (plskern never reads it).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float v4f32b __attribute__((ext_vector_type(4)));
template <int NT, int NH>
__global__ __launch_bounds__(256, NH <= 2 ? 2 : 1) void k_xty_bf16_panel_m32(
    const bf16_t *__restrict__ Xc, int64_t ldx, const bf16_t *__restrict__ Yc, int64_t ldy, int64_t nfull, int p, int q,
    bf16_t *__restrict__ Xr, int ldr, double *__restrict__ Kpart, int kp_rows)
{
    constexpr int TH = 32 * NH;
    constexpr int pitch = 516;
    extern __shared__ __attribute__((aligned(16))) double bp_lds[];
    bf16_t *xt = reinterpret_cast<bf16_t *>(bp_lds);           // [TH][pitch] raw tile, row-major
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = lane >> 4, cidx = lane & 15;
    const int cg0 = blockIdx.y * 512;
    const int wcols = min(ldr - cg0, 512);
    const int64_t istep = (int64_t)gridDim.x * TH;
    // B operand: lane (g, n) holds rows 32 h + 8 g ..+7 of y column n (n < q), ones for n == q (column sums of X), zeros beyond
    const size_t yoff = (size_t)min(cidx, q - 1) * (size_t)ldy;
    const unsigned ymask = cidx < q ? 0xffffffffu : 0u, yones = cidx == q ? 0x3f803f80u : 0u;
    v4u32 Bf[NH], Bn[NH];
    auto issue_b = [&](int64_t i0) {
#pragma unroll
        for (int h = 0; h < NH; ++h) Bn[h] = *reinterpret_cast<const v4u32 *>(Yc + (size_t)(i0 + 32 * h + 8 * g) + yoff);
    };
    auto take_b = [&]() {
#pragma unroll
        for (int h = 0; h < NH; ++h)
            Bf[h] = v4u32{(Bn[h].x & ymask) | yones, (Bn[h].y & ymask) | yones, (Bn[h].z & ymask) | yones, (Bn[h].w & ymask) | yones};
    };
    v4f64 acc[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) acc[ct] = v4f64{0.0, 0.0, 0.0, 0.0};
    constexpr int RING = NT < 4 ? NT : 4;
    v4u32 R[RING][NH];
    const int jlane = cg0 + 16 * wv + cidx;
    auto issue_piece = [&](int64_t i0, int ct) {
        const size_t coff = (size_t)min(jlane + 64 * ct, p - 1) * (size_t)ldx;
#pragma unroll
        for (int h = 0; h < NH; ++h)
            R[ct % RING][h] = __builtin_nontemporal_load(reinterpret_cast<const v4u32 *>(Xc + (size_t)(i0 + 32 * h + 8 * g) + coff));
    };
    int64_t i0 = (int64_t)blockIdx.x * TH;
    if (i0 < nfull) {
        issue_b(i0);
#pragma unroll
        for (int ct = 0; ct < RING; ++ct) issue_piece(i0, ct);
    }
    auto tile = [&](auto has_next, int64_t i0) {
        take_b();
        if (decltype(has_next)::value) issue_b(i0 + istep);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const int cc = 64 * ct + 16 * wv + cidx;
            const unsigned lmask = cg0 + cc < p ? 0xffffffffu : 0u;
            bf16_t *xw = xt + 8 * g * pitch + (cc < wcols ? cc : wcols + (cc & 3));
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const v4u32 a = {R[ct % RING][h].x & lmask, R[ct % RING][h].y & lmask, R[ct % RING][h].z & lmask, R[ct % RING][h].w & lmask};
                const unsigned w4[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned w = w4[e >> 1];
                    xw[(32 * h + e) * pitch] = (bf16_t)((e & 1) ? (w >> 16) : w);
                }
                const v4f32b pr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, Bf[h]),
                                                                          v4f32b{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) acc[ct][reg] += (double)pr[reg];
            }
            if (ct + RING < NT) issue_piece(i0, ct + RING);
            else if (decltype(has_next)::value) issue_piece(i0 + istep, ct + RING - NT);
        }
        __syncthreads();   // tile complete in LDS
        {
            const int c16n = wcols >> 3;
#pragma unroll 4
            for (int row = wv; row < TH; row += 4) {
                for (int c16 = lane; c16 < c16n; c16 += 64) {
                    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(xt + row * pitch + 8 * c16);
                    const v2u64b v = {src[0], src[1]};
                    __builtin_nontemporal_store(v, reinterpret_cast<v2u64b *>(Xr + (size_t)(i0 + row) * (size_t)ldr + cg0 + 8 * c16));
                }
            }
        }
        __syncthreads();   // the tile may be overwritten
    };
    for (; i0 + istep < nfull; i0 += istep) tile(std::true_type{}, i0);
    if (i0 < nfull) tile(std::false_type{}, i0);
    // D[m][n] of v_mfma_f32_16x16x32: n = lane & 15 (y column), m = 4 (lane >> 4) + reg (x column within the wave's 16)
    double *kp = Kpart + ((size_t)blockIdx.x * kp_rows) * 16;
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int jj = 64 * ct + 16 * wv + 4 * g + reg;
            if (jj < 512 && cg0 + jj < kp_rows) kp[(size_t)(cg0 + jj) * 16 + cidx] = acc[ct][reg];
        }
    }
}
// the ragged tail (n % TH rows) of the m32 path: raw sums x' [y | 1] in f64 into one Kpart slot + the tail's rows of the copy
__global__ __launch_bounds__(256) void k_xty_bf16_tail_m32(const bf16_t *__restrict__ Xc, int64_t ldx, const bf16_t *__restrict__ Yc, int64_t ldy,
                                                           int64_t r0, int nt, int p, int q, bf16_t *__restrict__ Xr, int ldr,
                                                           double *__restrict__ kslot, int kp_rows)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= kp_rows * 16) return;
    const int j = e >> 4, k = e & 15;
    double s = 0.0;
    if (j < p && k <= q)
        for (int r = 0; r < nt; ++r) {
            const double x = (double)bf2f(Xc[(size_t)(r0 + r) + (size_t)j * (size_t)ldx]);
            s += k < q ? x * (double)bf2f(Yc[(size_t)(r0 + r) + (size_t)k * (size_t)ldy]) : x;
        }
    kslot[e] = s;
    if (k == 0 && j < ldr)
        for (int r = 0; r < nt; ++r) Xr[(size_t)(r0 + r) * (size_t)ldr + j] = j < p ? Xc[(size_t)(r0 + r) + (size_t)j * (size_t)ldx] : (bf16_t)0;
}
// K = (X'Y - (X'1) ybar') / n, means = X'1 / n from the all-reduced raw sums (column q of K = X'1; ybar at ymeans); hdr[1] = n
__global__ __launch_bounds__(256) void k_bf16_m32_fix(double *__restrict__ K, int qpad, int p, int q, const double *__restrict__ ymeans,
                                                      const double *__restrict__ hdr, double *__restrict__ means)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= p) return;
    const double inv = 1.0 / hdr[1];
    const double mj = K[(size_t)j * qpad + q] * inv;
    for (int k = 0; k < q; ++k) K[(size_t)j * qpad + k] = K[(size_t)j * qpad + k] * inv - mj * ymeans[k];
    for (int k = q; k < qpad; ++k) K[(size_t)j * qpad + k] = 0.0;
    means[j] = mj;
}

// fixed-order sum of the per-block XtY partials [nbx][kp_rows][16] (several hundred slots): 4 groups of blocks per entry with
// 4 independent chains each, combined in group order (same scheme as k_reduce_kpart_wide, prologue.hip)
__global__ __launch_bounds__(256) void k_reduce_kpart_bw(const double *__restrict__ Kpart, int nbx, int kp_rows, int p, int qpad,
                                                         double *__restrict__ K)
{
    __shared__ double sc[4][64];
    const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    const size_t stride = (size_t)kp_rows * qpad;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (e < p * qpad) {
        int b = g;
        for (; b + 12 < nbx; b += 16) {
            s0 += Kpart[(size_t)b * stride + e]; s1 += Kpart[(size_t)(b + 4) * stride + e];
            s2 += Kpart[(size_t)(b + 8) * stride + e]; s3 += Kpart[(size_t)(b + 12) * stride + e];
        }
        for (; b < nbx; b += 4) s0 += Kpart[(size_t)b * stride + e];
    }
    sc[g][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && e < p * qpad) K[e] = (sc[0][el] + sc[1][el]) + (sc[2][el] + sc[3][el]);
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
template <int KC, int R, bool PF>
__global__ __launch_bounds__(256) void k_sweep_bf16(const bf16_t *__restrict__ Xr, int64_t n, int ldr, const double *__restrict__ dw,
                                                    const double *__restrict__ rvec, const double *__restrict__ mom,
                                                    const double *__restrict__ scl, int p,
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
        for (int e = 0; e < 8; ++e) { rf[k][e] = 0.f; zp[k][e] = 0.f; }
    }
    double tt = 0.0, st = 0.0;
    const int64_t ngroups = (n + R - 1) / R;
    const int64_t gstride = (int64_t)gridDim.x * 4;
    // software pipeline (PF): the rows of the wave's NEXT group are requested before the current group is reduced — a
    // bf16 row is only 1 KB, so without it a CU has barely the bytes in flight that 8 TB/s x memory latency asks for
    v4u32 xn[R][KC];
    double dwn[R];
    auto fetch = [&](int64_t gg) {
        const int64_t r0 = gg * R;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const bool live = r0 + rr < n;
            const v4u32 *rp = reinterpret_cast<const v4u32 *>(Xr + (size_t)(r0 + rr) * (size_t)ldr) + lane;
#pragma unroll
            for (int k = 0; k < KC; ++k) xn[rr][k] = (live && in[k]) ? __builtin_nontemporal_load(rp + 64 * k) : v4u32{0u, 0u, 0u, 0u};
            dwn[rr] = live ? dw[r0 + rr] : 0.0;
        }
    };
    int64_t g = (int64_t)blockIdx.x * 4 + wv;
    if (g < ngroups) fetch(g);   // (requested before the coefficients below: one memory latency at the head of a launch, not two)
    // rt_j = r_j / s_j (fp32) and off = sum_j m_j * rt_j (fp64, from the fp32-rounded rt so that the row sums and the offset
    // use the same coefficients): every wave derives them from the replicated fp64 r — identical bits in all waves
    double offd = 0.0;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int col = 8 * lane + 512 * k;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (col + e < p) {
                const float v = (float)(rvec[col + e] / scl[col + e]);
                rf[k][e] = v;
                offd += mom[col + e] * (double)v;
            }
    }
    const float off = (float)jch_wave_sum(offd);
    for (; g < ngroups; g += gstride) {
        const int64_t row0 = g * R;
        v4u32 x[R][KC];
        double dwc[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            dwc[rr] = dwn[rr];
#pragma unroll
            for (int k = 0; k < KC; ++k) x[rr][k] = xn[rr][k];
        }
        if (PF && g + gstride < ngroups) fetch(g + gstride);
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
            const float dtf = live ? (float)dwc[rr] * t : 0.f;
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
        if (!PF && g + gstride < ngroups) fetch(g + gstride);
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

// ---------------------------------------------------------------- K4 v2 (bf16 storage): lighter instruction stream
// k_sweep_bf16<1, 4> issues ~400 instructions per 4 KB wave-iteration (177 register copies of the prefetch buffer, a
// branch pair around every load, 19 ds_bpermute, per-row f32 <-> f64 conversions): at 3 waves per SIMD that is as long as
// the memory time of those bytes — the bf16 sweep was issue-bound at 5.8 TB/s.  Same recipe as k_sweep_v2 (sweep.hip):
// R = 8 rows per wave-iteration in two rotating register buffers, unconditional clamped loads, row sums folded with
// v_permlane32_swap / v_permlane16_swap + DPP (one dword per value in fp32), tt / st accumulated in fp32 inside an
// iteration and in fp64 across iterations, first rows requested before the coefficients.
typedef unsigned v2u32b __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bf_fold32(float &a, float b)
{
    const v2u32b r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ void bf_fold16(float &a, float b)
{
    const v2u32b r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x) + __uint_as_float(r.y);
}
template <int CTRL>
__device__ __forceinline__ float bf_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// every lane of the 8-lane group g = lane >> 3 ends up with the 64-lane total of row rowmap(g) (see jch_rowsums, sweep.hip)
template <int R>
__device__ __forceinline__ float bf_rowsums(float (&s)[R], int lane)
{
    static_assert(R == 4 || R == 8, "R");
#pragma unroll
    for (int i = 0; i < R; i += 2) bf_fold32(s[i], s[i + 1]);
#pragma unroll
    for (int i = 0; i < R; i += 4) bf_fold16(s[i], s[i + 2]);
    float h;
    if (R == 8) {
        const bool up = (lane & 8) != 0;
        const float w = up ? s[4] : s[0], z = up ? s[0] : s[4];
        h = w + bf_dpp<0x128>(z);
    } else {
        h = s[0] + bf_dpp<0x128>(s[0]);
    }
    h += bf_dpp<0x141>(h);
    h += bf_dpp<0xB1>(h);
    h += bf_dpp<0x4E>(h);
    return h;
}
template <int R>
__device__ __forceinline__ constexpr int bf_rowsum_lane(int rr)
{
    return 16 * (((rr & 3) == 1) ? 2 : ((rr & 3) == 2) ? 1 : (rr & 3)) + (R == 8 ? 8 * (rr >> 2) : 0);
}

template <int KC, int R, int NBUF = 2>   // NBUF register buffers in rotation: NBUF - 1 row groups in flight behind the one being reduced
__global__ __launch_bounds__(256) void k_sweep_bf16_v2(const bf16_t *__restrict__ Xr, int64_t n, int ldr, const double *__restrict__ dw,
                                                       const double *__restrict__ rvec, const double *__restrict__ mom,
                                                       const double *__restrict__ scl, int p,
                                                       double *__restrict__ tcol, double *__restrict__ part, int ldpart)
{
    extern __shared__ __attribute__((aligned(16))) double red[];  // [4][KC*512] + [8]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int coff[KC];   // element offset of this lane's 8 columns in chunk k, clamped to the row's last 16 bytes
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int col = 8 * lane + 512 * k;
        coff[k] = col < ldr ? col : ldr - 8;
    }
    const int64_t ngroups = (n + R - 1) / R;
    const int64_t gstride = (int64_t)gridDim.x * 4;
    v4u32 X[NBUF][R][KC];
    double D[NBUF][R];   // (kept as loaded: converting here would make the prefetch wait for its own loads)
    auto fetch = [&](v4u32 (&xb)[R][KC], double (&db)[R], int64_t gg) {
        const int64_t r0 = gg * R;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int64_t row = r0 + rr < n ? r0 + rr : n - 1;   // wave-uniform clamp; the row gets weight 0 below
            const bf16_t *rp = Xr + (size_t)row * (size_t)ldr;
#pragma unroll
            for (int k = 0; k < KC; ++k) xb[rr][k] = __builtin_nontemporal_load(reinterpret_cast<const v4u32 *>(rp + coff[k]));
            db[rr] = dw[row];
        }
    };
    int64_t g = (int64_t)blockIdx.x * 4 + wv;
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b)
        if (g + b * gstride < ngroups) fetch(X[b], D[b], g + b * gstride);
    // rt_j = r_j / s_j (fp32) and off = sum_j m_j * rt_j (fp64, from the fp32-rounded rt): see k_sweep_bf16
    float rf[KC][8], zp[KC][8];
    double offd = 0.0;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int col = 8 * lane + 512 * k;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            rf[k][e] = 0.f; zp[k][e] = 0.f;
            if (col + e < p) {
                const float v = (float)(rvec[col + e] / scl[col + e]);
                rf[k][e] = v;
                offd += mom[col + e] * (double)v;
            }
        }
    }
    const float off = (float)jch_wave_sum(offd);
    double tt = 0.0, st = 0.0;
    auto process = [&](v4u32 (&x)[R][KC], double (&dv)[R], int64_t gg) {
        const int64_t row0 = gg * R;
        float sp[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < KC; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) a += bflo(x[rr][k][e]) * rf[k][2 * e] + bfhi(x[rr][k][e]) * rf[k][2 * e + 1];
            sp[rr] = a;
        }
        const float h = bf_rowsums<R>(sp, lane) - off;
        float ttg = 0.f, stg = 0.f;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const float t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h), bf_rowsum_lane<R>(rr)));
            const float dtf = row0 + rr < n ? (float)dv[rr] * t : 0.f;
            ttg += dtf * t;
            stg += dtf;
#pragma unroll
            for (int k = 0; k < KC; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    zp[k][2 * e] += dtf * bflo(x[rr][k][e]);
                    zp[k][2 * e + 1] += dtf * bfhi(x[rr][k][e]);
                }
        }
        tt += (double)ttg;
        st += (double)stg;
        {
            const int src = 16 * (((lane & 3) == 1) ? 2 : ((lane & 3) == 2) ? 1 : (lane & 3)) + (R == 8 ? 8 * ((lane >> 2) & 1) : 0);
            const float tl = __shfl(h, src, 64);
            if (lane < R && row0 + lane < n) tcol[row0 + lane] = (double)tl;
        }
    };
    while (g < ngroups) {
#pragma unroll
        for (int b = 0; b < NBUF; ++b) {
            if (g < ngroups) {
                const int64_t ahead = g + (NBUF - 1) * gstride;
                if (ahead < ngroups) fetch(X[(b + NBUF - 1) % NBUF], D[(b + NBUF - 1) % NBUF], ahead);
                process(X[b], D[b], g);
                g += gstride;
            }
        }
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

// zp_j <- (zp_raw_j - m_j * st) / s_j ;  slot [ldz_tt] <- tt      (after the cross-GPU all-reduce)
// (zt holds nslice partial slices of [zp_raw (ldr_b), tt, st], ld ldzb: summed here in fixed order)
__global__ __launch_bounds__(256) void k_bf16_fix_zt(const double *__restrict__ zt, int nslice, int ldzb, int ldr_b, int p, int ldr_small,
                                                     const double *__restrict__ mom, const double *__restrict__ scl,
                                                     double *__restrict__ zt_small)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    double tt = 0.0, st = 0.0, z = 0.0;
    for (int sl = 0; sl < nslice; ++sl) {
        tt += zt[(size_t)sl * ldzb + ldr_b];
        st += zt[(size_t)sl * ldzb + ldr_b + 1];
        if (j < p) z += zt[(size_t)sl * ldzb + j];
    }
    if (j < ldr_small) zt_small[j] = j < p ? (z - mom[j] * st) / scl[j] : 0.0;
    if (j == 0) zt_small[ldr_small] = tt;
}

template <int KC, int R, bool PF = true>
static int32_t launch_sweep_bf16_t(jch_ctx *ctx, const bf16_t *Xr, int64_t n, int ldr_b, const double *d, const double *rvec,
                                   const double *mom, const double *scl, int p, double *tcol, double *zt8, int ldzb, int *nslice)
{
    const size_t lds = sizeof(double) * (4 * KC * 512 + 8);
    static int bpc = 0;
    static jch_per_device_once occ_once;
    if (!occ_once.done(ctx->device)) {
        int nblk = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_sweep_bf16<KC, R, PF>, 256, lds);
        bpc = (e == hipSuccess && nblk > 0) ? nblk : 2;
        if (lds > 64 * 1024)
            JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_sweep_bf16<KC, R, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        occ_once.mark(ctx->device);
    }
    const int64_t ngroups = (n + R - 1) / R;
    static int bpc_env = -1;
    if (bpc_env < 0) { const char *e = getenv("JCH_BF16_BPC"); bpc_env = e ? atoi(e) : 0; }
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((ngroups + 3) / 4, (int64_t)ctx->cus * (bpc_env > 0 ? bpc_env : bpc)));
    const int m = ldr_b + 2, ldpart = (m + 7) & ~7;
    JCH_TRY(jch_reserve(ctx, ctx->part, sizeof(double) * (size_t)nb * ldpart));
    double *part = (double *)ctx->part.ptr;
    (void)jch_ev(ctx);
    hipLaunchKernelGGL((k_sweep_bf16<KC, R, PF>), dim3(nb), dim3(256), lds, ctx->stream, Xr, n, ldr_b, d, rvec, mom, scl, p, tcol, part, ldpart);
    (void)jch_ev(ctx);
    JCH_TRY(jch_launch_reduce_part8(ctx, part, nb, ldpart, m, zt8, ldzb, nslice));
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

template <int KC, int R, int NBUF = 2>
static int32_t launch_sweep_bf16_v2_t(jch_ctx *ctx, const bf16_t *Xr, int64_t n, int ldr_b, const double *d, const double *rvec,
                                      const double *mom, const double *scl, int p, double *tcol, double *zt8, int ldzb, int *nslice,
                                      jch_part_view *pv = nullptr /*split small-state path on one rank / with the per-block inbox: leave the block partials unreduced*/)
{
    const size_t lds = sizeof(double) * (4 * KC * 512 + 8);
    static int bpc = 0;
    static jch_per_device_once occ_once;
    if (!occ_once.done(ctx->device)) {
        int nblk = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_sweep_bf16_v2<KC, R, NBUF>, 256, lds);
        bpc = (e == hipSuccess && nblk > 0) ? nblk : 2;
        if (lds > 64 * 1024)
            JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_sweep_bf16_v2<KC, R, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        occ_once.mark(ctx->device);
    }
    const int64_t ngroups = (n + R - 1) / R;
    const char *eb = getenv("JCH_BF16_BPC");
    const int use_bpc = (eb && atoi(eb) > 0) ? atoi(eb) : bpc;
    int nb = (int)std::max<int64_t>(1, std::min<int64_t>((ngroups + 3) / 4, (int64_t)ctx->cus * use_bpc));
    if (const char *e_nb = getenv("JCH_SWEEP_NB")) { const int v = atoi(e_nb); if (v > 0 && v < nb) nb = v; }   // (A/B runs: the grid)
    const int m = ldr_b + 2, ldpart = (m + 7) & ~7;
    JCH_TRY(jch_reserve(ctx, ctx->part, sizeof(double) * (size_t)nb * ldpart));
    double *part = (double *)ctx->part.ptr;
    const bool timed = jch_prof_sample(ctx);
    if (timed) (void)jch_ev(ctx);
    hipLaunchKernelGGL((k_sweep_bf16_v2<KC, R, NBUF>), dim3(nb), dim3(256), lds, ctx->stream, Xr, n, ldr_b, d, rvec, mom, scl, p, tcol, part, ldpart);
    if (timed) (void)jch_ev(ctx);
    if (pv) { pv->part = part; pv->nb = nb; pv->ldpart = ldpart; *nslice = 1; JCH_HIP(ctx, hipGetLastError()); return JCH_OK; }
    JCH_TRY(jch_launch_reduce_part8(ctx, part, nb, ldpart, m, zt8, ldzb, nslice));
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

// raw prologue helpers: pivot = mean of a strided sample of (up to) 256 rows per shard, the ranks' sample means weighted by
// n_r / n_total (hdr[1]) — the same rule as the f64 path (prologue.hip k_pivot_rows); means = pivot + K[:, col].  Here the
// pivot only conditions K = (X - c)'D Yc (f64 arithmetic on the exact bf16 values); the sweeps use the exact means.
__global__ __launch_bounds__(256) void k_pivot_rows_bf16(const bf16_t *__restrict__ Xc, int64_t ldx, int64_t n, int p,
                                                         const double *__restrict__ hdr, double *__restrict__ pivot)
{
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= p) return;
    const int m = (int)(n < 256 ? n : 256);
    const int64_t stride = n / m;
    double s = 0.0;
    for (int i = lane; i < m; i += 64) s += (double)bf2f(Xc[(size_t)i * (size_t)stride + (size_t)j * (size_t)ldx]);
    s = jch_wave_sum(s);
    if (lane == 0) pivot[j] = s / m * ((double)n / hdr[1]);
}
__global__ __launch_bounds__(256) void k_extract_means_b(double *__restrict__ K, int qpad, int p, int col, const double *__restrict__ pivot,
                                                         double *__restrict__ means)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < p) { means[j] = pivot[j] + K[(size_t)j * qpad + col]; K[(size_t)j * qpad + col] = 0.0; }
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
    const int ldzb = (ldr_b + 2 + 7) & ~7;
    JCH_TRY(jch_reserve(ctx, ctx->gemm_b, sizeof(double) * (size_t)JCH_ZT_SLICES * ldzb));
    double *zt8 = (double *)ctx->gemm_b.ptr;           // [JCH_ZT_SLICES][ldzb]: partial slices of [zp_raw, tt, st]
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
    // RAW prologue (no scaling, 16-byte kernels, q <= 15): as in the f64 path (fit.hip) the means pass over X is dropped —
    // K2 subtracts a pivot (mean of a strided row sample) instead of the means and multiplies against [Yc | 1], so
    // the means come out of the XtY pass: mu = pivot + K[:, q].
    const bool raw_b = !d.scal && q <= 15 && ldx % 8 == 0 && ((uintptr_t)Xc) % 16 == 0 && !getenv("JCH_BF16_SCALAR_PROLOGUE") &&
                       !getenv("JCH_CENTRED_COPY");
    if (raw_b) {
        hipLaunchKernelGGL(k_pivot_rows_bf16, dim3((p + 3) / 4), dim3(256), 0, ctx->stream, Xc, ldx, n, p, s.hdr, s.scl);
        JCH_TRY(jch_allreduce_f64(ctx, s.scl, (size_t)p));
        JCH_TRY(launch_moments_bf16(ctx, Yc, ldy, Yc, ldy, dn, n, q, 0, nullptr, s.scl + p));   // Y means -> scl[p..p+q)
    } else {
    JCH_TRY(launch_moments_bf16(ctx, Xc, ldx, Yc, ldy, dn, n, p, q, nullptr, s.mom));
    if (d.scal) JCH_TRY(launch_moments_bf16(ctx, Xc, ldx, Yc, ldy, dn, n, p, q, s.mom, s.scl));
    else hipLaunchKernelGGL(k_fill_b, dim3((p + q + 255) / 256), dim3(256), 0, ctx->stream, s.scl, p + q, 1.0);
    }
    bool m32 = false;
    {
        const int ptiles = (ldr_b + 63) / 64, kp_rows = ptiles * 64, ygroups = qpad / 16;
        const int64_t nchunks = (n + 63) / 64;
        static int k2skip = -1;
        if (k2skip < 0) { const char *e = getenv("JCH_K2_SKIP"); k2skip = e ? atoi(e) : 0; }
        static int k2bpc = -1;
        if (k2bpc < 0) { const char *e = getenv("JCH_BF16_K2_BPC"); k2bpc = e ? atoi(e) : 3; }
        int nbx = std::max(1, (ctx->cus * k2bpc + ptiles * ygroups - 1) / (ptiles * ygroups));
        if (nbx > nchunks) nbx = (int)std::max<int64_t>(nchunks, 1);
        JCH_TRY(jch_reserve(ctx, ctx->kpart, sizeof(double) * (size_t)nbx * kp_rows * qpad));
        double *Kpart = (double *)ctx->kpart.ptr;
        dim3 grid(nbx, ptiles, ygroups);
        const bool v8 = ldx % 8 == 0 && ((uintptr_t)Xc) % 16 == 0 && !getenv("JCH_BF16_SCALAR_PROLOGUE");
        // row-panel kernel (round 3, default): q <= 16, 16-B aligned X / Y columns and weights, at least one full row tile
        // (measurement knobs, read on every call so that one process can compare the variants)
        const char *e_pn = getenv("JCH_BF16_K2_PANEL"), *e_nh = getenv("JCH_BF16_K2_NH"), *e_pb = getenv("JCH_BF16_K2_PBPC");
        const int k2panel = e_pn ? atoi(e_pn) : 1, k2nh = e_nh ? atoi(e_nh) : 2, k2pbpc = e_pb ? atoi(e_pb) : 0;
        const int NHv = k2nh == 4 ? 4 : 2, THv = 32 * NHv;
        const int64_t nfull = (n / THv) * THv;
        const bool panel = k2panel && v8 && qpad == 16 && ldy % 8 == 0 && ((uintptr_t)Yc) % 16 == 0 && ((uintptr_t)dn) % 16 == 0 && nfull > 0;
        // unit weights, raw mode: the products on the bf16 matrix pipe (k_xty_bf16_panel_m32; JCH_BF16_K2_M32=0: the f64 products)
        const char *e_m32 = getenv("JCH_BF16_K2_M32");
        m32 = panel && raw_b && !wdev && !(e_m32 && atoi(e_m32) == 0) && NHv == 2;
        if (m32) {
            const int ncg = (ldr_b + 511) / 512;
            const int wmax = std::min(ldr_b, 512), ntile = (wmax + 63) / 64;
            const int NTv = ntile <= 1 ? 1 : (ntile <= 2 ? 2 : (ntile <= 4 ? 4 : 8));
            const int bpc = k2pbpc > 0 ? k2pbpc : 2;
            int G = std::max(1, ctx->cus * bpc / ncg);
            G = (int)std::min<int64_t>(G, nfull / THv);
            const int kpr = ncg * 512;
            const bool tail = nfull < n;
            JCH_TRY(jch_reserve(ctx, ctx->kpart, sizeof(double) * (size_t)(G + 1) * kpr * 16));
            Kpart = (double *)ctx->kpart.ptr;
            const size_t lds = sizeof(bf16_t) * (size_t)THv * 516 + 16;
#define JCH_K2M(NT) do { \
                static jch_per_device_once once_; \
                if (!once_.done(ctx->device)) { JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_xty_bf16_panel_m32<NT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once_.mark(ctx->device); } \
                hipLaunchKernelGGL((k_xty_bf16_panel_m32<NT, 2>), dim3(G, ncg), dim3(256), lds, ctx->stream, Xc, ldx, Yc, ldy, nfull, p, q, Xr, ldr_b, Kpart, kpr); } while (0)
            if (NTv == 1) JCH_K2M(1); else if (NTv == 2) JCH_K2M(2); else if (NTv == 4) JCH_K2M(4); else JCH_K2M(8);
#undef JCH_K2M
            JCH_HIP(ctx, hipGetLastError());
            if (tail)
                hipLaunchKernelGGL(k_xty_bf16_tail_m32, dim3((kpr * 16 + 255) / 256), dim3(256), 0, ctx->stream, Xc, ldx, Yc, ldy, nfull, (int)(n - nfull), p, q,
                                   Xr, ldr_b, Kpart + (size_t)G * kpr * 16, kpr);
            hipLaunchKernelGGL(k_reduce_kpart_bw, dim3((p * qpad + 63) / 64), dim3(256), 0, ctx->stream, Kpart, G + (tail ? 1 : 0), kpr, p, qpad, s.K);
        } else
        if (panel) {
            const int ncg = (ldr_b + 511) / 512;                       // 512-column groups (blockIdx.y)
            const int wmax = std::min(ldr_b, 512), ntile = (wmax + 63) / 64;
            const int NTv = ntile <= 1 ? 1 : (ntile <= 2 ? 2 : (ntile <= 4 ? 4 : 8));
            const int bpc = k2pbpc > 0 ? k2pbpc : (NHv == 2 ? 2 : 1);
            int G = std::max(1, ctx->cus * bpc / ncg);
            G = (int)std::min<int64_t>(G, nfull / THv);
            const int kpr = ncg * 512;
            const bool tail = nfull < n;
            JCH_TRY(jch_reserve(ctx, ctx->kpart, sizeof(double) * (size_t)(G + 1) * kpr * 16));
            Kpart = (double *)ctx->kpart.ptr;
            const size_t lds = sizeof(double) * (8 * NHv * 64 + (d.scal ? 1024 : 512)) + sizeof(bf16_t) * (size_t)THv * 516 + 16;
            const double *momp = raw_b ? s.scl : s.mom;
            const int onesc = raw_b ? q : -1;
#define JCH_K2PB_S(NT, NH, SC, SK) do { \
                static jch_per_device_once once_; \
                if (!once_.done(ctx->device)) { JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_center_xty_bf16_panel<NT, NH, SC, SK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once_.mark(ctx->device); } \
                hipLaunchKernelGGL((k_center_xty_bf16_panel<NT, NH, SC, SK>), dim3(G, ncg), dim3(256), lds, ctx->stream, Xc, ldx, Yc, ldy, dn, nfull, p, q, momp, s.scl, \
                                   Xr, ldr_b, Yr, Kpart, kpr, onesc); } while (0)
#define JCH_K2PB(NT, NH, SC) JCH_K2PB_S(NT, NH, SC, 0)
#define JCH_K2PB_NT(NH, SC) do { if (NTv == 1) JCH_K2PB(1, NH, SC); else if (NTv == 2) JCH_K2PB(2, NH, SC); else if (NTv == 4) JCH_K2PB(4, NH, SC); else JCH_K2PB(8, NH, SC); } while (0)
            if (NHv == 2 && NTv == 8 && !d.scal && (k2skip & 3)) {   // measurement variants of the headline instantiation
                if ((k2skip & 3) == 1) JCH_K2PB_S(8, 2, false, 1); else if ((k2skip & 3) == 2) JCH_K2PB_S(8, 2, false, 2); else JCH_K2PB_S(8, 2, false, 3);
            } else
            if (NHv == 2) { if (d.scal) JCH_K2PB_NT(2, true); else JCH_K2PB_NT(2, false); }
            else { if (d.scal) JCH_K2PB_NT(4, true); else JCH_K2PB_NT(4, false); }
#undef JCH_K2PB_NT
#undef JCH_K2PB
#undef JCH_K2PB_S
            JCH_HIP(ctx, hipGetLastError());
            if (tail) {   // the last n % TH rows: the tile kernel on that row range, its partial in slot G
                const int64_t nt = n - nfull;
                dim3 tgrid(1, ptiles, 1);
                double *kslot = Kpart + (size_t)G * kpr * 16;
                if (d.scal) hipLaunchKernelGGL(k_center_xty_bf16_v8<true>, tgrid, dim3(256), 0, ctx->stream, Xc + nfull, ldx, Yc + nfull, ldy, dn + nfull, nt, p, q,
                                               s.mom, s.scl, Xr + (size_t)nfull * ldr_b, ldr_b, Yr + (size_t)nfull * qpad, qpad, kslot, kpr, -1, k2skip);
                else hipLaunchKernelGGL(k_center_xty_bf16_v8<false>, tgrid, dim3(256), 0, ctx->stream, Xc + nfull, ldx, Yc + nfull, ldy, dn + nfull, nt, p, q,
                                        momp, s.scl, Xr + (size_t)nfull * ldr_b, ldr_b, Yr + (size_t)nfull * qpad, qpad, kslot, kpr, onesc, k2skip);
            }
            hipLaunchKernelGGL(k_reduce_kpart_bw, dim3((p * qpad + 63) / 64), dim3(256), 0, ctx->stream, Kpart, G + (tail ? 1 : 0), kpr, p, qpad, s.K);
        } else
        if (v8 && d.scal) hipLaunchKernelGGL(k_center_xty_bf16_v8<true>, grid, dim3(256), 0, ctx->stream, Xc, ldx, Yc, ldy, dn, n, p, q, s.mom,
                                             s.scl, Xr, ldr_b, Yr, qpad, Kpart, kp_rows, -1, k2skip);
        else if (v8) hipLaunchKernelGGL(k_center_xty_bf16_v8<false>, grid, dim3(256), 0, ctx->stream, Xc, ldx, Yc, ldy, dn, n, p, q,
                                        raw_b ? s.scl : s.mom, s.scl, Xr, ldr_b, Yr, qpad, Kpart, kp_rows, raw_b ? q : -1, k2skip);
        else if (d.scal) hipLaunchKernelGGL(k_center_xty_bf16<true>, grid, dim3(256), 0, ctx->stream, Xc, ldx, Yc, ldy, dn, n, p, q, s.mom,
                                       s.scl, Xr, ldr_b, Yr, qpad, Kpart, kp_rows);
        else hipLaunchKernelGGL(k_center_xty_bf16<false>, grid, dim3(256), 0, ctx->stream, Xc, ldx, Yc, ldy, dn, n, p, q, s.mom,
                                s.scl, Xr, ldr_b, Yr, qpad, Kpart, kp_rows);
        if (!panel) hipLaunchKernelGGL(k_reduce_kpart_b, dim3((p * qpad + 255) / 256), dim3(256), 0, ctx->stream, Kpart, nbx, kp_rows, p, qpad, s.K);
        JCH_TRY(jch_allreduce_f64(ctx, s.K, (size_t)p * qpad));
        if (raw_b) {   // means = pivot + K[:, q] (m32: centring of the raw sums); Y means next to them; divisors = 1
            if (m32) hipLaunchKernelGGL(k_bf16_m32_fix, dim3((p + 255) / 256), dim3(256), 0, ctx->stream, s.K, qpad, p, q, s.scl + p, s.hdr, s.mom);
            else
            hipLaunchKernelGGL(k_extract_means_b, dim3((p + 255) / 256), dim3(256), 0, ctx->stream, s.K, qpad, p, q, s.scl, s.mom);
            JCH_HIP(ctx, hipMemcpyAsync(s.mom + p, s.scl + p, sizeof(double) * (size_t)q, hipMemcpyDeviceToDevice, ctx->stream));
            hipLaunchKernelGGL(k_fill_b, dim3((p + q + 255) / 256), dim3(256), 0, ctx->stream, s.scl, p + q, 1.0);
        }
    }
    (void)jch_ev(ctx);  // end of prologue
    ctx->ev_mark = ctx->ev_used;  // (begin, end) event pairs of the sweeps start here
    ctx->coll_phase = 1;          // all-reduces from here on belong to the LV loop (profile: collective_ms)
    // ---- LV loop
    JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr_small, -1, nlv, 0, 1, ldz, fast));
    for (int a = 0; a < nlv; ++a) {
        double *tcol = Tdev + (size_t)a * (size_t)n;
        static int rsel = -1;
        if (rsel < 0) { const char *e = getenv("JCH_BF16_R"); rsel = e ? atoi(e) : 4; }   // measured at n = 1e6, p = 500 with the prefetch: R = 2 / 4 / 8 -> 5.35 / 5.84 / 5.40 TB/s (without: 4.67 / 5.07)
        int nslice = 1;
        jch_part_view pv;
        const bool fuse_now = fast && ctx->p2p.ready && !ctx->loop && !getenv("JCH_P2P_UNFUSED") && (size_t)(ldr_b + 2) <= ctx->p2p.cap;
        jch_part_view *pvp = (fast && s.kr && (ctx->nranks == 1 || fuse_now)) ? &pv : nullptr;   // split path: k_lv_spread sums the block partials
        {   // v2 kernels (permlane row sums, rotating buffers); JCH_BF16_V2=0 selects the round-1 kernels below (read per call)
            const char *e2 = getenv("JCH_BF16_V2");
            const int v2 = e2 ? atoi(e2) : 1;
            if (v2 && ldr_b >= 8 && ldr_b <= 1024) {
                // (Measured, round 4: deeper rotations of the row-group buffers, NBUF template parameter — <1,8,3> 221-225 us per launch at
                // n = 1e6 (one wave per SIMD), <1,4,4> 196-201, <1,4,3> 164-168 against 160-168 for the default <1,8,2>; n = 8e6: 1 575 /
                // 1 338 against 1 282 — more bytes in flight per wave do not make up for the waves they cost.)
                if (ldr_b <= 512) { if (v2 == 4) JCH_TRY((launch_sweep_bf16_v2_t<1, 4>(ctx, Xr, n, ldr_b, dn, s.r, s.mom, s.scl, p, tcol, zt8, ldzb, &nslice, pvp)));
                                    else JCH_TRY((launch_sweep_bf16_v2_t<1, 8>(ctx, Xr, n, ldr_b, dn, s.r, s.mom, s.scl, p, tcol, zt8, ldzb, &nslice, pvp))); }
                else JCH_TRY((launch_sweep_bf16_v2_t<2, 8>(ctx, Xr, n, ldr_b, dn, s.r, s.mom, s.scl, p, tcol, zt8, ldzb, &nslice, pvp)));
                goto swept;
            }
        }
#define JCH_SWB(KC, R) JCH_TRY((launch_sweep_bf16_t<KC, R>(ctx, Xr, n, ldr_b, dn, s.r, s.mom, s.scl, p, tcol, zt8, ldzb, &nslice)))
        static int pfsel = -1;
        if (pfsel < 0) { const char *e = getenv("JCH_BF16_PF"); pfsel = e ? atoi(e) : 1; }
        if (ldr_b <= 512 && rsel == 4 && !pfsel) JCH_TRY((launch_sweep_bf16_t<1, 4, false>(ctx, Xr, n, ldr_b, dn, s.r, s.mom, s.scl, p, tcol, zt8, ldzb, &nslice)));
        else if (ldr_b <= 512 && rsel == 2 && !pfsel) JCH_TRY((launch_sweep_bf16_t<1, 2, false>(ctx, Xr, n, ldr_b, dn, s.r, s.mom, s.scl, p, tcol, zt8, ldzb, &nslice)));
        else if (ldr_b <= 512 && rsel == 4) JCH_SWB(1, 4);
        else if (ldr_b <= 512 && rsel == 2) JCH_SWB(1, 2);
        else if (ldr_b <= 512) JCH_SWB(1, 8);
        else if (ldr_b <= 1024) JCH_SWB(2, 4);
        else JCH_SWB(4, 2);
#undef JCH_SWB
    swept:;
        if (ctx->nranks > 1) nslice = JCH_ZT_SLICES;                      // rank-independent message size (unused slices hold zeros)
        // ONE collective per LV: [zp_raw, tt, st].  Fast small-state kernel: it adds the slices, (with the inbox transport)
        // all-reduces them and applies the centring / scaling fix-up itself; generic kernel: separate steps.
        if (fast) {
            const bool fuse = ctx->p2p.ready && !ctx->loop && !getenv("JCH_P2P_UNFUSED") && (size_t)(ldr_b + 2) <= ctx->p2p.cap;
            if (!fuse && !pv.part) JCH_TRY(jch_allreduce_slices(ctx, zt8, ldr_b + 2, nslice, ldzb, &nslice));
            else if (fuse) ctx->coll_transport = JCH_TRANSPORT_INBOX_FUSED;
            if (s.kr) {   // split small-state path (smallstate_split.hip): the partial rows / slices are summed (and, with the inbox, exchanged) by its p-parallel kernel
                if (!pv.part) { pv.part = zt8; pv.nb = nslice; pv.ldpart = ldzb; }
                JCH_TRY(jch_launch_lv_split(ctx, s, p, q, ldr_small, a, nlv, pv.part, pv.nb, pv.ldpart, ldr_b, ldr_b + 1, 2, a + 1 < nlv, fuse));
            }
            else
            JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr_small, a, nlv, 0, nslice, ldz, true, fuse, zt8, ldzb, ldr_b));
        } else {
            JCH_TRY(jch_allreduce_slices(ctx, zt8, ldr_b + 2, nslice, ldzb, &nslice));
            hipLaunchKernelGGL(k_bf16_fix_zt, dim3((ldr_small + 255) / 256), dim3(256), 0, ctx->stream, zt8, nslice, ldzb, ldr_b, p, ldr_small,
                               s.mom, s.scl, s.zt);
            JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr_small, a, nlv, 0, 1, ldz, false));
        }
    }
    JCH_HIP(ctx, hipGetLastError());
    *nlv_out = nlv;
    return JCH_OK;
}
