// Prologue kernels of a fit (src/plskern.jl:114-132 == src/plsnipals.jl:39-56):
//   K0  weight normalisation            mweight            utility.jl:715-723
//   K1  weighted column means / vars    colmean, colvar    utility.jl:195, 314-323 (two-pass variance)
//   K2  centre/scale + re-layout + XtY  center!/cscale!    utility.jl:76-81, 482-487; plskern.jl:131-132
// Input X, Y are column-major (Julia); the working copy Xr is ROW-major (ld = ldr, even, pad column zero)
// so the per-LV sweep streams whole rows.  K2 is the one real tall-skinny GEMM of the fit and runs on
// v_mfma_f64_16x16x4_f64 (A = X tile^T, B = d.*Y tile, N = q padded to 16).
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "jch_internal.h"
#include "rowsum_dev.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- K0: weights
__global__ __launch_bounds__(256) void k_wsum_part(const double *__restrict__ w, int64_t n, double *__restrict__ part)
{
    __shared__ double sc[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += w[i];
    s = jch_block_sum<256>(s, sc);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_wsum_final(const double *__restrict__ part, int nb, int64_t n, int have_w,
                                                    double *__restrict__ hdr, double *__restrict__ zero0, int nzero0,
                                                    double *__restrict__ zero1, int nzero1)
{
    __shared__ double sc[4];
    // (small per-fit state that has to start at zero rides along: one launch instead of a memset each)
    for (int i = threadIdx.x; i < nzero0; i += 256) zero0[i] = 0.0;
    for (int i = threadIdx.x; i < nzero1; i += 256) zero1[i] = 0.0;
    double s = 0.0;
    if (have_w)
        for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
    s = jch_block_sum<256>(s, sc);
    if (threadIdx.x == 0) {
        hdr[0] = have_w ? s : (double)n;  // sum of weights of this shard
        hdr[1] = (double)n;               // rows of this shard
    }
}

__global__ __launch_bounds__(256) void k_wnorm(const double *__restrict__ w, int64_t n, const double *__restrict__ hdr,
                                               double *__restrict__ d)
{
    const double sw = hdr[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        d[i] = (w ? w[i] : 1.0) / sw;
}

int32_t jch_launch_weights(jch_ctx *ctx, const double *w, int64_t n, double *dnorm, double *hdr, double *zero0, int nzero0,
                           double *zero1, int nzero1)
{
    const int nb = (int)std::min<int64_t>(1024, (n + 255) / 256 > 0 ? (n + 255) / 256 : 1);
    JCH_TRY(jch_reserve(ctx, ctx->colpart, sizeof(double) * 4096));
    double *part = (double *)ctx->colpart.ptr;
    if (w) hipLaunchKernelGGL(k_wsum_part, dim3(nb), dim3(256), 0, ctx->stream, w, n, part);
    hipLaunchKernelGGL(k_wsum_final, dim3(1), dim3(256), 0, ctx->stream, part, nb, n, w ? 1 : 0, hdr, zero0, nzero0, zero1, nzero1);
    JCH_TRY(jch_allreduce_f64(ctx, hdr, 2));  // global sum of weights, global row count
    hipLaunchKernelGGL(k_wnorm, dim3(nb), dim3(256), 0, ctx->stream, w, n, hdr, dnorm);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

// ---------------------------------------------------------------- K1: weighted column moments
// grid (p+q, S): block (j, s) reduces rows [s*chunk, (s+1)*chunk) of column j.
template <bool VAR>
__global__ __launch_bounds__(256) void k_moments(const double *__restrict__ Xc, int64_t ldx, const double *__restrict__ Yc,
                                                  int64_t ldy, const double *__restrict__ d, int64_t n, int p, int q,
                                                  int64_t chunk, const double *__restrict__ means,
                                                  double *__restrict__ colpart, int aligned16)
{
    __shared__ double sc[4];
    const int j = blockIdx.x;
    const double *col = j < p ? Xc + (size_t)j * (size_t)ldx : Yc + (size_t)(j - p) * (size_t)ldy;
    const int64_t i0 = (int64_t)blockIdx.y * chunk;
    const int64_t i1 = i0 + chunk < n ? i0 + chunk : n;
    const double m = VAR ? means[j] : 0.0;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (aligned16) {   // 16-B loads: two consecutive rows per lane
        typedef double v2 __attribute__((ext_vector_type(2)));
        int64_t i = i0 + 2 * threadIdx.x;
        for (; i + 1536 + 1 < i1; i += 2048) {
            v2 a0 = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(col + i)), a1 = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(col + i + 512)),
               a2 = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(col + i + 1024)), a3 = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(col + i + 1536));
            const v2 d0 = *reinterpret_cast<const v2 *>(d + i), d1 = *reinterpret_cast<const v2 *>(d + i + 512),
                     d2 = *reinterpret_cast<const v2 *>(d + i + 1024), d3 = *reinterpret_cast<const v2 *>(d + i + 1536);
            if (VAR) { a0 -= m; a1 -= m; a2 -= m; a3 -= m; a0 *= a0; a1 *= a1; a2 *= a2; a3 *= a3; }
            s0 += d0.x * a0.x; s0 += d0.y * a0.y; s1 += d1.x * a1.x; s1 += d1.y * a1.y;
            s2 += d2.x * a2.x; s2 += d2.y * a2.y; s3 += d3.x * a3.x; s3 += d3.y * a3.y;
        }
        for (; i < i1; i += 512) {
            double a0 = col[i], a1 = i + 1 < i1 ? col[i + 1] : 0.0;
            if (VAR) { a0 -= m; a0 *= a0; a1 = i + 1 < i1 ? (a1 - m) * (a1 - m) : 0.0; }
            s0 += d[i] * a0;
            if (i + 1 < i1) s0 += d[i + 1] * a1;
        }
    } else {
    int64_t i = i0 + threadIdx.x;
    for (; i + 768 < i1; i += 1024) {
        double a0 = __builtin_nontemporal_load(col + i), a1 = __builtin_nontemporal_load(col + i + 256),
               a2 = __builtin_nontemporal_load(col + i + 512), a3 = __builtin_nontemporal_load(col + i + 768);
        double d0 = d[i], d1 = d[i + 256], d2 = d[i + 512], d3 = d[i + 768];
        if (VAR) { a0 -= m; a1 -= m; a2 -= m; a3 -= m; a0 *= a0; a1 *= a1; a2 *= a2; a3 *= a3; }
        s0 += d0 * a0; s1 += d1 * a1; s2 += d2 * a2; s3 += d3 * a3;
    }
    for (; i < i1; i += 256) {
        double a0 = col[i];
        if (VAR) { a0 -= m; a0 *= a0; }
        s0 += d[i] * a0;
    }
    }
    double s = jch_block_sum<256>((s0 + s1) + (s2 + s3), sc);
    if (threadIdx.x == 0) colpart[(size_t)blockIdx.y * (size_t)(p + q) + j] = s;
}

__global__ __launch_bounds__(256) void k_colreduce(const double *__restrict__ colpart, int S, int m, int sqrt_out,
                                                   double *__restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    double s = 0.0;
    // same order of the sum, but 16 independent loads per trip instead of S dependent round trips (12 us -> a third at S = 64)
    for (int k0 = 0; k0 < S; k0 += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = colpart[(size_t)min(k0 + u, S - 1) * m + j];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += k0 + u < S ? v[u] : 0.0;
    }
    out[j] = s;
    (void)sqrt_out;
}

__global__ __launch_bounds__(256) void k_sqrt_inplace(double *__restrict__ v, int m)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < m) v[j] = sqrt(v[j]);
}

int32_t jch_launch_moments(jch_ctx *ctx, const double *Xc, int64_t ldx, const double *Yc, int64_t ldy, const double *d,
                           int64_t n, int p, int q, const double *means, double *out, bool do_sqrt)
{
    const int m = p + q;
    int S = (ctx->cus * 8 + m - 1) / m;
    if (S < 1) S = 1;
    if (S > 64) S = 64;
    int64_t chunk = (n + S - 1) / S;
    chunk = (chunk + 255) / 256 * 256;
    if (chunk < 256) chunk = 256;
    S = (int)((n + chunk - 1) / chunk);
    if (S < 1) S = 1;
    JCH_TRY(jch_reserve(ctx, ctx->colpart, sizeof(double) * (size_t)S * m + 4096 * sizeof(double)));
    double *colpart = (double *)ctx->colpart.ptr;
    // 16-B loads need every column start 16-B aligned (chunk is a multiple of 256 rows)
    const int al16 = (ldx % 2 == 0) && ((uintptr_t)Xc % 16 == 0) && (q == 0 || ((ldy % 2 == 0) && ((uintptr_t)Yc % 16 == 0))) &&
                     ((uintptr_t)d % 16 == 0) && !getenv("JCH_K1_V1");
    if (means)
        hipLaunchKernelGGL(k_moments<true>, dim3(m, S), dim3(256), 0, ctx->stream, Xc, ldx, Yc, ldy, d, n, p, q, chunk,
                           means, colpart, al16);
    else
        hipLaunchKernelGGL(k_moments<false>, dim3(m, S), dim3(256), 0, ctx->stream, Xc, ldx, Yc, ldy, d, n, p, q, chunk,
                           means, colpart, al16);
    hipLaunchKernelGGL(k_colreduce, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, colpart, S, m, 0, out);
    JCH_TRY(jch_allreduce_f64(ctx, out, (size_t)m));
    if (means && do_sqrt) hipLaunchKernelGGL(k_sqrt_inplace, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, out, m);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

// ---------------------------------------------------------------- K2: centre/scale + row-major copy + XtY
// Tile: 64 rows x 64 columns, 256 threads (4 waves).  grid = (row slots, column tiles, y groups of 16).
#define XT_LD 65
#define YT_LD 17
// WRITEBACK (plskern!/plsnipals! semantics) stores the centred X back into the caller's column-major
// array; it is only legal when each X element is read by exactly one block (one y group).  Y is never
// written here (every column tile re-reads the raw Y rows): the launcher exports Yr afterwards.
template <bool WRITEBACK, bool SCAL>
__global__ __launch_bounds__(256) void k_center_xty(double *__restrict__ Xc, int64_t ldx, const double *__restrict__ Yc,
                                                     int64_t ldy, const double *__restrict__ d, int64_t n, int p, int q,
                                                     const double *__restrict__ mom, const double *__restrict__ scl,
                                                     double *__restrict__ Xr, int ldr, double *__restrict__ Yr, int qpad,
                                                     double *__restrict__ Kpart, int kp_rows, int dbg_skip, int ones_col)
{
    __shared__ double xt[64 * XT_LD];
    __shared__ double yt[64 * YT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j0 = blockIdx.y * 64;
    const int yg = blockIdx.z;
    const int64_t nchunks = (n + 63) / 64;
    // per-thread column constants: lane = row, columns j0 + wv + 4k (k < 16)
    double cm[16], cs[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int j = j0 + wv + 4 * k;
        cm[k] = j < p ? mom[j] : 0.0;
        cs[k] = (SCAL && j < p) ? scl[j] : 1.0;  // divisor, used only when scaling (cscale!: (x - u) / v)
    }
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    // software pipeline: the global loads of chunk c+1 are in flight while chunk c is stored / multiplied
    double xr[16];
    auto prefetch = [&](int64_t cc) {
        const int64_t i = cc * 64 + lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = j0 + wv + 4 * k;
            xr[k] = (i < n && j < p) ? __builtin_nontemporal_load(Xc + (size_t)i + (size_t)j * (size_t)ldx) : 0.0;
        }
    };
    int64_t c = blockIdx.x;
    if (c < nchunks) prefetch(c);
    for (; c < nchunks; c += gridDim.x) {
        const int64_t i0 = c * 64;
        // ---- Y tile: 64 rows x 16 cols, element e = tid + 256k -> (row e&63, col e>>6)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = tid + 256 * k, row = e & 63, col = e >> 6;
            const int yc = yg * 16 + col;
            const int64_t i = i0 + row;
            double v = 0.0, dv = 0.0;
            if (i < n && yc < q) {
                v = Yc[(size_t)i + (size_t)yc * (size_t)ldy] - mom[p + yc];
                if (SCAL) v /= scl[p + yc];
                dv = d[i];
            }
            if (blockIdx.y == 0 && i < n) Yr[(size_t)i * qpad + yc] = v;
            // raw mode (fit.hip): pad column `ones_col` of the y tile carries the weights themselves, so column ones_col
            // of X'D[Yc | 1] is the vector of weighted column sums of X — the means come out of the same pass
            yt[row * YT_LD + col] = (yc == ones_col && i < n) ? d[i] : dv * v;
        }
        // ---- X tile: centre/scale the prefetched registers into LDS (and back into the caller's array)
        {
            const int64_t i = i0 + lane;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int j = j0 + wv + 4 * k;
                double v = 0.0;
                if (i < n && j < p) {
                    v = SCAL ? (xr[k] - cm[k]) / cs[k] : xr[k] - cm[k];
                    if (WRITEBACK && yg == 0) Xc[(size_t)i + (size_t)j * (size_t)ldx] = v;
                }
                xt[lane * XT_LD + wv + 4 * k] = v;
            }
        }
        __syncthreads();
        // ---- prefetch the next chunk
        if (c + gridDim.x < nchunks) prefetch(c + gridDim.x);
        // ---- row-major store (y group 0 only): (row wv+4k, col lane)
        if (yg == 0 && !(dbg_skip & 2)) {
#pragma unroll 4
            for (int k = 0; k < 16; ++k) {
                const int row = wv + 4 * k, j = j0 + lane;
                const int64_t i = i0 + row;
                if (i < n && j < ldr) __builtin_nontemporal_store(xt[row * XT_LD + lane], Xr + (size_t)i * ldr + j);
            }
        }
        // ---- XtY on the matrix cores: wave wv owns columns j0+16wv .. +15
        //      A[m = x column][k = row] , B[k = row][n = y column]; lane l: (k = l>>4, m|n = l&15)
        if (!(dbg_skip & 1)) {
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
    // D[m][n]: n = lane&15 (y column), m = (lane>>4) + 4*reg (x column within the wave's 16)
    double *kp = Kpart + ((size_t)blockIdx.x * kp_rows) * qpad;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int j = j0 + 16 * wv + (lane >> 4) + 4 * reg;
        if (j < kp_rows) kp[(size_t)j * qpad + yg * 16 + (lane & 15)] = acc[reg];
    }
}

// ---------------------------------------------------------------- K2p: row-panel version of K2 (q <= 16, 16-B aligned columns)
// Same outputs as k_center_xty (row-major copy, Yr, XtY partials).  What round 2 measured about this pass
// (profiles/r02_k2_notes.md; tools/k2_modes.py compares variants inside one process, on the same buffers):
//   * the tile kernel below is NOT short of TLB reach (99.9 % UTCL1 hits) or starved by latency (1.2 k cycles per read
//     request, like the sweep); its 8.4 GB move at 3.9 TB/s;
//   * reading alone (stores compiled out) runs at 5 TB/s whatever the block mapping; the stores add ~1.05 ms for their 4 GB
//     with 512-B or 1-KiB row pieces, temporal or non-temporal, 1-3 blocks per CU — the write stream of a transposition is
//     what the memory system dislikes, and a float4 copy (6.3 TB/s, MI355X_MICROARCH.md) bounds this pass at 1.33 ms;
//   * fewer, fatter streams are better: ONE 256-thread block per CU beat two (1.90 vs 1.97 ms) and three (2.13); blocks
//     that each own a private row range are faster or slower by 8 % depending on where hipMalloc put the buffers, row tiles
//     interleaved over the blocks are not; a mapping that wrote 512-B strips from 512 scattered ranges fell to 2.9 TB/s.
//   * with one block per CU, 64-column pieces beat 128-column ones (1.68-1.75 vs 1.80-1.84 ms); cutting the 1-KiB pieces at
//     128-B aligned addresses (a carry strip in LDS for the rows that start mid-line) or padding the row pitch to 128 B
//     changed nothing once the kernel was free of register spills.
// So: a block walks row tiles b, b + G, b + 2G, ... (G = blocks = CUs); each tile is TH rows x all columns, worked through
// in TW-column pieces: every row tile leaves the block as TH complete rows of the copy.  The next piece's loads are
// issued right after the barrier that publishes the current one.  The B operand of the XtY product (d .* yc, plus the
// weights in the ones column) is built ONCE per row tile, directly in the MFMA lane layout, from the column-major Y (no Y
// tile in LDS, no re-read per column tile); the eight 64-column tiles of a 512-column group keep their accumulators in
// registers (8 x 4 doubles per lane) — the column loop is a runtime loop, only the MFMA section names its accumulator
// statically (a fully unrolled loop spilled 2 300 registers).  v_mfma_f64_16x16x4 runs at 64 cycles on gfx950 (f64 matrix
// rate == f64 vector rate): 0.21 ms of matrix-pipe time per cfg2 prologue, 3 % of the pass when compiled out.
typedef double v2f64p __attribute__((ext_vector_type(2)));
// TH rows x TW columns per LDS tile (TW = 64 or 128: the row pieces written to the copy are TW * 8 bytes); the tile is
// filled and multiplied in 64-column halves, stored as whole TW-wide rows.
template <int TH, int TW, bool WRITEBACK, bool SCAL>
__global__ __launch_bounds__(256, 2) void k_center_xty_panel(double *__restrict__ Xc, int64_t ldx, const double *__restrict__ Yc,
                                                             int64_t ldy, const double *__restrict__ d, int64_t n, int p, int q,
                                                             const double *__restrict__ mom, const double *__restrict__ scl,
                                                             double *__restrict__ Xr, int ldr, double *__restrict__ Yr,
                                                             double *__restrict__ Kpart, int kp_rows, int ones_col, int dbg_skip)
{
    extern __shared__ __attribute__((aligned(16))) double xp_lds[];
    constexpr int PT = TW + 2;           // LDS row pitch in doubles (even: 16-B aligned rows for b128 reads; 2-way conflicts on the column-wise fill)
    constexpr int NSUB = TW / 64;        // 64-column halves per tile
    double *xt = xp_lds;                 // [TH][PT]
    double *cm_s = xt + TH * PT;         // [512] column shifts of this column group
    double *cs_s = cm_s + 512;           // [512] column divisors (SCAL)
    constexpr bool BLDS = TH > 64;       // tall tiles: the B operand lives in LDS (TH x 16 doubles) instead of TH / 4 registers
    double *yt_s = cs_s + 512;           // [TH][16]  (BLDS only)
    constexpr int NL = TH / 8;           // 16-B loads per thread and 64-column half
    constexpr int HALF = TH / 2;         // row pairs per column
    constexpr int CPI = 128 / TH;        // columns per wave-instruction (1: TH = 128, 2: TH = 64)
    typedef double v4f64p __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cg0 = blockIdx.y * 512;
    const int ntile = min(8, (ldr - cg0 + 63) / 64);   // 64-column tiles of this column group
    // Row tiles of a block.  Interleaved (default): block b takes tiles b, b + G, b + 2G, ... — at any moment the G blocks
    // work on G consecutive row tiles, so for every column they consume one contiguous run of the input (G * TH * 8 B) and
    // together they write one contiguous region of the copy; the run time then does not depend on where the buffers live.
    // Contiguous (dbg_skip & 16): one balanced row range per block — faster or slower by 8 % depending on the placement.
    int64_t rbeg, rend, istep;
    if (dbg_skip & 16) {
        const int64_t nunits = (n + 31) / 32;
        rbeg = 32 * ((nunits * blockIdx.x) / gridDim.x);
        rend = 32 * ((nunits * (blockIdx.x + 1)) / gridDim.x);
        if (rend > n) rend = n;
        istep = TH;
    } else {
        rbeg = (int64_t)blockIdx.x * TH;
        rend = n;
        istep = (int64_t)gridDim.x * TH;
    }
    for (int c = tid; c < 512; c += 256) {
        const int j = cg0 + c;
        cm_s[c] = j < p ? mom[j] : 0.0;
        cs_s[c] = (SCAL && j < p) ? scl[j] : 1.0;
    }
    const int rp = lane % HALF, csub = lane / HALF;
    const int ycol = lane & 15;
    const double ym = ycol < q ? mom[p + ycol] : 0.0;
    const double ysd = (SCAL && ycol < q) ? scl[p + ycol] : 1.0;
    v4f64p acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = v4f64p{0.0, 0.0, 0.0, 0.0};
    v2f64p R[NL];
    auto fetch = [&](int64_t i0, int ct) {   // ct: 64-column tile
        const int64_t i = i0 + 2 * rp;
        const bool rowok = i < rend;
        const int jb = cg0 + 64 * ct + CPI * wv + csub;
        const double *base = Xc + (size_t)i + (size_t)jb * (size_t)ldx;
        const size_t step = (size_t)(4 * CPI) * (size_t)ldx;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const bool ok = rowok && jb + 4 * CPI * k < p;
            R[k] = ok ? __builtin_nontemporal_load(reinterpret_cast<const v2f64p *>(base + k * step)) : v2f64p{0.0, 0.0};
        }
    };
    int64_t i0 = rbeg;
    if (i0 < rend) fetch(i0, 0);
    __syncthreads();   // cm_s / cs_s
    for (; i0 < rend; i0 += istep) {
        // ---- B operand of this row range in the MFMA layout: lane l -> (k = row 4 kk + (l >> 4), n = y column l & 15)
        //      (registers: every wave builds all of it; LDS: each wave builds a quarter, published by the first fill barrier)
        double breg[BLDS ? 1 : TH / 4];
        {
            // all loads of the tile's Y rows and weights go out back to back, UNCONDITIONALLY (clamped row / column, masked
            // afterwards): written as `if (row < rend) { load ... }` per k-step the compiler put a vmcnt(0) into every one of
            // the TH / 4 iterations — 16 dependent memory round trips at the head of every row tile
            constexpr int NB = BLDS ? TH / 16 : TH / 4;
            double yraw[NB], draw[NB];
            const int yc_ = min(ycol, q - 1);
#pragma unroll
            for (int kq = 0; kq < NB; ++kq) {
                const int kk = BLDS ? wv * (TH / 16) + kq : kq;
                const int64_t row = min(i0 + 4 * kk + (lane >> 4), rend - 1);
                draw[kq] = d[row];
                yraw[kq] = Yc[(size_t)row + (size_t)yc_ * (size_t)ldy];
            }
#pragma unroll
            for (int kq = 0; kq < NB; ++kq) {
                const int kk = BLDS ? wv * (TH / 16) + kq : kq;
                const int64_t row = i0 + 4 * kk + (lane >> 4);
                const bool ok = row < rend;
                double yv = 0.0;
                if (ycol < q) {
                    yv = yraw[kq] - ym;
                    if (SCAL) yv /= ysd;
                }
                const double dv = ok ? draw[kq] : 0.0;
                if (!ok) yv = 0.0;
                if (ok && (BLDS || wv == 0) && blockIdx.y == 0) Yr[(size_t)row * 16 + ycol] = yv;
                // raw mode: the pad column `ones_col` carries the weights themselves -> that column of X'D[Yc | 1] is the vector
                // of weighted column sums of X (the means come out of the same pass, fit.hip)
                const double bv = ycol == ones_col ? dv : dv * yv;
                if (BLDS) yt_s[(4 * kk + (lane >> 4)) * 16 + ycol] = bv;
                else breg[kq] = bv;
            }
        }
        // (runtime loop over the tiles: only the MFMA section below names its accumulator statically)
#pragma unroll 1
        for (int ct0 = 0; ct0 < ntile; ct0 += NSUB) {
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                const int ct = ct0 + sub;   // (ct may equal ntile for the last half of an odd tile count: an all-zero half)
                // ---- registers -> LDS half tile [row][64 sub + col], centred / scaled (and back into the caller's array: plskern!)
                {
                    const int64_t i = i0 + 2 * rp;
                    const bool ok0 = i < rend, ok1 = i + 1 < rend;
                    const int cb = CPI * wv + csub;
#pragma unroll
                    for (int k = 0; k < NL; ++k) {
                        const int c = cb + 4 * CPI * k;
                        const int j = cg0 + 64 * ct + c;
                        const bool live = ct < ntile && j < p;
                        const double m = cm_s[(64 * ct + c) & 511];
                        v2f64p v = R[k];
                        if (SCAL) { const double sdv = cs_s[(64 * ct + c) & 511]; v.x = (v.x - m) / sdv; v.y = (v.y - m) / sdv; }
                        else { v.x -= m; v.y -= m; }
                        if (!(ok0 && live)) v.x = 0.0;
                        if (!(ok1 && live)) v.y = 0.0;
                        if (WRITEBACK && ok0 && live) {
                            if (ok1) *reinterpret_cast<v2f64p *>(Xc + (size_t)i + (size_t)j * (size_t)ldx) = v;
                            else Xc[(size_t)i + (size_t)j * (size_t)ldx] = v.x;
                        }
                        xt[(2 * rp) * PT + 64 * sub + c] = v.x;
                        xt[(2 * rp + 1) * PT + 64 * sub + c] = v.y;
                    }
                }
                __syncthreads();
                // ---- the next (non-empty) half tile's loads fly while this one is multiplied (and the tile stored); an empty
                //      trailing half (odd tile count, TW = 128) leaves R alone: it already holds the next row tile's first half
                if (ct + 1 < ntile) fetch(i0, ct + 1);
                else if (ct + 1 == ntile && i0 + istep < rend) fetch(i0 + istep, 0);
                // ---- XtY on the matrix cores: wave wv owns x columns 16 wv .. 16 wv + 15 of the half tile
                if (!(dbg_skip & 1) && ct < ntile) {
                    const double *ap = xt + (lane >> 4) * PT + 64 * sub + 16 * wv + (lane & 15);
                    const double *bp = yt_s + (lane >> 4) * 16 + (lane & 15);
#define JCH_XP_MM(T) case T: _Pragma("unroll") for (int kk = 0; kk < TH / 4; ++kk) \
                             acc[T] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * kk * PT], BLDS ? bp[64 * kk] : breg[BLDS ? 0 : kk], acc[T], 0, 0, 0); break;
                    switch (ct) { JCH_XP_MM(0) JCH_XP_MM(1) JCH_XP_MM(2) JCH_XP_MM(3) JCH_XP_MM(4) JCH_XP_MM(5) JCH_XP_MM(6) JCH_XP_MM(7) default: break; }
#undef JCH_XP_MM
                }
                if (sub + 1 == NSUB) {
                    // ---- row-major store of the whole tile: one wave-instruction = 1 KiB (TW = 128: one row; TW = 64: two rows)
                    if constexpr (TW > 128) {
                        // whole-row tile (round 3 experiment, TH = 32 x TW = 512): the tile's rows are complete rows of the copy, i.e. ONE
                        // contiguous 128 KB region per tile; wave wv stores rows wv, wv + 4, ..., a quarter row per instruction
                        if (!(dbg_skip & 2)) {
                            for (int r = wv; r < TH; r += 4) {
                                const int64_t irow = i0 + r;
#pragma unroll
                                for (int c2 = lane; c2 < TW / 2; c2 += 64) {
                                    const int j = cg0 + 2 * c2;
                                    const v2f64p v = *reinterpret_cast<const v2f64p *>(xt + r * PT + 2 * c2);
                                    if (irow < rend && j < ldr) {
                                        if (dbg_skip & 4) *reinterpret_cast<v2f64p *>(Xr + (size_t)irow * (size_t)ldr + j) = v;
                                        else __builtin_nontemporal_store(v, reinterpret_cast<v2f64p *>(Xr + (size_t)irow * (size_t)ldr + j));
                                    }
                                }
                            }
                        }
                    } else
                    if (!(dbg_skip & 2)) {
                        constexpr int LPR = TW / 2;                  // lanes per row (16 B each)
                        constexpr int RPI = 64 / LPR;                // rows per wave-instruction
                        const int col = 2 * (lane % LPR);
                        const int j = cg0 + 64 * ct0 + col;
                        const int64_t ibase = i0 + wv * (TH / 4) + lane / LPR;
                        double *dst = Xr + (size_t)ibase * (size_t)ldr + j;
                        const double *src = xt + (wv * (TH / 4) + lane / LPR) * PT + col;
#pragma unroll
                        for (int sidx = 0; sidx < TH / (4 * RPI); ++sidx) {
                            const v2f64p v = *reinterpret_cast<const v2f64p *>(src + RPI * sidx * PT);
                            if (ibase + RPI * sidx < rend && j < ldr) {
                                if (dbg_skip & 4) *reinterpret_cast<v2f64p *>(dst + (size_t)(RPI * sidx) * (size_t)ldr) = v;
                                else __builtin_nontemporal_store(v, reinterpret_cast<v2f64p *>(dst + (size_t)(RPI * sidx) * (size_t)ldr));
                            }
                        }
                    }
                    __syncthreads();   // the tile may be overwritten
                }
            }
        }
    }
    // D[m][n]: n = lane & 15 (y column), m = (lane >> 4) + 4 reg (x column within the wave's 16)
    double *kp = Kpart + ((size_t)blockIdx.x * kp_rows) * 16;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
        if (ct < ntile) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = cg0 + 64 * ct + 16 * wv + (lane >> 4) + 4 * reg;
                if (j < kp_rows) kp[(size_t)j * 16 + (lane & 15)] = acc[ct][reg];
            }
        }
}

// Fixed-order sum of the per-block XtY partials [nbx][kp_rows][qpad] -> K [p][qpad]: 4 groups of blocks per entry summed
// with 4 independent chains each, combined in group order (bit-reproducible; nbx may be several hundred).
__global__ __launch_bounds__(256) void k_reduce_kpart_wide(const double *__restrict__ Kpart, int nbx, int kp_rows, int p, int qpad,
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

__global__ __launch_bounds__(256) void k_reduce_kpart(const double *__restrict__ Kpart, int nbx, int kp_rows, int p, int qpad,
                                                      double *__restrict__ K)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= p * qpad) return;
    const size_t stride = (size_t)kp_rows * qpad;
    double s = 0.0;
    for (int b = 0; b < nbx; ++b) s += Kpart[(size_t)b * stride + e];
    K[e] = s;
}

// raw mode: column `col` of K holds sum_i d_i (x_ij - pivot_j); means = pivot + that, and the sweeps need (means - pivot).
// Pivot quality: qual <- max_j |means_j - pivot_j| / spread_j (spread = the sample standard deviation k_pivot_rows saw);
// the rounding error of the raw formulation grows with the square of that ratio, so the fit checks it when it fetches
// its results and falls back to the centred copy when the pivot turned out to be far from the means (fit.hip).
__global__ __launch_bounds__(256) void k_extract_means(double *__restrict__ K, int qpad, int p, int col, const double *pivot,
                                                       double *__restrict__ means, double *__restrict__ mshift,
                                                       const double *__restrict__ spread2, double *__restrict__ qual, int q,
                                                       double *ones_out)
{
    // (`pivot` and `ones_out` ALIAS in raw mode — neither is __restrict__, and the entry is read into a register before
    // the slot is overwritten)
    const int j = blockIdx.x * 256 + threadIdx.x;
    const double pv = j < p + q ? pivot[j] : 0.0;
    // raw mode, fit.hip: `pivot` is the scale vector's storage ([pivot (p) | Y means (q)]); once this kernel has read its
    // entry it turns the slot into the divisor 1 the rest of the fit expects there, and the Y means move next to the X
    // means (ones_out == pivot's storage, or null)
    if (ones_out && j >= p && j < p + q) { means[j] = pv; ones_out[j] = 1.0; }
    if (j < p) {
        const double dm = K[(size_t)j * qpad + col];
        means[j] = pv + dm;
        mshift[j] = dm;
        K[(size_t)j * qpad + col] = 0.0;
        if (ones_out) ones_out[j] = 1.0;
        if (qual) {
            const double a = fabs(dm);
            const double ratio = a > 0.0 ? a / sqrt(spread2[j]) : 0.0;   // (NaN data: comparison false -> 0, NaN propagates through the fit itself)
            // non-negative doubles order like their bit patterns
            atomicMax(reinterpret_cast<unsigned long long *>(qual), (unsigned long long)__double_as_longlong(ratio));
        }
    }
}

// Pivot of the raw mode: the plain mean of a STRIDED sample of (up to) 256 rows of every rank's shard — rows 0, s, 2s, ...
// with s = n / 256 —, the ranks' sample means combined with weights n_r / n_total (hdr[1] = n_total).  Any vector within a
// few standard deviations of the column means will do; it only has to be the SAME on every rank (all-reduced by the
// caller).  spread2 receives the sample variances combined the same way.  One wave per column.
__global__ __launch_bounds__(256) void k_pivot_rows(const double *__restrict__ Xc, int64_t ldx, int64_t n, int p,
                                                    const double *__restrict__ hdr, double *__restrict__ pivot,
                                                    double *__restrict__ spread2)
{
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= p) return;
    const int m = (int)(n < 256 ? n : 256);
    const int64_t stride = n / m;
    double v[4], s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k;
        v[k] = i < m ? Xc[(size_t)i * (size_t)stride + (size_t)j * (size_t)ldx] : 0.0;
        s += v[k];
    }
    const double mean = jch_wave_sum(s) / m;
    double s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double e = lane + 64 * k < m ? v[k] - mean : 0.0;
        s2 += e * e;
    }
    s2 = jch_wave_sum(s2) / m;
    const double f = (double)n / hdr[1];
    if (lane == 0) { pivot[j] = mean * f; spread2[j] = s2 * f; }
}

int32_t jch_launch_pivot(jch_ctx *ctx, const double *Xc, int64_t ldx, int64_t n, int p, const double *hdr, double *pivot, double *spread2)
{
    hipLaunchKernelGGL(k_pivot_rows, dim3((p + 3) / 4), dim3(256), 0, ctx->stream, Xc, ldx, n, p, hdr, pivot, spread2);
    JCH_HIP(ctx, hipGetLastError());
    JCH_TRY(jch_allreduce_f64(ctx, pivot, (size_t)p));
    return jch_allreduce_f64(ctx, spread2, (size_t)p);
}

// K2r (round 4, second half) — the raw-mode prologue WITHOUT the copy: when the caller promises that X has not changed since the
// previous fit on this ctx (desc->reserved & JCH_REUSE_XCOPY: the folds of a cross-validation, the combinations of a parameter
// grid — same X, other weights / Y), the row-major working copy x - pivot of that fit is still in the workspace, and all a new fit
// needs from X before its first sweep is X'D[Yc | 1] (src/plskern.jl:131 + the weighted column sums, i.e. the means, :119).  One
// streaming READ of the copy, rows as the sweep reads them (a row per wave-instruction quartet, lane l owns the column pairs
// 2 l + 128 k), instead of a read of X plus a write of the copy: 4 GB at the sweep's rate instead of 8.2 GB at the transposing
// kernel's.  Per row the q + 1 coefficients d_i yc_ik (and d_i for the ones column) are wave-uniform: four rows x 16 columns of
// [Yc | 1] are ONE 64-lane load, a v_readlane pair turns an entry into a scalar operand, and the products are plain FMAs into
// (q + 1) accumulators per owned column.  Partial matrices per block in k_center_xty_panel's layout: the same reduction follows.
typedef double v2f64r __attribute__((ext_vector_type(2)));
template <int KC, int NQ>   // KC 128-column chunks per row (ldr <= 128 KC), NQ >= q + 1 coefficient columns
__global__ __launch_bounds__(256) void k_xty_rows(const double *__restrict__ Xr, int64_t n, int ldr, const double *__restrict__ Yc, int64_t ldy,
                                                  const double *__restrict__ d, int q, const double *__restrict__ ymeans, int ones_col,
                                                  double *__restrict__ Yr, double *__restrict__ Kpart, int kp_rows)
{
    extern __shared__ __attribute__((aligned(16))) double xr_red[];   // [4][KC * 128]
    constexpr int R = 4, NBUF = 3;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int coff[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) { const int col = 2 * lane + 128 * k; coff[k] = col < ldr ? col : ldr - 2; }
    v2f64r acc[KC][NQ];
#pragma unroll
    for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int c = 0; c < NQ; ++c) acc[k][c] = v2f64r{0.0, 0.0};
    const int64_t ngroups = (n + R - 1) / R, gstride = (int64_t)gridDim.x * 4;
    const int yk = lane & 15, yr = lane >> 4;                 // this lane's entry of the 4 x 16 block of [Yc | 1]
    const double ym = yk < q ? ymeans[yk] : 0.0;
    v2f64r X[NBUF][R][KC];
    double Yv[NBUF], Dv[NBUF];
    auto fetch = [&](int b, int64_t gg) {
        const int64_t r0 = gg * R;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int64_t row = r0 + rr < n ? r0 + rr : n - 1;
            const double *rp = Xr + (size_t)row * (size_t)ldr;
#pragma unroll
            for (int k = 0; k < KC; ++k) X[b][rr][k] = __builtin_nontemporal_load(reinterpret_cast<const v2f64r *>(rp + coff[k]));
        }
        const int64_t myrow = r0 + yr < n ? r0 + yr : n - 1;
        Yv[b] = Yc[(size_t)myrow + (size_t)min(yk, q - 1) * (size_t)ldy];
        Dv[b] = d[myrow];
    };
    int64_t g = (int64_t)blockIdx.x * 4 + wv;
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b)
        if (g + b * gstride < ngroups) fetch(b, g + b * gstride);
    auto process = [&](int b, int64_t gg) {
        const int64_t r0 = gg * R;
        const bool live = r0 + yr < n;
        const double yc = (live && yk < q) ? Yv[b] - ym : 0.0;
        const double dv = live ? Dv[b] : 0.0;
        if (live) Yr[(size_t)(r0 + yr) * 16 + yk] = yc;       // the centred responses, row-major (one 512-byte run per instruction)
        const double bv = yk == ones_col ? dv : dv * yc;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
#pragma unroll
            for (int c = 0; c < NQ; ++c) {
                const double bc = jch_readlane(bv, 16 * rr + c);   // wave-uniform
#pragma unroll
                for (int k = 0; k < KC; ++k) { acc[k][c].x += X[b][rr][k].x * bc; acc[k][c].y += X[b][rr][k].y * bc; }
            }
        }
    };
    while (g < ngroups) {
#pragma unroll
        for (int b = 0; b < NBUF; ++b) {
            if (g < ngroups) {
                const int64_t ahead = g + (NBUF - 1) * gstride;
                if (ahead < ngroups) fetch((b + NBUF - 1) % NBUF, ahead);
                process(b, g);
                g += gstride;
            }
        }
    }
    // the block's four waves in wave order, one coefficient column at a time; columns past the row end are not stored (coff clamps
    // them onto the last pair, which its own lane stores)
    double *kp = Kpart + (size_t)blockIdx.x * kp_rows * 16;
#pragma unroll 1
    for (int c = 0; c < 16; ++c) {
        if (c < NQ) {
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                v2f64r v = acc[k][0];
#pragma unroll
                for (int cc = 1; cc < NQ; ++cc) v = c == cc ? acc[k][cc] : v;
                *reinterpret_cast<v2f64r *>(xr_red + wv * (KC * 128) + 2 * lane + 128 * k) = v;
            }
        }
        __syncthreads();
        for (int j = threadIdx.x; j < kp_rows; j += 256) {
            double t = 0.0;
            if (c < NQ && j < ldr) t = ((xr_red[j] + xr_red[KC * 128 + j]) + xr_red[2 * KC * 128 + j]) + xr_red[3 * KC * 128 + j];
            kp[(size_t)j * 16 + c] = t;
        }
        __syncthreads();
    }
}

// the raw-mode prologue from the workspace's row-major copy (see k_xty_rows); same outputs as jch_launch_center_xty in raw mode.
// False in *done when the shape is outside this kernel (the caller then runs the full prologue).
int32_t jch_launch_xty_rows(jch_ctx *ctx, const double *Xr, int ldr, const double *Yc, int64_t ldy, const double *d, int64_t n, int p, int q,
                            const double *mom, double *Yr, int qpad, double *K, double *means_out, double *mshift_out,
                            const double *spread2, double *qual, double *ones_out, bool *done)
{
    *done = false;
    if (qpad != 16 || q + 1 > 12 || ldr > 512 || ldr < 2 || !means_out) return JCH_OK;
    const int ones_col = q;
    const int kc = (ldr + 127) / 128, kp_rows = 512;
    const int64_t ngroups = (n + 3) / 4;
    const int nbx = (int)std::max<int64_t>(1, std::min<int64_t>((ngroups + 3) / 4, ctx->cus));
    JCH_TRY(jch_reserve(ctx, ctx->kpart, sizeof(double) * (size_t)nbx * kp_rows * 16));
    double *Kpart = (double *)ctx->kpart.ptr;
    const int nq = q + 1 <= 4 ? 4 : (q + 1 <= 8 ? 8 : 12);
#define JCH_XR(KC, NQ) hipLaunchKernelGGL((k_xty_rows<KC, NQ>), dim3(nbx), dim3(256), sizeof(double) * 4 * KC * 128, ctx->stream, Xr, n, ldr, Yc, ldy, d, q, \
                                          mom + p, ones_col, Yr, Kpart, kp_rows)
#define JCH_XR_NQ(KC) do { if (nq == 4) JCH_XR(KC, 4); else if (nq == 8) JCH_XR(KC, 8); else JCH_XR(KC, 12); } while (0)
    if (kc == 1) JCH_XR_NQ(1); else if (kc == 2) JCH_XR_NQ(2); else if (kc == 3) JCH_XR_NQ(3); else JCH_XR_NQ(4);
#undef JCH_XR_NQ
#undef JCH_XR
    hipLaunchKernelGGL(k_reduce_kpart_wide, dim3((p * 16 + 63) / 64), dim3(256), 0, ctx->stream, Kpart, nbx, kp_rows, p, 16, K);
    JCH_TRY(jch_allreduce_f64(ctx, K, (size_t)p * qpad));
    hipLaunchKernelGGL(k_extract_means, dim3((p + q + 255) / 256), dim3(256), 0, ctx->stream, K, qpad, p, ones_col, mom, means_out, mshift_out, spread2, qual, q, ones_out);
    JCH_HIP(ctx, hipGetLastError());
    *done = true;
    return JCH_OK;
}

int32_t jch_launch_center_xty(jch_ctx *ctx, double *Xc, int64_t ldx, double *Yc, int64_t ldy, const double *d, int64_t n,
                              int p, int q, const double *mom, const double *scl, bool writeback, double *Xr, int ldr,
                              double *Yr, int qpad, double *K, bool scal, double *means_out, double *mshift_out,
                              const double *spread2, double *qual, double *ones_out)
{
    const int ones_col = means_out ? q : -1;   // raw mode: needs a free pad column in y group 0 (q <= 15)
    // Row-panel kernel (k_center_xty_panel): q <= 16 and 16-B aligned columns; defaults (64-row x 64-column pieces, one block
    // per CU) are the measured best.  JCH_K2_PANEL=0 keeps the tile kernel below; JCH_K2_TH / JCH_K2_TW / JCH_K2_BPC / JCH_K2_NB
    // select tile height / piece width / blocks per CU / block count, JCH_K2_SKIP bits compile parts out for timing.
    // (read on every call, not cached: tools/k2_modes.py compares the variants inside one process, on the same buffers —
    // the run time of this pass depends on where the buffers happen to live)
    auto env_int = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
    const int dbg_skip = env_int("JCH_K2_SKIP", 0);
    const int panel_sel = env_int("JCH_K2_PANEL", 1), th_sel = env_int("JCH_K2_TH", 64);
    if (panel_sel && qpad == 16 && ldx % 2 == 0 && ((uintptr_t)Xc) % 16 == 0) {
        const int groups = (ldr + 511) / 512;
        const int kp_rows = groups * 512;
        const int64_t nunits = (n + 31) / 32;
        const int bpc_sel = env_int("JCH_K2_BPC", 1);
        int nbx = std::max(1, (ctx->cus * bpc_sel) / groups);
        if (env_int("JCH_K2_NB", 0) > 0) nbx = env_int("JCH_K2_NB", 0);
        if (nbx > nunits) nbx = (int)nunits;
        JCH_TRY(jch_reserve(ctx, ctx->kpart, sizeof(double) * (size_t)nbx * kp_rows * 16));
        double *Kpart = (double *)ctx->kpart.ptr;
        dim3 grid(nbx, groups);
        const bool wholerow = th_sel == 32;                     // TH = 32 x TW = 512: complete rows leave the LDS tile (experiment)
        const int th = wholerow ? 32 : (th_sel == 128 ? 128 : 64);
        const int tw = wholerow ? 512 : (env_int("JCH_K2_TW", 64) == 128 ? 128 : 64);
        const size_t lds = sizeof(double) * ((size_t)th * (tw + 2) + 1024 + (th > 64 ? (size_t)th * 16 : 0));
        static jch_per_device_once attr_once;
        if (!attr_once.done(ctx->device)) {
#define JCH_K2P_ATTR(TH, TW, WB, SC) JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_center_xty_panel<TH, TW, WB, SC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
#define JCH_K2P_ATTR4(TH, TW) JCH_K2P_ATTR(TH, TW, false, false); JCH_K2P_ATTR(TH, TW, false, true); JCH_K2P_ATTR(TH, TW, true, false); JCH_K2P_ATTR(TH, TW, true, true)
            JCH_K2P_ATTR4(128, 64); JCH_K2P_ATTR4(64, 64); JCH_K2P_ATTR4(64, 128); JCH_K2P_ATTR4(128, 128); JCH_K2P_ATTR4(32, 512);
#undef JCH_K2P_ATTR4
#undef JCH_K2P_ATTR
            attr_once.mark(ctx->device);
        }
#define JCH_K2P(TH, TW, WB, SC) hipLaunchKernelGGL((k_center_xty_panel<TH, TW, WB, SC>), grid, dim3(256), lds, ctx->stream, Xc, ldx, Yc, ldy, d, n, p, q, \
                                                   mom, scl, Xr, ldr, Yr, Kpart, kp_rows, ones_col, dbg_skip)
#define JCH_K2P_TH(TH, TW) do { if (writeback && scal) JCH_K2P(TH, TW, true, true); else if (writeback) JCH_K2P(TH, TW, true, false); \
                                else if (scal) JCH_K2P(TH, TW, false, true); else JCH_K2P(TH, TW, false, false); } while (0)
        if (wholerow) JCH_K2P_TH(32, 512); else if (th == 128 && tw == 128) JCH_K2P_TH(128, 128); else if (th == 128) JCH_K2P_TH(128, 64); else if (tw == 128) JCH_K2P_TH(64, 128); else JCH_K2P_TH(64, 64);
#undef JCH_K2P_TH
#undef JCH_K2P
        hipLaunchKernelGGL(k_reduce_kpart_wide, dim3((p * 16 + 63) / 64), dim3(256), 0, ctx->stream, Kpart, nbx, kp_rows, p, 16, K);
        JCH_TRY(jch_allreduce_f64(ctx, K, (size_t)p * qpad));
        if (means_out) hipLaunchKernelGGL(k_extract_means, dim3((p + q + 255) / 256), dim3(256), 0, ctx->stream, K, qpad, p, ones_col, mom, means_out, mshift_out, spread2, qual, q, ones_out);
        JCH_HIP(ctx, hipGetLastError());
        if (writeback)  // X went back inside the kernel; Y from the row-major copy
            JCH_TRY(jch_launch_export_colmajor(ctx, nullptr, ldr, Yr, qpad, n, p, q, Xc, ldx, Yc, ldy));
        return JCH_OK;
    }
    // Tile 64 rows x 64 columns, 8-B loads.  Measured and dropped (cfg2): a 128 x 32 tile with 16-B loads (+1.1 ms: its 256-B row
    // segments are mostly partial 128-B lines), 16-B loads into this tile (no gain), a second prefetch stage, a Y / weight
    // prefetch, a 64-column panel layout of the copy (K2 -0.24 ms, sweeps +0.33 ms): the kernel is bound by its 512-B
    // column-major reads (3.5 TB/s with or without the stores).
    const int tw = 64, th = 64;
    const int ptiles = (ldr + tw - 1) / tw;
    const int kp_rows = ptiles * tw;
    const int ygroups = qpad / 16;
    const int64_t nchunks = (n + th - 1) / th;
    int nbx = (ctx->cus * 3 + ptiles * ygroups - 1) / (ptiles * ygroups);
    if (nbx < 1) nbx = 1;
    if (nbx > nchunks) nbx = (int)(nchunks > 0 ? nchunks : 1);
    const int nslots = nbx;
    JCH_TRY(jch_reserve(ctx, ctx->kpart, sizeof(double) * (size_t)nslots * kp_rows * qpad));
    double *Kpart = (double *)ctx->kpart.ptr;
    dim3 grid(nbx, ptiles, ygroups);
    const bool wb_fused = writeback && ygroups == 1;
#define JCH_K2(WB, SC) hipLaunchKernelGGL((k_center_xty<WB, SC>), grid, dim3(256), 0, ctx->stream, Xc, ldx, Yc, ldy, d, n, p, q, \
                                          mom, scl, Xr, ldr, Yr, qpad, Kpart, kp_rows, dbg_skip, ones_col)
    if (wb_fused && scal) JCH_K2(true, true);
    else if (wb_fused) JCH_K2(true, false);
    else if (scal) JCH_K2(false, true);
    else JCH_K2(false, false);
#undef JCH_K2
    hipLaunchKernelGGL(k_reduce_kpart, dim3((p * qpad + 255) / 256), dim3(256), 0, ctx->stream, Kpart, nslots, kp_rows, p, qpad,
                       K);
    JCH_TRY(jch_allreduce_f64(ctx, K, (size_t)p * qpad));
    if (means_out) hipLaunchKernelGGL(k_extract_means, dim3((p + q + 255) / 256), dim3(256), 0, ctx->stream, K, qpad, p, ones_col, mom, means_out, mshift_out, spread2, qual, q, ones_out);
    JCH_HIP(ctx, hipGetLastError());
    if (writeback)  // Y always, X only when it could not be fused above
        JCH_TRY(jch_launch_export_colmajor(ctx, wb_fused ? nullptr : Xr, ldr, Yr, qpad, n, p, q, Xc, ldx, Yc, ldy));
    return JCH_OK;
}

// ---------------------------------------------------------------- export: row-major working copy -> column-major
// (plsnipals! hands back the deflated X, Y: src/plsnipals.jl:86-87)
__global__ __launch_bounds__(256) void k_export_colmajor(const double *__restrict__ Xr, int ldr, const double *__restrict__ Yr,
                                                          int qpad, int64_t n, int p, int q, double *__restrict__ Xc,
                                                          int64_t ldx, double *__restrict__ Yc, int64_t ldy,
                                                          const double *__restrict__ dsc)
{
    __shared__ double xt[64 * XT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j0 = blockIdx.y * 64;
    const int64_t nchunks = (n + 63) / 64;
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t i0 = c * 64;
        if (Xr) {
#pragma unroll 4
            for (int k = 0; k < 16; ++k) {
                const int row = wv + 4 * k, j = j0 + lane;
                const int64_t i = i0 + row;
                xt[row * XT_LD + lane] = (i < n && j < p) ? Xr[(size_t)i * ldr + j] : 0.0;
            }
            __syncthreads();
#pragma unroll 4
            for (int k = 0; k < 16; ++k) {
                const int col = wv + 4 * k, j = j0 + col;
                const int64_t i = i0 + lane;
                if (i < n && j < p) Xc[(size_t)i + (size_t)j * (size_t)ldx] = xt[lane * XT_LD + col] * (dsc ? sqrt(dsc[i]) : 1.0);
            }
        }
        if (blockIdx.y == 0 && Yc) {
            for (int e = tid; e < 64 * q; e += 256) {
                const int row = e & 63, yc = e >> 6;
                const int64_t i = i0 + row;
                if (i < n) Yc[(size_t)i + (size_t)yc * (size_t)ldy] = Yr[(size_t)i * qpad + yc] * (dsc ? sqrt(dsc[i]) : 1.0);
            }
        }
        __syncthreads();
    }
}

int32_t jch_launch_export_colmajor(jch_ctx *ctx, const double *Xr, int ldr, const double *Yr, int qpad, int64_t n, int p,
                                   int q, double *Xc, int64_t ldx, double *Yc, int64_t ldy, const double *sqrt_rowscale)
{
    const int ptiles = Xr ? (p + 63) / 64 : 1;
    const int64_t nchunks = (n + 63) / 64;
    int nbx = (ctx->cus * 4 + ptiles - 1) / ptiles;
    if (nbx > nchunks) nbx = (int)(nchunks > 0 ? nchunks : 1);
    hipLaunchKernelGGL(k_export_colmajor, dim3(nbx, ptiles), dim3(256), 0, ctx->stream, Xr, ldr, Yr, qpad, n, p, q, Xc, ldx,
                       Yc, ldy, sqrt_rowscale);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}
