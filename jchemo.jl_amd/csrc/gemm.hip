// K7 — out = X * Bs + bias on v_mfma_f64_16x16x4_f64: the device primitive behind `transform`
// (src/plskern.jl:187-195) and `predict` (src/plskern.jl:226-238).  Centring/scaling is folded into Bs and
// bias on the host (p x k work), so X is read once, untouched, with no m x p temporary (the reference
// allocates one in `cscale`, plskern.jl:191).  X and out are column-major (Julia).
//   M = 16 rows of X per wave (4 waves = 64 rows per block), K = 4 columns of X per MFMA, N = 16 output columns
//   per accumulator, up to 8 accumulators (128 output columns) per block; wider outputs use grid.y.
// Bound: HBM on X (m*p*8 bytes) for k <= 128.
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "jch_internal.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));
#define KCH 32  // X columns per LDS-staged chunk of Bs

__global__ __launch_bounds__(256) void k_affine_gemm(const double *__restrict__ Xc, int64_t m, int p, int64_t ldx,
                                                      const double *__restrict__ Bs, int kpad,
                                                      const double *__restrict__ bias, int k, double *__restrict__ out,
                                                      int64_t ldo)
{
    __shared__ double bl[KCH * 128];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n0 = blockIdx.y * 128;                       // first output column of this block
    const int ncols = min(128, kpad - n0);                 // multiple of 16
    const int ntiles = ncols / 16;
    const int64_t i0 = (int64_t)blockIdx.x * 64 + 16 * wv;
    const int64_t irow = i0 + (lane & 15);
    const bool rlive = irow < m;
    v4f64 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = v4f64{0.0, 0.0, 0.0, 0.0};
    for (int j0 = 0; j0 < p; j0 += KCH) {
        __syncthreads();
        for (int e = tid; e < KCH * ncols; e += 256) {
            const int jj = e / ncols, cc = e % ncols;
            bl[jj * 128 + cc] = (j0 + jj < p) ? Bs[(size_t)(j0 + jj) * kpad + n0 + cc] : 0.0;
        }
        double a[KCH / 4];
#pragma unroll
        for (int kk = 0; kk < KCH / 4; ++kk) {
            const int j = j0 + 4 * kk + (lane >> 4);
            a[kk] = (rlive && j < p) ? Xc[(size_t)irow + (size_t)j * (size_t)ldx] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KCH / 4; ++kk) {
            const double *brow = bl + (4 * kk + (lane >> 4)) * 128 + (lane & 15);
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (t < ntiles) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], brow[16 * t], acc[t], 0, 0, 0);
        }
    }
    // D[row = (lane>>4) + 4 reg][col = lane&15]
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (t >= ntiles) break;
        const int col = n0 + 16 * t + (lane & 15);
        if (col >= k) continue;
        const double bv = bias[col];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t i = i0 + (lane >> 4) + 4 * reg;
            if (i < m) out[(size_t)i + (size_t)col * (size_t)ldo] = acc[t][reg] + bv;
        }
    }
}

// Narrow-output variant (kpad <= 32: `transform` with nlv <= 32, `predict` for one nlv): the hot accessor shapes.
//   * 16-B loads: lane l holds rows (2m, 2m+1), m = l & 15, of column j + (l >> 4): one wave-instruction = 4 x 256 B.
//     The even rows feed one MFMA row-tile, the odd rows a second one (same B operand): 32 rows per wave, 128 per block.
//   * B is staged in chunks of 128 X-columns (32 KB), so 128 MFMAs per wave sit between two barriers, and the X loads
//     are software-pipelined 8 k-steps ahead inside a chunk.
typedef double v2f64 __attribute__((ext_vector_type(2)));
#define G32_CH 128
__global__ __launch_bounds__(256) void k_affine_gemm32(const double *__restrict__ Xc, int64_t m, int p, int64_t ldx,
                                                        const double *__restrict__ Bs, int kpad, const double *__restrict__ bias,
                                                        int k, double *__restrict__ out, int64_t ldo)
{
    __shared__ double bl[G32_CH * 33];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // row tiles of 128 rows, interleaved over a persistent grid (JCH_GEMM_BPC blocks per CU; 0 = one block per tile)
    for (int64_t tile = blockIdx.x; tile * 128 < m; tile += gridDim.x) {
    const int64_t i0 = tile * 128 + 32 * wv;
    const int64_t irow = i0 + 2 * (lane & 15);        // this lane's row pair
    const bool two = irow + 1 < m, one = irow < m;
    const bool vec = two && (ldx % 2 == 0) && ((((uintptr_t)Xc) & 15) == 0) && (irow % 2 == 0);
    v4f64 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = v4f64{0.0, 0.0, 0.0, 0.0};
    const int ntiles = kpad / 16;   // 1 or 2
    auto loadx = [&](int j) -> v2f64 {
        if (j >= p || !one) return v2f64{0.0, 0.0};
        const double *src = Xc + (size_t)irow + (size_t)j * (size_t)ldx;
        if (vec) return __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(src));
        return v2f64{src[0], two ? src[1] : 0.0};
    };
    for (int j0 = 0; j0 < p; j0 += G32_CH) {
        __syncthreads();
        for (int e = tid; e < G32_CH * 32; e += 256) {
            const int jj = e >> 5, cc = e & 31;
            bl[jj * 33 + cc] = (j0 + jj < p && cc < kpad) ? Bs[(size_t)(j0 + jj) * kpad + cc] : 0.0;
        }
        v2f64 xa[8], xb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xa[u] = loadx(j0 + 4 * u + (lane >> 4));
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G32_CH / 32; ++g) {       // groups of 8 k-steps (32 columns)
            if (g + 1 < G32_CH / 32) {
#pragma unroll
                for (int u = 0; u < 8; ++u) xb[u] = loadx(j0 + 32 * (g + 1) + 4 * u + (lane >> 4));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kc = 32 * g + 4 * u + (lane >> 4);
                const double b0 = bl[kc * 33 + (lane & 15)];
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u].x, b0, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u].y, b0, acc[1][0], 0, 0, 0);
                if (ntiles > 1) {
                    const double b1 = bl[kc * 33 + 16 + (lane & 15)];
                    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u].x, b1, acc[0][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u].y, b1, acc[1][1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) xa[u] = xb[u];
        }
    }
    // D[mrow = (lane>>4) + 4 reg][col = lane & 15]; row-tile 0 = even rows, row-tile 1 = odd rows of the wave's 32
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t >= ntiles) break;
        const int col = 16 * t + (lane & 15);
        if (col >= k) continue;
        const double bv = bias[col];
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t i = i0 + 2 * ((lane >> 4) + 4 * reg) + par;
                if (i < m) out[(size_t)i + (size_t)col * (size_t)ldo] = acc[par][t][reg] + bv;
            }
    }
    }
}

// Short inputs (round 3: the query block of a kNN-LWPLSR predict, m = 1000 at cfg5): with 128 rows per workgroup the tiled kernel
// above runs 8 workgroups on a 256-CU chip and every wave issues all p / 4 k-steps of its 32 rows — 56 us at cfg5.  Here a
// workgroup takes 32 rows and its four waves split the COLUMNS of X (split-K inside the workgroup: p / 16 k-steps per wave), B
// comes straight from L2 (no staging, no barrier in the product loop), and the four partial tiles are summed through LDS in a
// fixed order.  Same products, a different association of the sum over k than the tiled kernel.
__global__ __launch_bounds__(256) void k_affine_gemm32s(const double *__restrict__ Xc, int64_t m, int p, int64_t ldx,
                                                         const double *__restrict__ Bs, int kpad, const double *__restrict__ bias,
                                                         int k, double *__restrict__ out, int64_t ldo)
{
    __shared__ double part[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * 32;
    const int64_t irow = i0 + 2 * (lane & 15);        // this lane's row pair
    const bool two = irow + 1 < m, one = irow < m;
    const bool vec = two && (ldx % 2 == 0) && ((((uintptr_t)Xc) & 15) == 0);
    const int ntiles = kpad / 16;   // 1 or 2
    const int ksteps = (p + 3) / 4, per = (ksteps + 3) / 4;
    const int s0 = wv * per, s1 = min(ksteps, s0 + per);
    v4f64 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = v4f64{0.0, 0.0, 0.0, 0.0};
    auto loadx = [&](int j) -> v2f64 {
        if (j >= p || !one) return v2f64{0.0, 0.0};
        const double *src = Xc + (size_t)irow + (size_t)j * (size_t)ldx;
        if (vec) return *reinterpret_cast<const v2f64 *>(src);
        return v2f64{src[0], two ? src[1] : 0.0};
    };
    auto loadb = [&](int j, int t) -> double { return (j < p && 16 * t < kpad) ? Bs[(size_t)j * kpad + 16 * t + (lane & 15)] : 0.0; };
    constexpr int U = 8;                              // k-steps of loads in flight
    v2f64 xa[U]; double ba[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int j = 4 * (s0 + u) + (lane >> 4); const bool in = s0 + u < s1; xa[u] = in ? loadx(j) : v2f64{0.0, 0.0}; ba[u][0] = in ? loadb(j, 0) : 0.0; ba[u][1] = (in && ntiles > 1) ? loadb(j, 1) : 0.0; }
    for (int sb = s0; sb < s1; sb += U) {
        v2f64 xb[U]; double bb[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int st = sb + U + u, j = 4 * st + (lane >> 4); const bool in = st < s1; xb[u] = in ? loadx(j) : v2f64{0.0, 0.0}; bb[u][0] = in ? loadb(j, 0) : 0.0; bb[u][1] = (in && ntiles > 1) ? loadb(j, 1) : 0.0; }
#pragma unroll
        for (int u = 0; u < U; ++u) {                 // (steps past s1 multiply zeros)
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u].x, ba[u][0], acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u].y, ba[u][0], acc[1][0], 0, 0, 0);
            if (ntiles > 1) {
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u].x, ba[u][1], acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u].y, ba[u][1], acc[1][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { xa[u] = xb[u]; ba[u][0] = bb[u][0]; ba[u][1] = bb[u][1]; }
    }
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) part[wv][(par * 2 + t) * 4 + reg][lane] = acc[par][t][reg];
    __syncthreads();
    // D[mrow = (lane >> 4) + 4 reg][col = lane & 15]; row-tile 0 = even rows, row-tile 1 = odd rows of the 32; thread (wave w, lane)
    // finishes entries e = 4 w .. 4 w + 3 of its lane: the four waves' partials in wave order
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
        const int e = 4 * wv + e4, par = e >> 3, t = (e >> 2) & 1, reg = e & 3;
        const int col = 16 * t + (lane & 15);
        const int64_t i = i0 + 2 * ((lane >> 4) + 4 * reg) + par;
        if (t < ntiles && col < k && i < m) out[(size_t)i + (size_t)col * (size_t)ldo] = ((part[0][e][lane] + part[1][e][lane]) + (part[2][e][lane] + part[3][e][lane])) + bias[col];
    }
}

// Persistent variant of the narrow-output kernel for long inputs (round 2): the WHOLE coefficient matrix is staged in LDS
// once per workgroup (p x 33 doubles = 132 KB at cfg2), after which the four waves stream their 32-row tiles without a
// single barrier, X loads PD k-steps ahead of the MFMAs in a rotating register window.  With f64 MFMA at the vector rate
// (64 cycles per 16x16x4) the product itself is 0.41 ms at cfg2 / nlv = 25 against 0.50 ms of HBM time for X: the kernel
// is co-bound, and what matters is that neither pipe waits for the other.
// PAIRED (round 4): the tile TRANSPOSED (operands swapped: the same registers) and rows 2 cl, 2 cl + 1 of a column stored as one
// 16-byte piece — see k_affine_gemm_wideout.
template <int NT, int NW, int RT, bool PAIRED>   // NW waves per workgroup share the one LDS copy of the coefficients; RT row tiles of 32 per wave-tile
__global__ __launch_bounds__(64 * NW) void k_affine_gemm32p(const double *__restrict__ Xc, int64_t m, int p, int64_t ldx,
                                                         const double *__restrict__ Bs, int kpad, const double *__restrict__ bias,
                                                         int k, double *__restrict__ out, int64_t ldo)
{
    extern __shared__ __attribute__((aligned(16))) double blp[];   // [nksp * 4][PB], zero rows beyond p
    // RT (round 4): a wave's tile is 32 RT rows — per column RT back-to-back 256-B pieces = one 256 RT-byte run of the column-major
    // X instead of 256 B (the DRAM access granularity of this kernel); the prefetch window holds the same bytes (PD k-steps x RT)
    constexpr int PB = 16 * NT + 1, PD = 16 / RT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nksp = ((p + 3) / 4 + 15) / 16 * 16;                  // k-steps, padded to the largest prefetch window
    for (int e = tid; e < nksp * 4 * 16 * NT; e += 64 * NW) {
        const int jj = e / (16 * NT), cc = e % (16 * NT);
        blp[jj * PB + cc] = jj < p ? Bs[(size_t)jj * kpad + cc] : 0.0;
    }
    __syncthreads();
    const int kq = lane >> 4, cl = lane & 15;
    constexpr int TR = 32 * RT;
    const int64_t ntile = (m + TR - 1) / TR, tstep = (int64_t)gridDim.x * NW;
    int64_t tile = (int64_t)blockIdx.x * NW + wv;
    if (tile >= ntile) return;
    // EVERY load is unconditional (m is even: row pairs past the end re-read the last pair, columns past p re-read column
    // p - 1 against zero coefficients): with a branch around a load the compiler loses track of the outstanding-load
    // count and waits for all of them (vmcnt(0)) before every MFMA group — measured 2.3x slower than the tiled kernel.
    auto rowoff = [&](int64_t t, int rt) { int64_t ir = t * TR + 32 * rt + 2 * cl; if (ir > m - 2) ir = m - 2; return ir; };
    auto ld = [&](int64_t ir, int ks) {
        const int j = min(4 * ks + kq, p - 1);
        return __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(Xc + ir + (size_t)j * (size_t)ldx));
    };
    int64_t cur[RT], nxt[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { cur[rt] = rowoff(tile, rt); nxt[rt] = rowoff(min(tile + tstep, ntile - 1), rt); }
    v2f64 x[PD][RT];
#pragma unroll
    for (int u = 0; u < PD; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) x[u][rt] = ld(cur[rt], u);
    for (; tile < ntile; tile += tstep) {
        v4f64 acc[RT][2][NT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[rt][a][b] = v4f64{0.0, 0.0, 0.0, 0.0};
        for (int ks0 = 0; ks0 < nksp; ks0 += PD) {
            const bool wrap = ks0 + PD >= nksp;           // the window runs on into the wave's next tile
            const int kb = wrap ? ks0 + PD - nksp : ks0 + PD;
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const double *bp = blp + (4 * (ks0 + u) + kq) * PB + cl;
                double b[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) b[t] = bp[16 * t];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if (PAIRED) {
                            acc[rt][0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[t], x[u][rt].x, acc[rt][0][t], 0, 0, 0);
                            acc[rt][1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[t], x[u][rt].y, acc[rt][1][t], 0, 0, 0);
                        } else {
                            acc[rt][0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][rt].x, b[t], acc[rt][0][t], 0, 0, 0);
                            acc[rt][1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][rt].y, b[t], acc[rt][1][t], 0, 0, 0);
                        }
                    }
                    x[u][rt] = ld(wrap ? nxt[rt] : cur[rt], kb + u);
                }
            }
        }
        const int64_t i0 = tile * TR;
        if (PAIRED) {
            // acc[rt][par][t][reg] of lane (kq, cl) = out[row i0 + 32 rt + 2 cl + par][column 16 t + kq + 4 reg]  (f64 16x16x4: D[kq + 4 reg][cl])
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int col = 16 * t + kq + 4 * reg;
                    if (col >= k) continue;
                    const double bv = bias[col];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const int64_t i = i0 + 32 * rt + 2 * cl;
                        if (i + 1 < m) *reinterpret_cast<v2f64 *>(out + (size_t)i + (size_t)col * (size_t)ldo) = v2f64{acc[rt][0][t][reg] + bv, acc[rt][1][t][reg] + bv};
                    }
                }
        } else
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = 16 * t + cl;
            if (col >= k) continue;
            const double bv = bias[col];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int par = 0; par < 2; ++par)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int64_t i = i0 + 32 * rt + 2 * (kq + 4 * reg) + par;
                        if (i < m) out[(size_t)i + (size_t)col * (size_t)ldo] = acc[rt][par][t][reg] + bv;
                    }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { cur[rt] = nxt[rt]; nxt[rt] = rowoff(min(tile + 2 * tstep, ntile - 1), rt); }
    }
}

// SHORT rows, WIDE output (round 4): the second stage of `predict` over an nlv range (m x nlv scores -> m x q (nlv + 1) predictions,
// 25 -> 260 columns at cfg2) and of `xfit` (scores -> m x p).  The general kernel re-read its input once per 128 output columns and
// wrote 64-row pieces; here a wave keeps its 64 input rows (p <= 64 columns: <= 16 k-steps) in registers, walks ALL output column
// tiles against the coefficient matrix in LDS and writes 512-B runs per output column.  The pass is bound by its OUTPUT (2.08 GB at
// cfg2 against 0.2 GB read).
// PAIRED (round 4): the tile is computed TRANSPOSED — the coefficients as the A operand, the rows as the B operand: the same registers,
// the operands swapped — so that a lane holds rows 2 cl and 2 cl + 1 of output column kq + 4 reg and stores them as ONE 16-byte
// piece: 16 lanes write a contiguous 256-byte run of a column.  Untransposed, a store instruction scattered 64 8-byte pieces over
// 16 columns x 4 rows 16 bytes apart, and the pieces of a line met in L2 from eight instructions (2.08 GB of output at 2.4 TB/s).
// Needs out 16-byte aligned and ldo even (the launcher checks); the values are the same sums in the same order.
template <int KS, int RT, bool PAIRED>   // KS k-steps of 4 input columns held in registers (p <= 4 KS); RT row tiles of 32 per wave-tile
__global__ __launch_bounds__(256) void k_affine_gemm_wideout(const double *__restrict__ Xc, int64_t m, int p, int64_t ldx,
                                                           const double *__restrict__ Bs, int kpad, const double *__restrict__ bias,
                                                           int k, double *__restrict__ out, int64_t ldo, int nt)
{
    extern __shared__ __attribute__((aligned(16))) double blw[];   // [4 KS][kpad + 1], zero rows beyond p
    const int PB = kpad + 1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < 4 * KS * kpad; e += 256) {
        const int jj = e / kpad, cc = e - jj * kpad;
        blw[jj * PB + cc] = jj < p ? Bs[(size_t)jj * kpad + cc] : 0.0;
    }
    __syncthreads();
    const int kq = lane >> 4, cl = lane & 15;
    constexpr int TR = 32 * RT;
    const int64_t ntile = (m + TR - 1) / TR, tstep = (int64_t)gridDim.x * 4;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < ntile; tile += tstep) {
        v2f64 x[KS][RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            int64_t ir = tile * TR + 32 * rt + 2 * cl;
            if (ir > m - 2) ir = m - 2;                         // (m even: re-read the last pair, the stores are guarded)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int j = min(4 * ks + kq, p - 1);          // (columns past p meet zero coefficients)
                x[ks][rt] = __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(Xc + ir + (size_t)j * (size_t)ldx));
            }
        }
        const int64_t i0 = tile * TR;
        for (int ct = 0; ct < kpad / 16; ++ct) {
            v4f64 acc[RT][2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) { acc[rt][0] = v4f64{0.0, 0.0, 0.0, 0.0}; acc[rt][1] = v4f64{0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const double b = blw[(4 * ks + kq) * PB + 16 * ct + cl];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if (PAIRED) {
                        acc[rt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, x[ks][rt].x, acc[rt][0], 0, 0, 0);
                        acc[rt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, x[ks][rt].y, acc[rt][1], 0, 0, 0);
                    } else {
                        acc[rt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[ks][rt].x, b, acc[rt][0], 0, 0, 0);
                        acc[rt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[ks][rt].y, b, acc[rt][1], 0, 0, 0);
                    }
                }
            }
            if (PAIRED) {
                // acc[rt][par][reg] of lane (kq, cl) = out[row i0 + 32 rt + 2 cl + par][column 16 ct + kq + 4 reg]  (f64 16x16x4: D[kq + 4 reg][cl])
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int col = 16 * ct + kq + 4 * reg;
                    if (col < k) {
                        const double bv = bias[col];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            const int64_t i = i0 + 32 * rt + 2 * cl;
                            if (i + 1 < m) {
                                const v2f64 val = {acc[rt][0][reg] + bv, acc[rt][1][reg] + bv};
                                v2f64 *dst = reinterpret_cast<v2f64 *>(out + (size_t)i + (size_t)col * (size_t)ldo);
                                if (nt) __builtin_nontemporal_store(val, dst); else *dst = val;
                            }
                        }
                    }
                }
                continue;
            }
            const int col = 16 * ct + cl;
            if (col < k) {
                const double bv = bias[col];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int par = 0; par < 2; ++par)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int64_t i = i0 + 32 * rt + 2 * (kq + 4 * reg) + par;
                            if (i < m) out[(size_t)i + (size_t)col * (size_t)ldo] = acc[rt][par][reg] + bv;   // (plain stores: L2 merges the 8-B pieces of a line; non-temporal ones were 2x slower)
                        }
            }
        }
    }
}

int32_t jch_launch_affine_gemm(jch_ctx *ctx, const double *Xc, int64_t m, int p, int64_t ldx, const double *Bs, int k, int kpad,
                               const double *bias, double *out, int64_t ldo)
{
    if (kpad <= 32 && !getenv("JCH_GEMM_GENERIC")) {
        // long inputs: persistent kernel with the whole coefficient matrix in LDS (JCH_GEMM_PERSIST=0: the tiled kernel below)
        const size_t ldsp = sizeof(double) * (size_t)(((p + 3) / 4 + 15) / 16 * 64) * (kpad + 1);
        const char *ep = getenv("JCH_GEMM_PERSIST");
        if (!(ep && atoi(ep) == 0) && ldsp <= 150 * 1024 && m >= 64 * 1024 && m % 2 == 0 && ldx % 2 == 0 && (((uintptr_t)Xc) & 15) == 0) {
            // two waves per SIMD, so that one wave's MFMAs cover the other's wait for its loads: two workgroups of 4 waves per CU
            // while two copies of the coefficients fit in LDS (kpad = 16 at cfg2), else ONE workgroup of 8 waves sharing one copy
            // (measured at cfg2, kernel time: nlv = 25 -> 32 columns: 4 waves 0.96 ms, 8 waves 0.87, 12 waves 0.87, tiled kernel 1.03;
            // 10 -> 16 columns: 2 x 4 waves 0.71 ms, 8 waves 0.73, 12 waves 0.75, tiled kernel 0.73.  JCH_GEMM_NW = 4 / 8 overrides.)
            const char *enw = getenv("JCH_GEMM_NW"), *ert = getenv("JCH_GEMM_RT");
            const int nw = enw ? (atoi(enw) == 4 ? 4 : 8) : (kpad == 16 ? 4 : 8);
            // row tiles per wave-tile = 256 rt bytes per column piece.  Measured at cfg2 (round 4, tools/bench_accessors.py, whole call):
            // transform (32 columns) rt = 1 / 2 / 4: 0.905 / 0.877 / 0.846 ms; predict at one nlv (16 columns): 0.839 / 0.769 / 0.768 ms
            const int rt = ert ? (atoi(ert) == 1 ? 1 : (atoi(ert) == 2 ? 2 : 4)) : 4;
            const char *ep32 = getenv("JCH_GEMM_PAIRED");          // (=0: the untransposed tile with 8-byte stores — A/B runs)
            const bool paired32 = !(ep32 && atoi(ep32) == 0) && ldo % 2 == 0 && (((uintptr_t)out) & 15) == 0;
            const int bpc = std::max(1, std::min((int)((158 * 1024) / ldsp), 8 / nw));
            const unsigned nb = (unsigned)std::min<int64_t>((m + 32 * rt * nw - 1) / (32 * rt * nw), (int64_t)ctx->cus * bpc);
#define JCH_G32P(NT, NW, RT) do { \
                static jch_per_device_once once_; \
                if (!once_.done(ctx->device)) { JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_affine_gemm32p<NT, NW, RT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                    JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_affine_gemm32p<NT, NW, RT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once_.mark(ctx->device); } \
                if (paired32) hipLaunchKernelGGL((k_affine_gemm32p<NT, NW, RT, true>), dim3(nb), dim3(64 * NW), ldsp, ctx->stream, Xc, m, p, ldx, Bs, kpad, bias, k, out, ldo); \
                else hipLaunchKernelGGL((k_affine_gemm32p<NT, NW, RT, false>), dim3(nb), dim3(64 * NW), ldsp, ctx->stream, Xc, m, p, ldx, Bs, kpad, bias, k, out, ldo); } while (0)
#define JCH_G32P_RT(NT, NW) do { if (rt == 1) JCH_G32P(NT, NW, 1); else if (rt == 2) JCH_G32P(NT, NW, 2); else JCH_G32P(NT, NW, 4); } while (0)
            if (kpad == 16) { if (nw == 4) JCH_G32P_RT(1, 4); else JCH_G32P_RT(1, 8); }
            else { if (nw == 4) JCH_G32P_RT(2, 4); else JCH_G32P_RT(2, 8); }
#undef JCH_G32P_RT
#undef JCH_G32P
            JCH_HIP(ctx, hipGetLastError());
            return JCH_OK;
        }
        // short inputs: 32 rows per workgroup, the columns of X split over its waves (JCH_GEMM_SMALL=0: the tiled kernel)
        const char *es = getenv("JCH_GEMM_SMALL");
        if (!(es && atoi(es) == 0) && m <= 32 * (int64_t)ctx->cus && p >= 64) {
            hipLaunchKernelGGL(k_affine_gemm32s, dim3((unsigned)((m + 31) / 32)), dim3(256), 0, ctx->stream, Xc, m, p, ldx, Bs, kpad, bias, k, out, ldo);
            JCH_HIP(ctx, hipGetLastError());
            return JCH_OK;
        }
        const char *eb = getenv("JCH_GEMM_BPC");
        const int bpc = eb ? atoi(eb) : 0;
        const int64_t ntile = (m + 127) / 128;
        const unsigned nb = (unsigned)(bpc > 0 ? std::min<int64_t>(ntile, (int64_t)ctx->cus * bpc) : ntile);
        hipLaunchKernelGGL(k_affine_gemm32, dim3(nb), dim3(256), 0, ctx->stream, Xc, m, p, ldx, Bs, kpad, bias,
                           k, out, ldo);
        JCH_HIP(ctx, hipGetLastError());
        return JCH_OK;
    }
    {   // short rows, wide output (JCH_GEMM_WIDEOUT=0: the general kernel; JCH_GEMM_WIDEOUT_RT=2: 64-row wave tiles).  predict over
        // nlv = 0..25 at cfg2 (1e6 x 25 scores -> 260 columns), whole call: general kernel 1.93 ms, this kernel with plain stores
        // 1.77 (64-row tiles); with NON-TEMPORAL stores 4.14 — the 8-B pieces of an output line come from eight store instructions
        // and must meet in L2.
        const char *ew = getenv("JCH_GEMM_WIDEOUT"), *er = getenv("JCH_GEMM_WIDEOUT_RT");
        const int ks = (p + 3) / 4;
        const int rtw = er ? (atoi(er) == 1 ? 1 : atoi(er) == 4 ? 4 : 2) : 2;   // (round 4, with the paired stores: 64-row wave tiles 1.50 ms, 128-row 1.65)
        const size_t ldsw = sizeof(double) * (size_t)(4 * (ks <= 4 ? 4 : ks <= 8 ? 8 : 16)) * (kpad + 1);
        if (!(ew && atoi(ew) == 0) && p <= 64 && kpad > 32 && ldsw <= 150 * 1024 && m >= 4096 && m % 2 == 0 && ldx % 2 == 0 && (((uintptr_t)Xc) & 15) == 0) {
            const unsigned nb = (unsigned)std::min<int64_t>((m + 32 * rtw * 4 - 1) / (32 * rtw * 4), (int64_t)ctx->cus * 2);
            const char *epair = getenv("JCH_GEMM_WIDEOUT_PAIRED");     // (=0: the untransposed tile with 8-byte stores — A/B runs)
            const bool paired = !(epair && atoi(epair) == 0) && ldo % 2 == 0 && (((uintptr_t)out) & 15) == 0;
            const char *ent = getenv("JCH_GEMM_WIDEOUT_NT");
            const int nt_ = ent ? atoi(ent) : 0;
#define JCH_GW(KS, RT) do { \
                static jch_per_device_once once_; \
                if (!once_.done(ctx->device)) { JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_affine_gemm_wideout<KS, RT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                    JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_affine_gemm_wideout<KS, RT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once_.mark(ctx->device); } \
                if (paired) hipLaunchKernelGGL((k_affine_gemm_wideout<KS, RT, true>), dim3(nb), dim3(256), ldsw, ctx->stream, Xc, m, p, ldx, Bs, kpad, bias, k, out, ldo, nt_); \
                else hipLaunchKernelGGL((k_affine_gemm_wideout<KS, RT, false>), dim3(nb), dim3(256), ldsw, ctx->stream, Xc, m, p, ldx, Bs, kpad, bias, k, out, ldo, 0); } while (0)
            if (ks <= 4) { if (rtw == 1) JCH_GW(4, 1); else if (rtw == 2) JCH_GW(4, 2); else JCH_GW(4, 4); }
            else if (ks <= 8) { if (rtw == 1) JCH_GW(8, 1); else if (rtw == 2) JCH_GW(8, 2); else JCH_GW(8, 4); }
            else { if (rtw == 1) JCH_GW(16, 1); else JCH_GW(16, 2); }
#undef JCH_GW
            JCH_HIP(ctx, hipGetLastError());
            return JCH_OK;
        }
    }
    dim3 grid((unsigned)((m + 63) / 64), (unsigned)((kpad + 127) / 128));
    hipLaunchKernelGGL(k_affine_gemm, grid, dim3(256), 0, ctx->stream, Xc, m, p, ldx, Bs, kpad, bias, k, out, ldo);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

extern "C" int32_t jch_affine_gemm(jch_ctx *ctx, int32_t loc, const double *X, int64_t m, int64_t p, int64_t ldx,
                                   const double *shift, const double *scale, const double *B, int64_t k, const double *bias,
                                   double *out, int64_t ldo)
{
    if (!ctx) return JCH_EINVAL;
    if (!X || !B || !out || m < 0 || p < 1 || k < 1 || ldx < m || ldo < m)
        return jch_fail(ctx, JCH_EINVAL, "jch_affine_gemm: bad arguments");
    if (loc != JCH_LOC_HOST && loc != JCH_LOC_DEVICE) return jch_fail(ctx, JCH_EINVAL, "jch_affine_gemm: bad loc");
    if (m == 0) return JCH_OK;
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    const int kpad = (int)((k + 15) / 16 * 16);
    // fold centring/scaling: Bs = diag(1/scale) B ; bias' = bias - shift' Bs      (host, p x k)
    std::vector<double> hb((size_t)p * kpad + kpad, 0.0);
    double *Bs = hb.data(), *b2 = hb.data() + (size_t)p * kpad;
    for (int64_t c = 0; c < k; ++c) {
        double acc = bias ? bias[c] : 0.0;
        for (int64_t j = 0; j < p; ++j) {
            const double v = B[j + c * p] / (scale ? scale[j] : 1.0);
            Bs[j * kpad + c] = v;
            if (shift) acc -= shift[j] * v;
        }
        b2[c] = acc;
    }
    JCH_TRY(jch_reserve(ctx, ctx->gemm_b, sizeof(double) * hb.size()));
    double *dB = (double *)ctx->gemm_b.ptr;
    JCH_HIP(ctx, hipMemcpyAsync(dB, hb.data(), sizeof(double) * hb.size(), hipMemcpyHostToDevice, ctx->stream));
    const double *dX = X;
    double *dO = out;
    int64_t ldxd = ldx, ldod = ldo;
    if (loc == JCH_LOC_HOST) {
        JCH_TRY(jch_reserve(ctx, ctx->xq, sizeof(double) * (size_t)m * p));
        JCH_TRY(jch_reserve(ctx, ctx->gemm_out, sizeof(double) * (size_t)m * k));
        if (ldx == m) JCH_HIP(ctx, hipMemcpyAsync(ctx->xq.ptr, X, sizeof(double) * (size_t)m * p, hipMemcpyHostToDevice, ctx->stream));
        else JCH_HIP(ctx, hipMemcpy2DAsync(ctx->xq.ptr, sizeof(double) * m, X, sizeof(double) * ldx, sizeof(double) * m, p,
                                           hipMemcpyHostToDevice, ctx->stream));
        dX = (const double *)ctx->xq.ptr; dO = (double *)ctx->gemm_out.ptr; ldxd = m; ldod = m;
    }
    JCH_TRY(jch_launch_affine_gemm(ctx, dX, m, (int)p, ldxd, dB, (int)k, kpad, dB + (size_t)p * kpad, dO, ldod));
    if (loc == JCH_LOC_HOST) {
        if (ldo == m) JCH_HIP(ctx, hipMemcpyAsync(out, dO, sizeof(double) * (size_t)m * k, hipMemcpyDeviceToHost, ctx->stream));
        else JCH_HIP(ctx, hipMemcpy2DAsync(out, sizeof(double) * ldo, dO, sizeof(double) * m, sizeof(double) * m, k,
                                           hipMemcpyDeviceToHost, ctx->stream));
    }
    JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // also keeps `hb` alive until the H2D copy is done
    return JCH_OK;
}

// sstot of `summary` (src/plskern.jl:250-251): sum_j (1/scale_j^2) * sum_i d_i (x_ij - shift_j)^2 — the K1
// second-moment kernel on X alone, combined on the host (p values).
extern "C" int32_t jch_weighted_ss(jch_ctx *ctx, int32_t loc, const double *X, int64_t n, int64_t p, int64_t ldx,
                                   const double *d, const double *shift, const double *scale, double *sstot)
{
    if (!ctx) return JCH_EINVAL;
    if (!X || !d || !sstot || n < 1 || p < 1 || ldx < n) return jch_fail(ctx, JCH_EINVAL, "jch_weighted_ss: bad arguments");
    if (loc != JCH_LOC_HOST && loc != JCH_LOC_DEVICE) return jch_fail(ctx, JCH_EINVAL, "jch_weighted_ss: bad loc");
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    const double *dX = X, *dd = d;
    int64_t ldxd = ldx;
    if (loc == JCH_LOC_HOST) {
        JCH_TRY(jch_reserve(ctx, ctx->xq, sizeof(double) * (size_t)n * p));
        JCH_TRY(jch_reserve(ctx, ctx->wstage, sizeof(double) * (size_t)n));
        if (ldx == n) JCH_HIP(ctx, hipMemcpyAsync(ctx->xq.ptr, X, sizeof(double) * (size_t)n * p, hipMemcpyHostToDevice, ctx->stream));
        else JCH_HIP(ctx, hipMemcpy2DAsync(ctx->xq.ptr, sizeof(double) * n, X, sizeof(double) * ldx, sizeof(double) * n, p,
                                           hipMemcpyHostToDevice, ctx->stream));
        JCH_HIP(ctx, hipMemcpyAsync(ctx->wstage.ptr, d, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        dX = (const double *)ctx->xq.ptr; dd = (const double *)ctx->wstage.ptr; ldxd = n;
    }
    JCH_TRY(jch_reserve(ctx, ctx->gemm_b, sizeof(double) * 2 * (size_t)p));
    double *dshift = (double *)ctx->gemm_b.ptr, *dvar = dshift + p;
    std::vector<double> hs((size_t)p, 0.0);
    if (shift) for (int64_t j = 0; j < p; ++j) hs[j] = shift[j];
    JCH_HIP(ctx, hipMemcpyAsync(dshift, hs.data(), sizeof(double) * (size_t)p, hipMemcpyHostToDevice, ctx->stream));
    JCH_TRY(jch_launch_moments(ctx, dX, ldxd, nullptr, 0, dd, n, (int)p, 0, dshift, dvar, false));
    std::vector<double> hv((size_t)p);
    JCH_HIP(ctx, hipMemcpyAsync(hv.data(), dvar, sizeof(double) * (size_t)p, hipMemcpyDeviceToHost, ctx->stream));
    JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double s = 0.0;
    for (int64_t j = 0; j < p; ++j) { const double sc = scale ? scale[j] : 1.0; s += hv[j] / (sc * sc); }
    *sstot = s;
    return JCH_OK;
}

// ---- named accessors of the §8b export list: thin host arithmetic over jch_affine_gemm --------------------------
// transform(object::Plsr, X; nlv) = cscale(X, xmeans, xscales) * R[:, 1:nlv]      (src/plskern.jl:187-195)
extern "C" int32_t jch_transform(jch_ctx *ctx, int32_t loc, const double *X, int64_t m, int64_t p, int64_t ldx,
                                 const double *xmeans, const double *xscales, const double *R, int32_t nlv, double *T, int64_t ldt)
{
    if (!ctx) return JCH_EINVAL;
    if (!R || nlv < 1) return jch_fail(ctx, JCH_EINVAL, "jch_transform: R is NULL or nlv < 1");
    return jch_affine_gemm(ctx, loc, X, m, p, ldx, xmeans, xscales, R, nlv, nullptr, T, ldt);
}

// predict(object::Plsr, X; nlv) for every nlv in [nlv_lo, nlv_hi] (0 allowed: intercept only), via coef
// (src/plskern.jl:207-217: B = diag(1/xscales) R_k C_k' diag(yscales), int = ymeans' - xmeans' B) and
// pred = int .+ X B (src/plskern.jl:226-238).  The coefficient blocks of the whole range are concatenated
// [B_lo | ... | B_hi] so X is read ONCE; block b of `pred` (columns b*q .. b*q+q-1) is the prediction at nlv_lo + b.
// `predict` over an nlv RANGE from the scores (round 4, second half): pred_a = ymeans + sum_{l <= a} t_l (c_l .* yscales)'
// (src/plskern.jl:207-217, :234-236 with B_a = R_a C_a': X_c B_a = T_a C_a') — the le prediction blocks are RUNNING SUMS over the
// score columns, so a thread keeps its rows' q sums in registers, adds one score column at a time and stores a block whenever the
// level is asked for: m nlv q fused multiply-adds instead of a GEMM with le q output columns, no matrix pipe, and every store
// instruction of a wave writes one contiguous V x 512-byte run of an output column.  The pass is its output (2.08 GB at cfg2
// against 0.2 GB of scores).
template <int QC, int V, bool NT>   // QC responses per thread (grid.y slices of QC), V rows per thread (2: 16-byte loads / stores)
__global__ __launch_bounds__(256) void k_predict_prefix(const double *__restrict__ T, int64_t m, int64_t ldt, const double *__restrict__ Cs,
                                                        const double *__restrict__ y0, int q, int lo, int hi, double *__restrict__ out, int64_t ldo)
{
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (row >= m) return;
    const int k0 = blockIdx.y * QC;
    double acc[QC][V];
#pragma unroll
    for (int k = 0; k < QC; ++k) {
        const double v = y0[min(k0 + k, q - 1)];
#pragma unroll
        for (int r = 0; r < V; ++r) acc[k][r] = v;
    }
    auto put = [&](int a) {
#pragma unroll
        for (int k = 0; k < QC; ++k) {
            if (k0 + k >= q) continue;
            double *dst = out + row + ((int64_t)(a - lo) * q + k0 + k) * ldo;
            if constexpr (V == 2) {
                typedef double d2 __attribute__((ext_vector_type(2)));
                d2 v; v.x = acc[k][0]; v.y = acc[k][1];
                if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(dst));
                else *reinterpret_cast<d2 *>(dst) = v;
            } else {
                if constexpr (NT) __builtin_nontemporal_store(acc[k][0], dst);
                else *dst = acc[k][0];
            }
        }
    };
    if (lo == 0) put(0);
    for (int a0 = 0; a0 < hi; a0 += 8) {
        double t[8][V];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double *src = T + row + (int64_t)min(a0 + u, hi - 1) * ldt;
            if constexpr (V == 2) {
                typedef double d2 __attribute__((ext_vector_type(2)));
                const d2 v = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(src));
                t[u][0] = v.x; t[u][1] = v.y;
            } else t[u][0] = __builtin_nontemporal_load(src);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int a = a0 + u + 1;          // level reached after adding score column a - 1
            if (a > hi) break;
#pragma unroll
            for (int k = 0; k < QC; ++k) {
                const double c = Cs[(size_t)(a - 1) * q + min(k0 + k, q - 1)];
#pragma unroll
                for (int r = 0; r < V; ++r) acc[k][r] += t[u][r] * c;
            }
            if (a >= lo) put(a);
        }
    }
}

static int32_t jch_launch_predict_prefix(jch_ctx *ctx, const double *T, int64_t m, int64_t ldt, const double *Cs, const double *y0, int q,
                                         int lo, int hi, double *out, int64_t ldo)
{
    const bool v2 = m % 2 == 0 && ldt % 2 == 0 && ldo % 2 == 0 && (((uintptr_t)T) & 15) == 0 && (((uintptr_t)out) & 15) == 0;
    const char *ent = getenv("JCH_PREDICT_NT");                 // (=0: plain stores — A/B runs)
    const bool nt = !(ent && atoi(ent) == 0);
    const int qc = q <= 4 ? 4 : (q <= 8 ? 8 : 16);
    const int64_t thr = v2 ? m / 2 : m;
    dim3 grid((unsigned)((thr + 255) / 256), (unsigned)((q + qc - 1) / qc));
#define JCH_PP(QC) do { \
        if (v2 && nt) hipLaunchKernelGGL((k_predict_prefix<QC, 2, true>), grid, dim3(256), 0, ctx->stream, T, m, ldt, Cs, y0, q, lo, hi, out, ldo); \
        else if (v2) hipLaunchKernelGGL((k_predict_prefix<QC, 2, false>), grid, dim3(256), 0, ctx->stream, T, m, ldt, Cs, y0, q, lo, hi, out, ldo); \
        else if (nt) hipLaunchKernelGGL((k_predict_prefix<QC, 1, true>), grid, dim3(256), 0, ctx->stream, T, m, ldt, Cs, y0, q, lo, hi, out, ldo); \
        else hipLaunchKernelGGL((k_predict_prefix<QC, 1, false>), grid, dim3(256), 0, ctx->stream, T, m, ldt, Cs, y0, q, lo, hi, out, ldo); } while (0)
    if (qc == 4) JCH_PP(4); else if (qc == 8) JCH_PP(8); else JCH_PP(16);
#undef JCH_PP
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

extern "C" int32_t jch_predict(jch_ctx *ctx, int32_t loc, const double *X, int64_t m, int64_t p, int64_t ldx, const double *xmeans,
                               const double *xscales, const double *ymeans, const double *yscales, const double *R,
                               const double *C, int64_t q, int32_t nlv_lo, int32_t nlv_hi, double *pred, int64_t ldo)
{
    if (!ctx) return JCH_EINVAL;
    if (!R || !C || q < 1 || nlv_lo < 0 || nlv_hi < nlv_lo) return jch_fail(ctx, JCH_EINVAL, "jch_predict: bad model arguments");
    const int64_t le = (int64_t)nlv_hi - nlv_lo + 1, kcols = le * q;
    // Three levels or more on a long input: the scores once (one pass over X: m x nlv_hi), then the prediction blocks as running
    // sums over the score columns (k_predict_prefix).  JCH_PREDICT_PREFIX=0: the one-GEMM path below for every range.
    const char *epp = getenv("JCH_PREDICT_PREFIX");
    if (le > 2 && m >= 4096 && !(epp && atoi(epp) == 0)) {
        if (!X || !pred || p < 1 || ldx < m || ldo < m) return jch_fail(ctx, JCH_EINVAL, "jch_predict: bad arguments");
        if (loc != JCH_LOC_HOST && loc != JCH_LOC_DEVICE) return jch_fail(ctx, JCH_EINVAL, "jch_predict: bad loc");
        JCH_HIP(ctx, hipSetDevice(ctx->device));
        const int kh = nlv_hi, kpad = (kh + 15) / 16 * 16;
        // host-side constants: Rs = diag(1 / xscales) R (p x kpad, row-major as the GEMM kernels take it), its bias -xmeans' Rs,
        // Cs[l][k] = C[k][l] yscales[k], ymeans
        std::vector<double> hb((size_t)p * kpad + kpad + (size_t)kh * q + q, 0.0);
        double *Bs = hb.data(), *b2 = Bs + (size_t)p * kpad, *Cs = b2 + kpad, *y0 = Cs + (size_t)kh * q;
        for (int c = 0; c < kh; ++c) {
            double acc = 0.0;
            for (int64_t j = 0; j < p; ++j) {
                const double v = R[j + (size_t)c * p] / (xscales ? xscales[j] : 1.0);
                Bs[j * kpad + c] = v;
                if (xmeans) acc -= xmeans[j] * v;
            }
            b2[c] = acc;
            for (int64_t k = 0; k < q; ++k) Cs[(size_t)c * q + k] = C[k + (size_t)c * q] * (yscales ? yscales[k] : 1.0);
        }
        for (int64_t k = 0; k < q; ++k) y0[k] = ymeans ? ymeans[k] : 0.0;
        JCH_TRY(jch_reserve(ctx, ctx->gemm_b, sizeof(double) * hb.size()));
        double *dB = (double *)ctx->gemm_b.ptr;
        JCH_HIP(ctx, hipMemcpyAsync(dB, hb.data(), sizeof(double) * hb.size(), hipMemcpyHostToDevice, ctx->stream));
        const int64_t ldt = (m + 1) & ~(int64_t)1;
        JCH_TRY(jch_reserve(ctx, ctx->tbuf, sizeof(double) * (size_t)ldt * kh));
        double *Tq = (double *)ctx->tbuf.ptr;
        const double *dX = X;
        double *dO = pred;
        int64_t ldxd = ldx, ldod = ldo;
        if (loc == JCH_LOC_HOST) {
            ldod = ldt;
            JCH_TRY(jch_reserve(ctx, ctx->xq, sizeof(double) * (size_t)m * p));
            JCH_TRY(jch_reserve(ctx, ctx->gemm_out, sizeof(double) * (size_t)ldod * kcols));
            if (ldx == m) JCH_HIP(ctx, hipMemcpyAsync(ctx->xq.ptr, X, sizeof(double) * (size_t)m * p, hipMemcpyHostToDevice, ctx->stream));
            else JCH_HIP(ctx, hipMemcpy2DAsync(ctx->xq.ptr, sizeof(double) * m, X, sizeof(double) * ldx, sizeof(double) * m, p,
                                               hipMemcpyHostToDevice, ctx->stream));
            dX = (const double *)ctx->xq.ptr; dO = (double *)ctx->gemm_out.ptr; ldxd = m;
        }
        JCH_TRY(jch_launch_affine_gemm(ctx, dX, m, (int)p, ldxd, dB, kh, kpad, dB + (size_t)p * kpad, Tq, ldt));
        JCH_TRY(jch_launch_predict_prefix(ctx, Tq, m, ldt, dB + (size_t)p * kpad + kpad, dB + (size_t)p * kpad + kpad + (size_t)kh * q, (int)q,
                                          nlv_lo, nlv_hi, dO, ldod));
        if (loc == JCH_LOC_HOST)
            JCH_HIP(ctx, hipMemcpy2DAsync(pred, sizeof(double) * ldo, dO, sizeof(double) * ldod, sizeof(double) * m, kcols,
                                          hipMemcpyDeviceToHost, ctx->stream));
        JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // also keeps `hb` alive until the H2D copy is done
        return JCH_OK;
    }
    std::vector<double> B((size_t)p * kcols, 0.0), b0((size_t)kcols, 0.0), cur((size_t)p * q, 0.0);
    for (int a = 0; a <= nlv_hi; ++a) {
        if (a > 0) {   // B_a = B_{a-1} + (r_a / xscales) (c_a * yscales)'
            const double *r = R + (size_t)(a - 1) * p, *c = C + (size_t)(a - 1) * q;
            for (int64_t k = 0; k < q; ++k) {
                const double ck = c[k] * (yscales ? yscales[k] : 1.0);
                for (int64_t j = 0; j < p; ++j) cur[j + k * p] += r[j] / (xscales ? xscales[j] : 1.0) * ck;
            }
        }
        if (a >= nlv_lo) {
            const int64_t blk = a - nlv_lo;
            for (int64_t k = 0; k < q; ++k) {
                double acc = ymeans ? ymeans[k] : 0.0;
                for (int64_t j = 0; j < p; ++j) {
                    const double v = cur[j + k * p];
                    B[j + (blk * q + k) * p] = v;
                    if (xmeans) acc -= xmeans[j] * v;
                }
                b0[blk * q + k] = acc;
            }
        }
    }
    return jch_affine_gemm(ctx, loc, X, m, p, ldx, nullptr, nullptr, B.data(), kcols, b0.data(), pred, ldo);
}
