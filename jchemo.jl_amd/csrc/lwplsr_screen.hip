// Screened kNN of the kNN-LWPLSR prediction path (round 4) — src/getknn.jl:29-57 (the k nearest training rows of every query in
// the score space, ties in row order), src/wdist.jl:64-75 (weights).
//
// k_knn_scan (lwplsr.hip) evaluates the exact f64 distance of every (row, query) pair on the vector pipe — 40 instructions per
// pair — and keeps a running bar per query; it is bound by the dependent chain of its trips (0.51 ms per 1000 queries at cfg5,
// 1e8 pairs).  The f64 matrix pipe runs at the vector rate on this chip, so it cannot screen; the bf16 one is 32 x faster.  This
// path:
//
//   pack      (once per model)  the training scores, centred on their column means, SPLIT into two bf16 pieces z ~ zh + zl
//             (16 significant bits) and laid out in the operand order of v_mfma_f32_32x32x16_bf16; the queries likewise,
//             -2 zq ~ qh + ql.  Three operand slots per score column — (zh, qh), (zh, ql), (zl, qh): everything of z.zq but the
//             zl.ql term — and four more for |z|^2 and |zq|^2 (two bf16 pieces each, against 1), so that ONE chain of matrix
//             instructions (4 of them for 20 score columns) delivers  a_ij ~ |z_i - zq_j|^2  for a 32 x 32 tile of pairs, in f32.
//             (First version of the round: f32 operands on v_mfma_f32_32x32x2_f32 — 12 instructions of twice the length per
//             tile; both passes ran at 0.57 of that pipe's peak: 54 and 86 us at cfg5.)
//   k_knn_gmin    pass 1 over all pairs: every lane keeps the minimum of a_ij over the rows of a GROUP (16 rows of each of T
//             tiles).  A value with k group minima at or below it is an upper bound of the k-th smallest a_ij (k groups hold k
//             different rows at or below it), and the k-th smallest group minimum is a tight one: with G = 4.5 k groups it is the
//             ~1.13 k-th smallest a_ij.
//   k_knn_bar     that value per query (one wave per query: sampled pivot, value bisection), widened by the error bound:
//             |a_ij - d_ij^2| <= eps_j = c (max_i |z_i|^2 + |zq_j|^2) for every row (ks_cfac: the 2^-16 of the two-piece operands
//             against the worst-case distance |z| + |zq|, the dropped zl.ql terms, the two-piece norms, K truncating f32
//             accumulations; x 1.25).  A row among the exact k nearest (ties included) has a_ij <= tau_j + 2 eps_j.
//   k_knn_survive pass 2: the same products, every a_ij against the bar; survivors (~1.15 k + the ties + the few within 2 eps) go
//             through a wave-private LDS list to the query's candidate list.
//   k_knn_finish_screen  one workgroup per query: EXACT distances of the candidates (the expression and column order of
//             k_knn_scan: the same bits) from a row-major f64 copy of the scores, (distance, index) order, the k nearest, the
//             weights (the tail shared with k_knn_finish).  A query whose list overflowed or came up short (non-finite scores)
//             is flagged ...
//   k_knn_scan    ... and done by the exact scan (lwplsr.hip; the groups of four queries with a flagged member only).  The bound
//             scales with the NORMS of the centred scores, the bar with neighbour DISTANCES: where the k-th distance^2 is below
//             ~1e-4 of the squared norms (score spaces of low intrinsic dimension and wide range) the bar admits thousands of
//             rows and most queries end up here; a prepared model that sees a quarter of a call's queries flagged stops
//             screening (lw_run).
//
// Neighbours, their order, distances and weights are identical to k_knn_scan's: the screen only decides WHICH rows get an exact
// distance, and it never drops a row at or below the exact k-th distance.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>
#include <stdio.h>

#include "jch_internal.h"
#include "lv_device.h"
#include "lwplsr_dev.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ks_bf16x8 __attribute__((ext_vector_type(8)));

#define KS_CCAP 2048        // candidate rows per query (global); more survivors: the query is flagged for the generic selection
#define KS_S 8              // survivors a (row chunk, query) pair keeps in its own slots of the candidate array; more: the query's overflow list
#define KS_PAD 1.0e30f      // |z|^2 of the pad rows of the last tile: never below a bar, never a group minimum that counts
#define KS_MAXSLOTS 1024    // group slots per query tile: G = 2 x slots <= 2048 values per query in k_knn_bar (32 per lane)

struct ks_args {
    knn_args a;
    knn_screen sc;
    uint4 *Qs;          // [nqt][KS][64] query operand (8 bf16 per lane and k-step)
    double *nq;         // [nqt * 32] |zq - mu|^2 of the f32-rounded query (f64 sum of the rounded values)
    float *gmin;        // [nqt][nslots][64]
    float *bar;         // [nqt * 32]
    int *cnt;           // [nqt * 32] entries of the query's overflow list
    int *cand;          // [m][KS_CCAP] overflow list (global atomics: only where one row chunk holds more than KS_S survivors of a query)
    int *scand;         // [m][nchunks][KS_S] rows of the survivors of (query, row chunk), -1 in the unused places: every place is
                        // written by the pair's one wave on every call (32 contiguous bytes), nothing to clear, no count to read
    int *flags;         // [m]
    int nqt, nslots, T;
    int gpw, nchunks, qw;      // k_knn_survive: group slots per row chunk, chunks, tiles of 32 queries per wave item (1, 2)
    int gpw_g, nchunks_g, qw_g;   // k_knn_gmin: the same (its own split of the work: it writes per group slot, not per chunk)
    double cfac;
};

// ---------------------------------------------------------------- model-constant part
#define KS_MB 32   // row blocks of the column sums
__global__ __launch_bounds__(256) void k_ks_colsum(const double *__restrict__ Zt, int64_t ldzt, int64_t n, double *__restrict__ part)
{
    __shared__ double red[4];
    const int c = blockIdx.x, blk = blockIdx.y, tid = threadIdx.x;
    const double *col = Zt + (size_t)c * (size_t)ldzt;
    const int64_t per = (n + KS_MB - 1) / KS_MB, lo = blk * per, hi = min(n, lo + per);
    double s = 0.0;
    for (int64_t i = lo + tid; i < hi; i += 256) s += col[i];
    s = jch_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) part[c * KS_MB + blk] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void k_ks_colmean(const double *__restrict__ part, int dd, int64_t n, double *__restrict__ mu, unsigned *__restrict__ hdr)
{
    for (int c = threadIdx.x; c < dd; c += 64) {
        double t = 0.0;
        for (int b = 0; b < KS_MB; ++b) t += part[c * KS_MB + b];
        t /= (double)n;
        mu[c] = (t == t && fabs(t) < 1e300) ? t : 0.0;           // (non-finite scores: the pack kernel raises hdr[1])
    }
    if (threadIdx.x == 0) { hdr[0] = 0u; hdr[1] = 0u; }
}

// f32 -> bf16, round to nearest even (Inf stays Inf, NaN stays NaN)
__device__ __forceinline__ unsigned ks_bf16(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float ks_bf16_f(unsigned b) { return __uint_as_float(b << 16); }
// the two bf16 pieces of f (hi + lo carries 16 significant bits of f)
__device__ __forceinline__ void ks_split(float f, unsigned &hi, unsigned &lo)
{
    hi = ks_bf16(f);
    lo = ks_bf16(f - ks_bf16_f(hi));
}

// Row order of the operand copy (ks_row): tile t holds the rows t, t + ntiles, t + 2 ntiles, ... (local row i <-> row i ntiles + t), and
// group slot s the tiles s, s + nslots, ...: CONSECUTIVE training rows sit in different tiles and different groups.  Training sets
// are often ordered (by time, by sample, by class): a query's neighbours are then runs of consecutive rows, and with rows grouped
// as they come the k-th smallest group minimum would be the distance of the k-th nearest RUN, not row — a useless bar.
// Operand slots (the K dimension of the products), dd score columns: slot 3 c + t, t = 0, 1, 2 -> A (zh, zh, zl), B (qh, ql, qh);
// slots 3 dd, 3 dd + 1 -> A the two pieces of |z|^2, B 1; slots 3 dd + 2, 3 dd + 3 -> A 1, B the two pieces of |zq|^2; zero beyond.
// v_mfma_f32_32x32x16_bf16, k-step s: lane l holds slots 16 s + 8 (l / 32) + 0 .. 7 of row (A) / query (B) l % 32.
// one thread per (tile, lane)
template <int KS>
__global__ __launch_bounds__(256) void k_ks_pack_rows(const double *__restrict__ Zt, int64_t ldzt, int64_t n, int dd, const double *__restrict__ mu,
                                                      uint4 *__restrict__ Zs, int64_t ntiles, unsigned *__restrict__ hdr)
{
    // thread -> (operand lane, tile), consecutive threads on consecutive TILES: their rows (ks_row) are consecutive too, so the column
    // reads coalesce; the 16-byte operand stores are 1 KB x KS apart instead (tile-major threads, coalesced stores and reads 8 n / 32
    // bytes apart, took 84 us at cfg5)
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = (int)(gid / ntiles);
    const int64_t tile = gid - (int64_t)lane * ntiles;
    if (lane >= 64) return;
    const int64_t row = (int64_t)(lane & 31) * ntiles + tile;      // (ks_row: consecutive rows sit in different tiles)
    const int h = lane >> 5;
    const bool live = row < n;
    double nz = 0.0;
    for (int c = 0; c < dd; ++c) {
        const double zv = live ? Zt[(size_t)row + (size_t)c * (size_t)ldzt] : 0.0;
        unsigned hi, lo;
        ks_split((float)(zv - mu[c]), hi, lo);
        const double zz = (double)ks_bf16_f(hi) + (double)ks_bf16_f(lo);
        nz += zz * zz;
    }
    float nzf = (float)nz;
    if ((double)nzf < nz) nzf = __uint_as_float(__float_as_uint(nzf) + 1u);   // (rounded UP: it enters the error bound through hdr[0])
    unsigned n1, n2;
    ks_split(live ? (float)nz : KS_PAD, n1, n2);
    unsigned short v[KS * 8];
#pragma unroll
    for (int sl = 0; sl < KS * 8; ++sl) {
        const int kk = 16 * (sl >> 3) + 8 * h + (sl & 7);
        const int c = kk / 3, t = kk - 3 * c;
        unsigned val = 0u;
        if (c < dd) {
            unsigned hi, lo;
            ks_split(live ? (float)(Zt[(size_t)row + (size_t)c * (size_t)ldzt] - mu[c]) : 0.0f, hi, lo);
            val = t == 2 ? lo : hi;
        } else if (kk == 3 * dd) val = n1;
        else if (kk == 3 * dd + 1) val = n2;
        else if (kk == 3 * dd + 2 || kk == 3 * dd + 3) val = 0x3f80u;          // 1.0
        v[sl] = (unsigned short)val;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        Zs[((size_t)tile * KS + ks) * 64 + lane] = make_uint4(v[8 * ks] | ((unsigned)v[8 * ks + 1] << 16), v[8 * ks + 2] | ((unsigned)v[8 * ks + 3] << 16),
                                                            v[8 * ks + 4] | ((unsigned)v[8 * ks + 5] << 16), v[8 * ks + 6] | ((unsigned)v[8 * ks + 7] << 16));
    if (live && h == 0) {
        if (!(nzf < 1.0e29f)) atomicOr(hdr + 1, 1u);              // NaN / Inf / beyond the pad value: no screen for this model
        else atomicMax(hdr, __float_as_uint(nzf));                // (non-negative floats order like their bits)
    }
}

// The query operand of one lane (query tile qt): its 8 KS slots of query j = 32 qt + lane % 32, and |zq|^2 of the two-piece query
// (rounded up to f32).  Every score column is loaded once, all loads in flight together; the slots are picked with compile-time
// indices for both halves of the wave and selected by the lane's half.  (Measured and dropped: the waves of k_knn_gmin building
// their operand with this themselves instead of a kernel in front of them — 11 us per call at cfg5 for 2048 threads' worth of work
// —: 7152 waves each running this once took the pass from 17 to 48 us.)
template <int KS>
__device__ __forceinline__ void ks_make_b(const ks_args &g, int qt, int lane, uint4 (&b)[KS], float &nqf)
{
    constexpr int DMAX = (16 * KS - 4) / 3;                        // score columns KS k-steps can hold: 3 dd + 4 <= 16 KS
    const int j = qt * 32 + (lane & 31), h = lane >> 5, dd = g.a.dd;
    const bool live = j < g.a.m;
    const double *zqp = g.a.Zq + (size_t)min(j, g.a.m - 1);
    float fv[DMAX];
#pragma unroll
    for (int c = 0; c < DMAX; ++c) { const int cc = min(c, dd - 1); fv[c] = -2.0f * (float)(zqp[(size_t)cc * (size_t)g.a.ldzq] - g.sc.mu[cc]); }
    unsigned short hi[DMAX], lo[DMAX];
    double nq = 0.0;
#pragma unroll
    for (int c = 0; c < DMAX; ++c) {
        unsigned h_, l_;
        ks_split((live && c < dd) ? fv[c] : 0.0f, h_, l_);
        hi[c] = (unsigned short)h_; lo[c] = (unsigned short)l_;
        const double qq = 0.5 * ((double)ks_bf16_f(h_) + (double)ks_bf16_f(l_));
        nq += qq * qq;
    }
    nqf = (float)nq;
    if ((double)nqf < nq) nqf = __uint_as_float(__float_as_uint(nqf) + 1u);
    unsigned m1, m2;
    ks_split((float)nq, m1, m2);
    auto slot = [&](int kk) -> unsigned {                          // (kk is a compile-time constant at every call)
        const int c = kk / 3, t = kk - 3 * c;
        unsigned val = 0u;
        if (c < DMAX) val = t == 1 ? lo[c] : hi[c];                  // (columns >= dd hold zeros: overwritten below where the norm slots sit)
        if (kk == 3 * dd || kk == 3 * dd + 1) val = 0x3f80u;        // 1.0
        else if (kk == 3 * dd + 2) val = m1;
        else if (kk == 3 * dd + 3) val = m2;
        else if (kk > 3 * dd + 3) val = 0u;
        return val;
    };
    unsigned short v[KS * 8];
#pragma unroll
    for (int sl = 0; sl < KS * 8; ++sl) {
        const int k0 = 16 * (sl >> 3) + (sl & 7);
        const unsigned v0 = slot(k0), v1 = slot(k0 + 8);
        v[sl] = (unsigned short)(h ? v1 : v0);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        b[ks] = make_uint4(v[8 * ks] | ((unsigned)v[8 * ks + 1] << 16), v[8 * ks + 2] | ((unsigned)v[8 * ks + 3] << 16),
                           v[8 * ks + 4] | ((unsigned)v[8 * ks + 5] << 16), v[8 * ks + 6] | ((unsigned)v[8 * ks + 7] << 16));
}

// one thread per (query tile, lane)
template <int KS>
__global__ __launch_bounds__(256) void k_ks_pack_queries(ks_args g)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int qt = gid >> 6, lane = gid & 63;
    if (qt >= g.nqt) return;
    uint4 b[KS];
    float nqf;
    ks_make_b<KS>(g, qt, lane, b, nqf);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) g.Qs[((size_t)qt * KS + ks) * 64 + lane] = b[ks];
    if (lane < 32) g.nq[qt * 32 + lane] = (double)nqf;
}

// ---------------------------------------------------------------- the two passes over all pairs
template <int KS>
__device__ __forceinline__ void ks_load_b(const ks_args &g, int qt, int lane, uint4 (&b)[KS])
{
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) b[ks] = g.Qs[((size_t)qt * KS + ks) * 64 + lane];
}
template <int KS>
__device__ __forceinline__ void ks_load_a(const uint4 *__restrict__ Zs, int64_t tile, int lane, uint4 (&a)[KS])
{
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = Zs[((size_t)tile * KS + ks) * 64 + lane];
}
// a_ij of one 32 x 32 tile of pairs: acc[r] of lane l <-> row 8 (r / 4) + 4 (l / 32) + r % 4, query column l % 32
template <int KS>
__device__ __forceinline__ f32x16 ks_tile(const uint4 (&a)[KS], const uint4 (&b)[KS])
{
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ks_bf16x8, a[ks]), __builtin_bit_cast(ks_bf16x8, b[ks]), acc, 0, 0, 0);
    return acc;
}

// Workgroup -> (row chunk, four query groups), XCD-aware: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so
// workgroup b runs on XCD b % 8; all the workgroups of a row chunk are given the same residue — the chunk's operand tiles cross the
// fabric once, into ONE L2, and every other wave that needs them hits there (the operand copy, 12.8 MB at cfg5, does not fit one L2:
// with chunks dealt to all XCDs every XCD pulled all of it from the Infinity Cache, one tile ahead of its use).
__device__ __forceinline__ bool ks_item(int nchunks, int nqg, int &qg, int &chunk)
{
    const int nqb = (nqg + 3) >> 2;
    const int b = blockIdx.x, xcd = b & 7, i = b >> 3;
    chunk = (i / nqb) * 8 + xcd;
    qg = (i % nqb) * 4 + (int)(threadIdx.x >> 6);
    return chunk < nchunks && qg < nqg;
}

// wave item: query group qg (QW tiles of 32 queries: every loaded row operand serves 32 QW queries — with QW = 1 both passes
// ran into the L2 bandwidth: 4 KB of operand per 1024 pairs, 17 TB/s at cfg5), row chunk w / nqg (gpw group slots); the four waves of
// a workgroup share the chunk
template <int KS, int QW>
__global__ __launch_bounds__(256) void k_knn_gmin(ks_args g)
{
    const int lane = threadIdx.x & 63;
    const int nqg = (g.nqt + QW - 1) / QW;
    int qg, chunk;
    if (!ks_item(g.nchunks_g, nqg, qg, chunk)) return;
    uint4 b[QW][KS];
#pragma unroll
    for (int u = 0; u < QW; ++u) ks_load_b<KS>(g, min(qg * QW + u, g.nqt - 1), lane, b[u]);
    for (int gi = 0; gi < g.gpw_g; ++gi) {
        const int slot = chunk * g.gpw_g + gi;
        if (slot >= g.nslots) break;
        float mn[QW];
#pragma unroll
        for (int u = 0; u < QW; ++u) mn[u] = __builtin_inff();
        // the slot's tiles (slot, slot + nslots, ...) through two operand buffers, the next tile's loads behind this tile's products.
        // (Written as one buffer + a copy at the end of the iteration the compiler waited for the loads at the copies, and without the
        // scheduling barriers it sank them to their use: every tile then waited for its own operands — 25 us instead of 12.)
        uint4 a0[KS], a1[KS];
        auto tile_min = [&](const uint4 (&a)[KS]) {
#pragma unroll
            for (int u = 0; u < QW; ++u) {
                const f32x16 acc = ks_tile<KS>(a, b[u]);
#pragma unroll
                for (int r = 0; r < 16; ++r) mn[u] = fminf(mn[u], acc[r]);  // (a NaN never replaces a number)
            }
        };
        int64_t t = slot;
        ks_load_a<KS>(g.sc.Zs, t, lane, a0);
        for (;;) {
            const int64_t t1 = t + g.nslots;
            const bool has1 = t1 < g.sc.ntiles;
            ks_load_a<KS>(g.sc.Zs, has1 ? t1 : t, lane, a1);
            __builtin_amdgcn_sched_barrier(0);
            tile_min(a0);
            if (!has1) break;
            const int64_t t2 = t1 + g.nslots;
            const bool has2 = t2 < g.sc.ntiles;
            ks_load_a<KS>(g.sc.Zs, has2 ? t2 : t1, lane, a0);
            __builtin_amdgcn_sched_barrier(0);
            tile_min(a1);
            if (!has2) break;
            t = t2;
        }
#pragma unroll
        for (int u = 0; u < QW; ++u)
            if (qg * QW + u < g.nqt) g.gmin[((size_t)(qg * QW + u) * g.nslots + slot) * 64 + lane] = mn[u];   // ([query tile][lane][slot] — contiguous reads for k_knn_bar — was measured: bar 10.5 -> 6.8 us, this kernel 17 -> 21.8 with its scattered stores)
    }
}

// one wave per query: the k-th smallest of its 2 nslots group minima, widened to the bar; zeroes the query's candidate counter
template <int NV>
__global__ __launch_bounds__(256) void k_knn_bar(ks_args g)
{
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= g.nqt * 32) return;
    const int qt = j >> 5, c = j & 31, G = 2 * g.nslots;
    unsigned u[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i;
        float f = __builtin_inff();
        if (v < G) f = g.gmin[((size_t)qt * g.nslots + (v >> 1)) * 64 + c + 32 * (v & 1)];
        const unsigned bits = __float_as_uint(f);
        u[i] = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);         // unsigned order = float order (a_ij may round below zero)
    }
    // ANY value with at least k group minima at or below it is a valid bar, so the wave first sorts 64 SAMPLES (each lane's first
    // value) across its lanes and bisects for the smallest sample with >= k values at or below it: 6 counting steps instead of
    // the 32 of the exact radix descent, for a bar ~G / 128 ranks above the k-th smallest (+3 % survivors at cfg5).  No sample
    // with k values below it (k close to G): the exact k-th smallest.
    unsigned prefix = 0u;
    {
        unsigned sv = u[0];
#pragma unroll
        for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                const unsigned pv = (unsigned)__shfl_xor((int)sv, stride, 64);
                const bool keepmin = ((lane & stride) == 0) == ((lane & size) == 0);
                sv = keepmin ? min(pv, sv) : max(pv, sv);
            }
        auto count = [&](unsigned pv) {
            int n = 0;
#pragma unroll
            for (int i = 0; i < NV; ++i) n += __popcll(__ballot(u[i] <= pv));
            return n;
        };
        if (count((unsigned)__shfl((int)sv, 63, 64)) >= g.a.k) {
            int lo = 0, hi = 63;
            while (lo < hi) {                                               // (wave-uniform)
                const int mid = (lo + hi) >> 1;
                if (count((unsigned)__shfl((int)sv, mid, 64)) >= g.a.k) hi = mid; else lo = mid + 1;
            }
            // (lo_v, hi_v]: the sample below the pivot (fewer than k values at or below it) and the pivot.  Value bisection on the keys
            // until at most k + k / 64 + 1 values lie at or below the bar — a few steps when the samples are representative, up to 32
            // (the exact radix descent) when they are not: the samples are each lane's FIRST value, i.e. the first 64 group slots, and a
            // query at the edge of an ordered training set has all its near groups elsewhere (2086 survivors instead of ~230, found by
            // test_screen_with_sorted_rows_and_a_wide_range)
            unsigned hi_v = (unsigned)__shfl((int)sv, hi, 64);
            unsigned lo_v = hi > 0 ? (unsigned)__shfl((int)sv, hi - 1, 64) : 0u;
            int c_hi = count(hi_v);
            const int want = g.a.k + (g.a.k >> 6) + 1;
            for (int it = 0; it < 32 && hi_v - lo_v > 1u && c_hi > want; ++it) {
                const unsigned mid_v = lo_v + ((hi_v - lo_v) >> 1);
                const int cm = count(mid_v);
                if (cm >= g.a.k) { hi_v = mid_v; c_hi = cm; } else lo_v = mid_v;
            }
            prefix = hi_v;
        } else {
            int need = g.a.k;
            for (int bit = 31; bit >= 0; --bit) {                           // (wave-uniform)
                const unsigned hi = bit == 31 ? 0u : (0xffffffffu << (bit + 1));
                int cnt0 = 0;
#pragma unroll
                for (int i = 0; i < NV; ++i) cnt0 += __popcll(__ballot((u[i] & (hi | (1u << bit))) == prefix));
                if (need > cnt0) { need -= cnt0; prefix |= 1u << bit; }
            }
        }
    }
    if (lane == 0) {
        const unsigned bits = (prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix;
        const double tau = (double)__uint_as_float(bits);
        const double zmax2 = (double)__uint_as_float(g.sc.hdr[0]);
        const double eps = g.cfac * (zmax2 + g.nq[j]);
        const double barx = tau + 2.0 * eps;
        float bf = -__builtin_inff();                                       // no screen for this query: no survivor, flagged by the finish
        if (g.sc.hdr[1] == 0u && barx < 1.0e29) {
            bf = (float)barx;
            if ((double)bf < barx) bf = __uint_as_float(__float_as_uint(bf) + (bf >= 0.f ? 1u : -1u));   // round UP
        }
        g.bar[j] = bf;
        g.cnt[j] = 0;
    }
}

// Survivors: a (row chunk, query) pair has its OWN slots in the candidate array — one writer, no counter to share.  (First version:
// every survivor took its place in the query's list by a global atomic with return; 245 k of them per call at cfg5, on 1000
// addresses, across 8 L2s: 50 of the kernel's 70 us.)  Inside the tile loop a survivor only goes to a wave-private list (its place
// from the ballot: no atomic, no wait); ks_place hands the list's entries to their queries' slots when the wave is done, or the
// list full.
#define KS_LCAP 512         // wave-private list of survivors: row << 6 | query column (0 .. 63)
// (Measured and dropped: the exact distance of every entry computed here, as it is placed — the ~230 k random 160-byte reads per call
// overlapping the other waves' products instead of standing in the finishing kernel: survive 36 -> 77 us at three waves per SIMD,
// the finishing kernel unchanged at 45: what costs there is the NUMBER of scattered sectors it reads, the rows' and the
// candidates'.)  (The fields by value: a reference to the kernel's argument block would put a copy of the block on the stack.)
__device__ __noinline__ void ks_place(int *cand, int *cnt, int m, const unsigned *list, int nl, int *mycnt, int *myslots, int qbase)
{
    const int lane = threadIdx.x & 63;
    wavesync();
    for (int e = lane; e < nl; e += 64) {
        const unsigned ent = list[e];
        const int c2 = (int)(ent & 63u), row = (int)(ent >> 6), j = qbase + c2;
        const int pos = atomicAdd(&mycnt[c2], 1);                           // (LDS)
        if (pos < KS_S) myslots[c2 * KS_S + pos] = row;
        else if (j < m) {                                                   // the chunk's slots are full: the query's overflow list
            const int op = atomicAdd(&cnt[j], 1);
            if (op < KS_CCAP) cand[(size_t)j * KS_CCAP + op] = row;
        }
    }
    wavesync();
}

// (five waves per SIMD forced through amdgpu_waves_per_eu: 96 registers + 11 spilled, 32 -> 41 us)
template <int KS, int QW>
__global__ __launch_bounds__(256) void k_knn_survive(ks_args g)
{
    __shared__ int wcnt[4][32 * QW];
    __shared__ int wslots[4][32 * QW * KS_S];
    __shared__ unsigned wlist[4][KS_LCAP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, col = lane & 31;
    const int nqg = (g.nqt + QW - 1) / QW;
    int qg, chunk;
    if (!ks_item(g.nchunks, nqg, qg, chunk)) return;
    uint4 b[QW][KS];
    float barv[QW];
#pragma unroll
    for (int u = 0; u < QW; ++u) {
        const int qt = min(qg * QW + u, g.nqt - 1);
        ks_load_b<KS>(g, qt, lane, b[u]);
        barv[u] = qg * QW + u < g.nqt ? g.bar[qt * 32 + col] : -__builtin_inff();
    }
    int *mycnt = wcnt[wv], *myslots = wslots[wv];
    unsigned *mylist = wlist[wv];
    for (int e = lane; e < 32 * QW; e += 64) mycnt[e] = 0;
    wavesync();
    // the chunk's tiles: slots s0 .. s1 - 1 of every stride of nslots tiles
    const int s0 = chunk * g.gpw, s1 = min(g.nslots, s0 + g.gpw);
    const int64_t nt = g.sc.ntiles;
    auto next_tile = [&](int64_t t) {                                       // the tile after t in this chunk's order, or -1
        int64_t base = t / g.nslots * g.nslots;
        int64_t tn = t + 1;
        if (tn - base >= s1) { base += g.nslots; tn = base + s0; }
        return tn < nt ? tn : (int64_t)-1;
    };
    const int lrow = 4 * (lane >> 5);
    const unsigned long long below = (1ull << lane) - 1ull;
    int nl = 0;                                                             // (wave-uniform)
    auto tile_test = [&](const uint4 (&a)[KS], int64_t t) {
#pragma unroll
        for (int u = 0; u < QW; ++u) {
            const f32x16 acc = ks_tile<KS>(a, b[u]);
            // all 16 comparisons first (their masks in scalar registers), then the branches: written as compare-and-branch per
            // register every branch waited for its own comparison — 25 of the kernel's 40 us
            unsigned long long m[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) m[r] = __ballot(acc[r] <= barv[u]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (m[r]) {                                                 // (wave-uniform; 1 register in 7 at cfg5)
                    if (nl > KS_LCAP - 64) { ks_place(g.cand, g.cnt, g.a.m, mylist, nl, mycnt, myslots, qg * QW * 32); nl = 0; }
                    if ((m[r] >> lane) & 1ull) {
                        const unsigned row = (unsigned)((lrow + 8 * (r >> 2) + (r & 3)) * (int)nt + (int)t);   // (ks_row)
                        mylist[nl + __popcll(m[r] & below)] = (row << 6) | (unsigned)(32 * u + col);
                    }
                    nl += __popcll(m[r]);
                }
            }
        }
    };
    // (two operand buffers, as in k_knn_gmin)
    uint4 a0[KS], a1[KS];
    int64_t t = s0 < nt ? (int64_t)s0 : (int64_t)-1;
    if (t >= 0) ks_load_a<KS>(g.sc.Zs, t, lane, a0);
    while (t >= 0) {
        const int64_t t1 = next_tile(t);
        ks_load_a<KS>(g.sc.Zs, t1 >= 0 ? t1 : t, lane, a1);
        __builtin_amdgcn_sched_barrier(0);
        tile_test(a0, t);
        if (t1 < 0) break;
        const int64_t t2 = next_tile(t1);
        ks_load_a<KS>(g.sc.Zs, t2 >= 0 ? t2 : t1, lane, a0);
        __builtin_amdgcn_sched_barrier(0);
        tile_test(a1, t1);
        t = t2;
    }
    if (nl) ks_place(g.cand, g.cnt, g.a.m, mylist, nl, mycnt, myslots, qg * QW * 32);
    wavesync();
    // this chunk's slots of every query of the group: all KS_S places (-1 beyond the count), 32 contiguous bytes per query
    for (int e = lane; e < 32 * QW; e += 64) {
        const int j = qg * QW * 32 + e;
        if (j < g.a.m) {
            const int cc = min(mycnt[e], KS_S);
            int v[KS_S];
#pragma unroll
            for (int i = 0; i < KS_S; ++i) v[i] = i < cc ? myslots[e * KS_S + i] : -1;
            int4 *dst = reinterpret_cast<int4 *>(g.scand + ((size_t)j * g.nchunks + chunk) * KS_S);
#pragma unroll
            for (int i = 0; i < KS_S / 4; ++i) dst[i] = make_int4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
        }
    }
}

// one workgroup per query: exact distances of the candidates, order, weights — or the flag for the generic selection
__global__ __launch_bounds__(256) void k_knn_finish_screen(ks_args g)
{
    extern __shared__ __attribute__((aligned(16))) double zq[];   // [dd]
    __shared__ double key[KS_CCAP];
    __shared__ int idx[KS_CCAP];
    __shared__ double okey[KNN_CAP];
    __shared__ int oidx[KNN_CAP];
    __shared__ double sred[8];
    __shared__ double smed[2];
    __shared__ int snn[4];
    __shared__ int tot;
    const int tid = threadIdx.x;
    const int qi = blockIdx.x, k = g.a.k, dd = g.a.dd;
    if (tid == 0) tot = 0;
    for (int e = tid; e < dd; e += 256) zq[e] = g.a.Zq[(size_t)qi + (size_t)e * (size_t)g.a.ldzq];
    for (int e = tid; e < k; e += 256) { okey[e] = __builtin_inf(); oidx[e] = 0x7fffffff; }
    __syncthreads();
    // gather: the chunks' own slots (one contiguous read of the query's nchunks x 32 bytes), then the overflow list (any order: the
    // entries are put in (distance, index) order below)
    const int dbg = g.a.dbg;   // JCH_KNN_SCREEN_DBG (measurement; results then wrong by design): 2 no distances, 4 no ordering, 8 no tail, 16 no gather
    const int oc = g.cnt[qi];
    const int4 *sl = reinterpret_cast<const int4 *>(g.scand + (size_t)qi * g.nchunks * KS_S);
    if (dbg & 16) { for (int e = tid; e < k + 20; e += 256) idx[e] = e * 37; if (tid == 0) tot = k + 20; }
    else
    for (int ch = tid; ch < g.nchunks; ch += 256) {
        int v[KS_S];
#pragma unroll
        for (int i = 0; i < KS_S / 4; ++i) { const int4 t4 = sl[ch * (KS_S / 4) + i]; v[4 * i] = t4.x; v[4 * i + 1] = t4.y; v[4 * i + 2] = t4.z; v[4 * i + 3] = t4.w; }
        int cc = 0;
#pragma unroll
        for (int i = 0; i < KS_S; ++i) cc += v[i] >= 0 ? 1 : 0;
        if (cc > 0) {
            const int pos = atomicAdd(&tot, cc);
#pragma unroll
            for (int i = 0; i < KS_S; ++i)
                if (i < cc && pos + i < KS_CCAP) idx[pos + i] = v[i];
        }
    }
    for (int e = tid; e < min(oc, KS_CCAP); e += 256) {
        const int pos = atomicAdd(&tot, 1);
        if (pos < KS_CCAP) idx[pos] = g.cand[(size_t)qi * KS_CCAP + e];
    }
    __syncthreads();
    const int c = tot;
    if (oc > KS_CCAP || c > KS_CCAP || c < k) {                   // (block-uniform) overflow, or a query / model with non-finite scores
        if (tid == 0) g.flags[qi] = 1;
        return;
    }
    if (tid == 0) g.flags[qi] = 0;
    int cap = 256;
    while (cap < c) cap <<= 1;
    for (int e = tid; e < cap; e += 256) {
        double acc = __builtin_inf();
        int row = 0x7fffffff;
        if (e < c) {
            row = idx[e];
            acc = 0.0;
            if (dbg & 2) { acc = (double)((row * 2654435761u) >> 8); key[e] = acc; continue; }
            // (the expression and column order of k_knn_scan: the same bits; the row from the row-major copy: 8 dd contiguous bytes
            // in whole 64-byte sectors instead of dd sectors 8 n bytes apart — those were 40 of this kernel's 66 us at cfg5)
            const double2 *zr = reinterpret_cast<const double2 *>(g.sc.Zr + (size_t)row * g.sc.ldzr);
            for (int c0 = 0; c0 < dd; c0 += 16) {
                double2 x[8];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) x[cc] = zr[min(c0 / 2 + cc, (g.sc.ldzr >> 1) - 1)];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    if (c0 + 2 * cc < dd) { const double d = x[cc].x - zq[c0 + 2 * cc]; acc += d * d; }
                    if (c0 + 2 * cc + 1 < dd) { const double d = x[cc].y - zq[c0 + 2 * cc + 1]; acc += d * d; }
                }
            }
        }
        key[e] = acc; idx[e] = row;
    }
    __syncthreads();
    if (dbg & 4) { for (int e = tid; e < k; e += 256) { okey[e] = key[e]; oidx[e] = idx[e]; } }
    else
    if (c <= 512) {
        // segments of 64 entries (four, or eight: two per wave), each put in order by one wave on its registers, then MERGED BY RANK: an
        // entry's place is its place in its own segment plus, for every other segment, the number of entries there that come before
        // it — a binary search each (two barriers; the workgroup-wide bitonic sort has 36 and more, and ordering by rank over ALL
        // entries — c steps per entry — was measured at 24 us of this kernel's 55)
        const int nseg = cap >> 6, per = cap >> 8;                 // cap = 256 or 512: 1 or 2 entries per thread
        double kv[2];
        int iv[2];
        for (int u = 0; u < per; ++u) { kv[u] = key[tid + 256 * u]; iv[u] = idx[tid + 256 * u]; knn_sort64_lanes(kv[u], iv[u]); }
        __syncthreads();
        for (int u = 0; u < per; ++u) { key[tid + 256 * u] = kv[u]; idx[tid + 256 * u] = iv[u]; }
        __syncthreads();
        for (int u = 0; u < per; ++u) {
            const int s_ = (tid + 256 * u) >> 6;
            int rank = tid & 63;
            for (int s2 = 0; s2 < nseg; ++s2) {
                if (s2 == s_) continue;
                const double *kl = key + 64 * s2;
                const int *il = idx + 64 * s2;
                int lo = 0, hi = 64;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const double kb = kl[mid];
                    if ((kb < kv[u]) || (kb == kv[u] && il[mid] < iv[u])) lo = mid + 1; else hi = mid;
                }
                rank += lo;
            }
            if (rank < k && iv[u] != 0x7fffffff) { okey[rank] = kv[u]; oidx[rank] = iv[u]; }
        }
    } else {
        bitonic_sort_n<256>(key, idx, cap);
        for (int e = tid; e < k; e += 256) { okey[e] = key[e]; oidx[e] = idx[e]; }
    }
    __syncthreads();
    if (dbg & 8) { for (int e = tid; e < k; e += 256) { g.a.ind[(size_t)qi * k + e] = oidx[e] < 0 || oidx[e] >= g.a.n ? e : oidx[e]; g.a.dist[(size_t)qi * k + e] = okey[e]; g.a.w[(size_t)qi * k + e] = 1.0; } return; }
    knn_finish_tail(g.a, qi, k, key, okey, oidx, sred, smed, snn);
}

// ---------------------------------------------------------------- host
// |a_ij - d_ij^2| <= ks_cfac (|z_i|^2 + |zq_j|^2): two-piece operands (each within 2^-16 of its value) against the worst-case
// distance |z| + |zq|: 2^-14; the dropped zl.ql terms and the two-piece norms: 2 x 2^-16; K = 3 dd + 4 truncating f32 accumulations
// of terms that sum to <= 2.02 (|z|^2 + |zq|^2) in absolute value: 2.02 K 2^-23; the f32 roundings of the centred values and the
// f64 arithmetic of the exact distance: 2^-20.  x 1.25.
static double ks_cfac(int dd)
{
    const double K = 3.0 * dd + 4.0;
    return 1.25 * (6.103515625e-05 + 2.0 * 1.52587890625e-05 + 2.02 * K * 1.1920928955078125e-07 + 9.5367431640625e-07);
}
static int ks_ks(int dd) { return (3 * dd + 4 + 15) / 16; }   // k-steps of 16 operand slots

// the shapes the screen takes: score space of at most 62 dimensions (8 operand-column groups), k inside the finishing kernel's
// buffers, and enough rows for the groups to give a bar the candidate list can hold the survivors of
static void ks_plan(int64_t n, int k, int64_t &ntiles, int &T, int &nslots)
{
    ntiles = (n + 31) / 32;
    const int64_t target = std::min<int64_t>(KS_MAXSLOTS, std::max<int64_t>(64, ((int64_t)5 * k + 1) / 2));
    T = (int)std::max<int64_t>(1, (ntiles + target - 1) / target);
    nslots = (int)((ntiles + T - 1) / T);
}
// the bar is the k-th smallest of G group minima, i.e. about the (-G ln(1 - k / G))-th smallest distance: the expected number of
// survivors per query (0: too few groups for this k).  (+ 15 % and the ties) they must fit the candidate list.
static double ks_survivors(int64_t n, int k)
{
    int64_t ntiles; int T, nslots;
    ks_plan(n, k, ntiles, T, nslots);
    const double G = 2.0 * (double)(n / (32 * (int64_t)T));
    if (G < 1.25 * k) return 0.0;
    return -G * log(1.0 - (double)k / G);
}
bool jch_knn_screen_shape_ok(int64_t n, int dd, int k)
{
    if (dd < 1 || dd > 62 || k < 1 || k > KNN_CAP - 256 || n >= ((int64_t)1 << 26)) return false;
    const double est = ks_survivors(n, k);
    return est > 0.0 && 1.15 * est + 32.0 <= 0.9 * KS_CCAP;
}
size_t jch_knn_screen_model_bytes(int64_t n, int dd)
{
    const size_t ntiles = (size_t)((n + 31) / 32);
    const size_t ldzr = ((size_t)dd + 7) & ~(size_t)7;   // (rows in whole 64-byte sectors)
    return ntiles * (size_t)ks_ks(dd) * 64 * 16 + (((size_t)n * ldzr * sizeof(double) + 255) & ~(size_t)255) + (((size_t)dd * sizeof(double) + 255) & ~(size_t)255) + 256 + sizeof(double) * 64 * KS_MB;
}
int32_t jch_knn_screen_build(jch_ctx *ctx, const double *dZt, int64_t ldzt, int64_t n, int dd, void *mem, knn_screen *out)
{
    knn_screen sc;
    sc.KS = ks_ks(dd);
    sc.ntiles = (n + 31) / 32;
    char *b = (char *)mem;
    sc.Zs = (uint4 *)b; b += (size_t)sc.ntiles * sc.KS * 64 * 16;
    sc.ldzr = (dd + 7) & ~7;
    sc.Zr = (double *)b; b += ((size_t)n * sc.ldzr * sizeof(double) + 255) & ~(size_t)255;
    sc.mu = (double *)b; b += ((size_t)dd * sizeof(double) + 255) & ~(size_t)255;
    sc.hdr = (unsigned *)b; b += 256;
    double *part = reinterpret_cast<double *>(b);                 // [dd][KS_MB] partial column sums
    // column means (two stages), the row-major copy (tiled transpose), the operand copy
    hipLaunchKernelGGL(k_ks_colsum, dim3(dd, KS_MB), dim3(256), 0, ctx->stream, dZt, ldzt, n, part);
    hipLaunchKernelGGL(k_ks_colmean, dim3(1), dim3(64), 0, ctx->stream, part, dd, n, sc.mu, sc.hdr);
    jch_lw_to_rowmajor(ctx, dZt, ldzt, n, dd, sc.Zr, sc.ldzr);
    const unsigned nb = (unsigned)((sc.ntiles * 64 + 255) / 256);
#define KS_PACK(KSv) case KSv: hipLaunchKernelGGL((k_ks_pack_rows<KSv>), dim3(nb), dim3(256), 0, ctx->stream, dZt, ldzt, n, dd, sc.mu, sc.Zs, sc.ntiles, sc.hdr); break
    switch (sc.KS) { KS_PACK(1); KS_PACK(2); KS_PACK(3); KS_PACK(4); KS_PACK(5); KS_PACK(6); KS_PACK(7); KS_PACK(8); KS_PACK(9); KS_PACK(10); KS_PACK(11); KS_PACK(12);
    default: return jch_fail(ctx, JCH_EINVAL, "internal: screened kNN: %d score dimensions", dd); }
#undef KS_PACK
    JCH_HIP(ctx, hipGetLastError());
    *out = sc;
    return JCH_OK;
}

template <int KS>
static void ks_launch_passes(jch_ctx *ctx, const ks_args &g)
{
    hipLaunchKernelGGL((k_ks_pack_queries<KS>), dim3((unsigned)((g.nqt * 64 + 255) / 256)), dim3(256), 0, ctx->stream, g);
    (void)jch_ev(ctx);
    constexpr int QWmax = KS <= 6 ? 2 : 1;                       // (registers: 4 KS per 32 queries for their operand)
    auto grid = [&](int qw, int nchunks) { return (unsigned)(((nchunks + 7) / 8) * (((g.nqt + qw - 1) / qw + 3) / 4) * 8); };   // (ks_item)
    if (g.qw_g == 2 && QWmax == 2) hipLaunchKernelGGL((k_knn_gmin<KS, QWmax>), dim3(grid(2, g.nchunks_g)), dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL((k_knn_gmin<KS, 1>), dim3(grid(1, g.nchunks_g)), dim3(256), 0, ctx->stream, g);
    const int G = 2 * g.nslots, nbq = (g.nqt * 32 + 3) / 4;
    if (G <= 256) hipLaunchKernelGGL((k_knn_bar<4>), dim3(nbq), dim3(256), 0, ctx->stream, g);
    else if (G <= 512) hipLaunchKernelGGL((k_knn_bar<8>), dim3(nbq), dim3(256), 0, ctx->stream, g);
    else if (G <= 1024) hipLaunchKernelGGL((k_knn_bar<16>), dim3(nbq), dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL((k_knn_bar<32>), dim3(nbq), dim3(256), 0, ctx->stream, g);
    if (g.qw == 2 && QWmax == 2) hipLaunchKernelGGL((k_knn_survive<KS, QWmax>), dim3(grid(2, g.nchunks)), dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL((k_knn_survive<KS, 1>), dim3(grid(1, g.nchunks)), dim3(256), 0, ctx->stream, g);
}

int32_t jch_launch_knn_screen(jch_ctx *ctx, const knn_args &a, const knn_screen &sc, int *flags)
{
    if (!jch_knn_screen_shape_ok(a.n, a.dd, a.k) || sc.KS != ks_ks(a.dd)) return jch_fail(ctx, JCH_EINVAL, "internal: screened kNN: shape outside its envelope");
    ks_args g;
    g.a = a; g.sc = sc;
    int64_t ntiles;
    ks_plan(a.n, a.k, ntiles, g.T, g.nslots);
    g.nqt = (a.m + 31) / 32;
    // The two passes split the work their own ways (measured at cfg5, one box: the group-minima pass 16.9 us with two query tiles per
    // wave item and one group slot per chunk, 21.8 with one tile; the survivor pass 31.2 us with one tile and two slots per chunk,
    // 36.4-40.7 with two tiles — its survivor bookkeeping wants registers, not operand reuse).
    // k_knn_gmin: ~6 wave items per SIMD of the chip
    g.qw_g = (g.nqt >= 2 && sc.KS <= 6) ? 2 : 1;
    g.gpw_g = (int)std::max<int64_t>(1, ((int64_t)g.nslots * ((g.nqt + g.qw_g - 1) / g.qw_g)) / ((int64_t)ctx->cus * 4 * 6));
    g.nchunks_g = (g.nslots + g.gpw_g - 1) / g.gpw_g;
    // k_knn_survive: at most 256 chunks (a query's slots, chunks x 32 bytes, are read in one trip by its finishing workgroup), and
    // at most 1 GB of them (m x chunks x KS_S ints)
    g.qw = 1;
    if (const char *e = getenv("JCH_KNN_SCREEN_QW")) g.qw = (atoi(e) == 2 && g.nqt >= 2 && sc.KS <= 6) ? 2 : 1;
    g.gpw = (int)std::max<int64_t>(1, ((int64_t)g.nslots * ((g.nqt + g.qw - 1) / g.qw)) / ((int64_t)ctx->cus * 4 * 6));
    g.gpw = std::max(g.gpw, (g.nslots + 255) / 256);
    if (const char *e = getenv("JCH_KNN_SCREEN_GPW")) g.gpw = std::max(1, atoi(e));
    g.gpw = (int)std::max<int64_t>(g.gpw, ((int64_t)a.m * g.nslots * KS_S * 4 + ((int64_t)1 << 30) - 1) >> 30);
    g.nchunks = (g.nslots + g.gpw - 1) / g.gpw;
    g.cfac = ks_cfac(a.dd);
    const size_t mpad = (size_t)g.nqt * 32;
    const size_t b_qs = (size_t)g.nqt * sc.KS * 64 * 16, b_nq = mpad * sizeof(double), b_gmin = (size_t)g.nqt * g.nslots * 64 * sizeof(float),
                 b_bar = mpad * sizeof(float), b_cnt = mpad * sizeof(int), b_cand = (size_t)a.m * KS_CCAP * sizeof(int),
                 b_scand = (size_t)a.m * g.nchunks * KS_S * sizeof(int);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    JCH_TRY(jch_reserve(ctx, ctx->gemm_b, up(b_qs) + up(b_nq) + up(b_gmin) + up(b_bar) + up(b_cnt) + up(b_cand) + up(b_scand) + 256));
    char *b = (char *)ctx->gemm_b.ptr;
    g.Qs = (uint4 *)b; b += up(b_qs);
    g.nq = (double *)b; b += up(b_nq);
    g.gmin = (float *)b; b += up(b_gmin);
    g.bar = (float *)b; b += up(b_bar);
    g.cnt = (int *)b; b += up(b_cnt);
    g.cand = (int *)b; b += up(b_cand);
    g.scand = (int *)b;
    g.flags = flags;
#define KS_PASSES(KSv) case KSv: ks_launch_passes<KSv>(ctx, g); break
    switch (sc.KS) { KS_PASSES(1); KS_PASSES(2); KS_PASSES(3); KS_PASSES(4); KS_PASSES(5); KS_PASSES(6); KS_PASSES(7); KS_PASSES(8); KS_PASSES(9);
    KS_PASSES(10); KS_PASSES(11); default: ks_launch_passes<12>(ctx, g); break; }
#undef KS_PASSES
    if (const char *e = getenv("JCH_KNN_SCREEN_DBG")) {
        g.a.dbg = atoi(e);
        if (g.a.dbg & 1) {   // survivors per query (host sync; measurement only)
            std::vector<int> hc(mpad), hs((size_t)a.m * g.nchunks * KS_S);
            JCH_HIP(ctx, hipMemcpyAsync(hc.data(), g.cnt, sizeof(int) * mpad, hipMemcpyDeviceToHost, ctx->stream));
            JCH_HIP(ctx, hipMemcpyAsync(hs.data(), g.scand, sizeof(int) * hs.size(), hipMemcpyDeviceToHost, ctx->stream));
            JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
            long long tot = 0, ovf = 0; int mx = 0, over256 = 0;
            for (int i = 0; i < a.m; ++i) {
                int c = hc[i];
                for (size_t e = 0; e < (size_t)g.nchunks * KS_S; ++e) c += hs[(size_t)i * g.nchunks * KS_S + e] >= 0 ? 1 : 0;
                tot += c; ovf += hc[i]; mx = std::max(mx, c); over256 += c > 256;
            }
            fprintf(stderr, "[jch] screened kNN: m=%d k=%d T=%d slots=%d gpw=%d chunks=%d survivors mean %.1f max %d, %d lists > 256, %lld through the overflow lists\n", a.m, a.k,
                    g.T, g.nslots, g.gpw, g.nchunks, (double)tot / a.m, mx, over256, ovf);
        }
    }
    hipLaunchKernelGGL(k_knn_finish_screen, dim3((unsigned)a.m), dim3(256), sizeof(double) * (size_t)a.dd, ctx->stream, g);
    JCH_HIP(ctx, hipGetLastError());
    // the flagged queries (the exception): the exact scan, for the groups of four queries with a flagged member
    knn_args ag = a;
    ag.only_flags = g.flags;
    return jch_launch_knn_scan(ctx, ag, ctx->lw_work);
}
