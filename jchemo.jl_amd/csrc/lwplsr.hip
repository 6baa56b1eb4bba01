// kNN-LWPLSR prediction path (BASELINE.json configs[4]):
//   K9  k_knn_scan + k_knn_finish   brute-force k nearest neighbours in a (whitened) score space + `wdist` weights
//                       replaces getknn (src/getknn.jl:29-57: NearestNeighbors.BruteTree + knn(sorted)) and the
//                       per-query weight loop of predict(::Lwplsr) (src/lwplsr.jl:152-159, src/wdist.jl:64-75,
//                       mad: src/utility.jl:679)
//   K8  k_locw_plskern  one workgroup per query: gather the k neighbour rows, weighted `plskern` on them and
//                       the 1-row predictions for the whole nlv range
//                       replaces locwlv (src/locwlv.jl:9-48: Threads.@threads over queries, plskern + predict per query)
// The reference runs m independent small fits on CPU threads; here a query's whole fit lives in one workgroup:
// its k x p block (0.8 MB at cfg5) is gathered once into an L2/MALL-resident scratch slab and swept once per LV.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <type_traits>
#include <new>
#include <vector>

#include "jch_internal.h"
#include "lv_device.h"
#include "rowsum_dev.h"
#include "lwplsr_dev.h"

typedef double v2f64 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------- row-major copy of a column-major matrix
__global__ __launch_bounds__(256) void k_to_rowmajor(const double *__restrict__ Xc, int64_t ldx, int64_t n, int p,
                                                     double *__restrict__ Xr, int ldr)
{
    __shared__ double xt[64 * 65];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j0 = blockIdx.y * 64;
    const int64_t nchunks = (n + 63) / 64;
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t i0 = c * 64;
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int col = wv + 4 * k, j = j0 + col;
            const int64_t i = i0 + lane;
            xt[lane * 65 + col] = (i < n && j < p) ? Xc[(size_t)i + (size_t)j * (size_t)ldx] : 0.0;
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int row = wv + 4 * k, j = j0 + lane;
            const int64_t i = i0 + row;
            if (i < n && j < ldr) Xr[(size_t)i * ldr + j] = xt[row * 65 + lane];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- K9: kNN + weights
#define KNN_QB 4       // queries per workgroup (each loaded training value serves KNN_QB queries; 8 measured slower)
// (KNN_CAP = 1024, lwplsr_dev.h: candidate buffer per query (LDS); k <= KNN_CAP - 256)
#define KNN_FCAP 2048  // candidates per query that k_knn_finish merges (segments x k)
#define KNN_RB 4        // 256-row chunks of the training scores in flight per trip
#define KNN_CB 8        // score columns loaded together

// (bitonic_sort_n, knn_finish_tail: lwplsr_dev.h — shared with lwplsr_screen.hip)
// (bitonic_sort_wave: lwplsr_dev.h)
// Compaction WITHOUT a sort (one wave, a query's candidate buffer of c0 <= KNN_CAP unordered entries): any bar tau' that leaves
// at least k entries <= tau' is a valid bar, so the wave sorts 64 SAMPLES of the keys across its lanes (21 shuffle passes),
// bisects for the smallest sample with >= k keys at or below it (6 counting steps: 16 compares + ballots per lane), and moves
// the survivors to the front (k .. k + ~30 of them).  ~700 instructions against ~13000 for the bitonic sort of 1024 entries,
// which was half of the scan's time.  Returns the number of survivors, or -1 if even the largest sample has fewer than k keys
// below it (the caller sorts then).  Keys are never NaN here (a NaN distance fails every test against the bar).
__device__ static int knn_wave_compact(double *key, int *idx, int c0, int k, double *tau_out)
{
    constexpr int E = KNN_CAP / 64;
    const int lane = threadIdx.x & 63;
    double kr[E];
#pragma unroll
    for (int j = 0; j < E; ++j) { const int e = lane + 64 * j; kr[j] = e < c0 ? key[e] : __builtin_inf(); }
    double sv = kr[0];                                   // (c0 >= 64: entries 0 .. 63 exist)
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const double pv = __shfl_xor(sv, stride, 64);
            const bool keepmin = ((lane & stride) == 0) == ((lane & size) == 0);
            sv = keepmin ? (pv < sv ? pv : sv) : (pv > sv ? pv : sv);
        }
    auto count = [&](double pv) {
        int n = 0;
#pragma unroll
        for (int j = 0; j < E; ++j) n += __popcll(__ballot((lane + 64 * j < c0) && kr[j] <= pv));
        return n;
    };
    int lo = 0, hi = 63;
    if (count(__shfl(sv, 63, 64)) < k) return -1;
    while (lo < hi) {                                    // (wave-uniform)
        const int mid = (lo + hi) >> 1;
        if (count(__shfl(sv, mid, 64)) >= k) hi = mid; else lo = mid + 1;
    }
    const double pivot = __shfl(sv, hi, 64);
    int ir[E];
#pragma unroll
    for (int j = 0; j < E; ++j) { const int e = lane + 64 * j; ir[j] = e < c0 ? idx[e] : 0x7fffffff; }
    wavesync();
    int base = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const bool keep = (lane + 64 * j < c0) && kr[j] <= pivot;
        const unsigned long long m = __ballot(keep);
        if (keep) { const int pos = base + __popcll(m & below); key[pos] = kr[j]; idx[pos] = ir[j]; }
        base += __popcll(m);
    }
    wavesync();
    *tau_out = pivot;
    return base;
}
__device__ __forceinline__ int knn_pow2_at_least(int v) { int c = 64; while (c < v) c <<= 1; return c; }

// (struct knn_args: lwplsr_dev.h)

// NT threads per workgroup, QB queries per workgroup (every loaded training value serves QB queries).  <256, 4>: two workgroups
// per CU (default); <512, 8> (round 3, JCH_KNN_WIDE=1): ONE workgroup of eight waves per CU — the same waves in flight, half the
// L2 traffic (the kernel reads the whole score matrix once per query group: 16 MB x 250 groups at cfg5) — measured SLOWER, see the
// launcher.
template <int NT, int QB>
__global__ __launch_bounds__(NT) void k_knn_scan(knn_args g)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *bkey = lds;                                            // [QB][CAP]
    int *bidx = reinterpret_cast<int *>(bkey + QB * KNN_CAP);  // [QB][CAP]
    double *zq = reinterpret_cast<double *>(bidx + QB * KNN_CAP);  // [QB][dd]
    double *tau = zq + QB * g.dd;                              // [QB]
    int *cnt = reinterpret_cast<int *>(tau + QB);              // [QB]
    static_assert(QB == NT / 64, "one wave per query in the compactions");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // segment = block mod nseg (with 8 segments every XCD would only read ONE segment of the score matrix, workgroups being dealt
    // round-robin over the XCDs — measured: no gain, the scan is not bound by where its 4 GB of L2 / Infinity Cache reads come
    // from but by its waves' dependent steps)
    const int seg = blockIdx.x % g.nseg;
    const int q0 = (blockIdx.x / g.nseg) * QB;
    const int nq = min(QB, g.m - q0);
    if (g.only_flags) {   // behind the screened search: only the query groups with a flagged member are scanned (block-uniform)
        int any = 0;
        for (int qq = 0; qq < nq; ++qq) any |= g.only_flags[q0 + qq];
        if (!any) return;
    }
    for (int e = tid; e < QB * g.dd; e += NT) {
        const int qq = e / g.dd, c = e - qq * g.dd;
        zq[e] = qq < nq ? g.Zq[(size_t)(q0 + qq) + (size_t)c * (size_t)g.ldzq] : 0.0;
    }
    if (tid < QB) { tau[tid] = (g.dbg & 1) ? -__builtin_inf() : __builtin_inf(); cnt[tid] = 0; }
    __syncthreads();
    const int k = g.k;
    // KNN_RB chunks of NT training rows per trip: all their loads go out together (the kernel is bound by the latency of the
    // score matrix in L2 / MALL with 4 waves per CU), then the chunks are offered to the candidate buffers one after the
    // other, exactly as if they had been read one at a time (round 2: 4.1 -> 1.3 ms per 1000 queries at cfg5)
    // this block's segment of the training rows (multiples of NT rows; the last one takes the rest)
    const int64_t seg_rows = ((g.n + g.nseg - 1) / g.nseg + NT - 1) / NT * NT;
    const int64_t row_lo = min(g.n, (int64_t)seg * seg_rows), row_hi = min(g.n, row_lo + seg_rows);
    for (int64_t base = row_lo; base < row_hi; base += NT * KNN_RB) {
        double d2[KNN_RB][QB];
#pragma unroll
        for (int r = 0; r < KNN_RB; ++r)
#pragma unroll
            for (int qq = 0; qq < QB; ++qq) d2[r][qq] = 0.0;
        unsigned bo[KNN_RB];
#pragma unroll
        for (int r = 0; r < KNN_RB; ++r) {
            const int64_t i = base + NT * r + tid;
            bo[r] = (unsigned)(i < row_hi ? i : g.n - 1) * 8u;
        }
        // KNN_CB score columns x KNN_RB row chunks = 32 loads per thread go out before the first is used (written as a plain
        // loop the compiler waited for each column's 4 loads before it issued the next: 20 round trips per trip instead of 3)
        for (int c0 = 0; c0 < g.dd; c0 += KNN_CB) {
            double x[KNN_CB][KNN_RB];
#pragma unroll
            for (int cc = 0; cc < KNN_CB; ++cc) {
                // a uniform column base + a 32-bit byte offset of the row (n < 2^29, checked by the launcher): no 64-bit vector
                // arithmetic per load
                const char *cp = reinterpret_cast<const char *>(g.Zt + (size_t)min(c0 + cc, g.dd - 1) * (size_t)g.ldzt);
#pragma unroll
                for (int r = 0; r < KNN_RB; ++r) x[cc][r] = *reinterpret_cast<const double *>(cp + bo[r]);
            }
#pragma unroll
            for (int cc = 0; cc < KNN_CB; ++cc)
                if (c0 + cc < g.dd) {                        // block-uniform
#pragma unroll
                    for (int r = 0; r < KNN_RB; ++r)
#pragma unroll
                        for (int qq = 0; qq < QB; ++qq) { const double e = x[cc][r] - zq[qq * g.dd + c0 + cc]; d2[r][qq] += e * e; }
                }
        }
#pragma unroll
        for (int r = 0; r < KNN_RB; ++r) {
            const int64_t i = base + NT * r + tid;
            if (base + NT * r >= row_hi) break;            // block-uniform
            double tq[QB];                               // one LDS round trip for the four bars instead of one per test
#pragma unroll
            for (int qq = 0; qq < QB; ++qq) tq[qq] = tau[qq];
            if (i < row_hi) {
#pragma unroll
                for (int qq = 0; qq < QB; ++qq)
                    if (qq < nq && (d2[r][qq] < tq[qq] || (d2[r][qq] == tq[qq] && cnt[qq] < k))) {
                        const int pos = atomicAdd(&cnt[qq], 1);
                        bkey[qq * KNN_CAP + pos] = d2[r][qq];
                        bidx[qq * KNN_CAP + pos] = (int)i;
                    }
            }
            __syncthreads();
            int cq[QB];
#pragma unroll
            for (int qq = 0; qq < QB; ++qq) cq[qq] = cnt[qq];
            // every wave must have taken its snapshot before any wave's next chunk bumps the counters: the compaction
            // below contains barriers, so the decision has to be the same in all four waves
            __syncthreads();
            bool need = false;
#pragma unroll
            for (int qq = 0; qq < QB; ++qq) need = need || (qq < nq && cq[qq] > KNN_CAP - NT);
            if (need) {   // compact: keep the k best, raise the bar (block-uniform decision); wave w sorts query w's buffer
                const int c0 = cnt[wv];                   // (stable: the next appends come after the barrier below)
                if (wv < nq && c0 > KNN_CAP - NT) {
                    double *key = bkey + wv * KNN_CAP;
                    int *idx = bidx + wv * KNN_CAP;
                    double bar = 0.0;
                    const int c1 = knn_wave_compact(key, idx, c0, k, &bar);
                    if (c1 >= k && c1 <= KNN_CAP - 2 * NT) { if (lane == 0) { cnt[wv] = c1; tau[wv] = bar; } }
                    else {                                // (no sample bar, or ties left too many entries: the exact k best)
                        const int c2 = c1 < 0 ? c0 : c1;
                        for (int e = c2 + lane; e < KNN_CAP; e += 64) { key[e] = __builtin_inf(); idx[e] = 0x7fffffff; }
                        bitonic_sort_wave(key, idx, KNN_CAP);
                        if (lane == 0) { cnt[wv] = k; tau[wv] = key[k - 1]; }
                    }
                }
                __syncthreads();
            }
        }
    }
    // the segment's k best of every query, in (distance, index) order: wave w takes query w
    __syncthreads();
    const bool fused = g.only_flags != nullptr;   // behind the screened search (one segment): the workgroup finishes its queries itself
    if (wv < nq) {
        int c0 = cnt[wv];
        double *key = bkey + wv * KNN_CAP;
        int *idx = bidx + wv * KNN_CAP;
        if (c0 > 256 && c0 > k) {                         // shrink to k .. k + ~30 entries first: the sort below is over those only
            double bar;
            const int c1 = knn_wave_compact(key, idx, c0, k, &bar);
            if (c1 >= k) c0 = c1;
        }
        const int cap = knn_pow2_at_least(c0);
        for (int e = c0 + lane; e < cap; e += 64) { key[e] = __builtin_inf(); idx[e] = 0x7fffffff; }
        bitonic_sort_wave(key, idx, cap);
        if (fused) {
            for (int e = c0 + lane; e < k; e += 64) { key[e] = __builtin_inf(); idx[e] = 0x7fffffff; }
        } else {
            double *ok = g.ckey + ((size_t)(q0 + wv) * g.nseg + seg) * k;
            int *oi = g.cidx + ((size_t)(q0 + wv) * g.nseg + seg) * k;
            for (int e = lane; e < k; e += 64) { ok[e] = e < c0 ? key[e] : __builtin_inf(); oi[e] = e < c0 ? idx[e] : 0x7fffffff; }
        }
    }
    if (fused) {   // (block-uniform) K9b on the lists where they lie: the members of a scanned group are all finished (the unflagged ones get the values they have)
        double *scr = reinterpret_cast<double *>(cnt + QB + 2);     // [KNN_CAP] scratch | sred [8] | smed [2] | snn [4] (the launcher sizes the LDS for it)
        double *sred = scr + KNN_CAP, *smed = sred + 8;
        int *snn = reinterpret_cast<int *>(smed + 2);
        const int kk = (int)min<int64_t>(k, g.n);
        for (int qq = 0; qq < nq; ++qq) {
            __syncthreads();
            knn_finish_tail(g, q0 + qq, kk, scr, bkey + qq * KNN_CAP, bidx + qq * KNN_CAP, sred, smed, snn);
        }
    }
}

// K9b: one workgroup per query merges the segments' candidates, orders the k nearest and turns the distances into weights
__global__ __launch_bounds__(256) void k_knn_finish(knn_args g)
{
    __shared__ double key[KNN_FCAP];
    __shared__ int idx[KNN_FCAP];
    __shared__ double okey[KNN_CAP];
    __shared__ int oidx[KNN_CAP];
    __shared__ double sred[8];
    __shared__ double smed[2];
    __shared__ int snn[4];
    const int tid = threadIdx.x;
    const int qi = blockIdx.x, k = g.k;
    const int ncand = g.nseg * k;
    const double *ck = g.ckey + (size_t)qi * ncand;
    const int *ci = g.cidx + (size_t)qi * ncand;
    const int kk = (int)min<int64_t>(k, g.n);
    for (int e = tid; e < ncand; e += 256) { key[e] = ck[e]; idx[e] = ci[e]; }
    for (int e = tid; e < kk; e += 256) { okey[e] = __builtin_inf(); oidx[e] = 0x7fffffff; }
    __syncthreads();
    // MERGE BY RANK instead of a sort (round 3; was a bitonic sort of all candidates, 55 barriers): every segment's list is
    // already in (distance, index) order, so an entry's place in the merged order is its place in its own list plus, for every
    // other list, the number of entries there that come before it — a binary search each.  Sentinels (+inf, 0x7fffffff) compare
    // equal and may land on the same place: the output is pre-filled with them.
    {
        int s_ = 0, i_ = tid;                                  // (list, position) of entry e, advanced without a division
        while (i_ >= k) { i_ -= k; ++s_; }
        for (int e = tid; e < ncand; e += 256) {
            const double a = key[e];
            const int ia = idx[e];
            int rank = i_;
            for (int s2 = 0; s2 < g.nseg; ++s2) {
                if (s2 == s_) continue;
                const double *kl = key + s2 * k;
                const int *il = idx + s2 * k;
                int lo = 0, hi = k;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const double bkey_ = kl[mid];
                    const bool before = (bkey_ < a) || (bkey_ == a && il[mid] < ia);
                    if (before) lo = mid + 1; else hi = mid;
                }
                rank += lo;
            }
            if (rank < kk) { okey[rank] = a; oidx[rank] = ia; }
            i_ += 256;
            while (i_ >= k) { i_ -= k; ++s_; }
        }
    }
    __syncthreads();
    knn_finish_tail(g, qi, kk, key, okey, oidx, sred, smed, snn);
}

// ---------------------------------------------------------------- K8: batched local weighted plskern (q <= 16)
// (struct locw_args: lwplsr_dev.h)

// block sum of NV values held one per thread-array slot: wave shuffles, then the 4 wave partials through LDS.
// out[v] valid in every thread after the call.  scratch: >= 4 * NV doubles.
template <int NV>
__device__ __forceinline__ void locw_block_sums(double (&v)[NV], double *scratch)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = jch_wave_sum(v[i]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[wv * NV + i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = ((scratch[i] + scratch[NV + i]) + scratch[2 * NV + i]) + scratch[3 * NV + i];
}

// Q = number of responses padded to {1, 2, 4, 8, 16}; the actual q = g.q <= Q (pad responses are all-zero columns).
template <int KC, int Q>
__global__ __launch_bounds__(256) void k_locw_plskern(locw_args g)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int p = g.p, ldr = g.ldr, k = g.k, q = g.q;
    constexpr int lda = Q + 2;
    double *dl = lds;                 // [k]      normalised weights
    double *yc = dl + k;              // [k][Q]   centred/scaled Y rows of the neighbours
    double *mu = yc + (size_t)k * Q;  // [ldr]    local means
    double *sg = mu + ldr;            // [ldr]    local scales
    double *Kv = sg + ldr;            // [Q][ldr] kernel matrix X'DY, one row per response
    double *wv_ = Kv + (size_t)Q * ldr;  // [ldr] w
    double *rv = wv_ + ldr;           // [ldr] r
    double *xq = rv + ldr;            // [ldr] centred/scaled query
    double *zred = xq + ldr;          // [4][KC*128]
    double *sc = zred + 4 * KC * 128; // [128] scratch scalars; dots w.P_l at sc[64 + l]
    double *ys = sc + 128;            // [4 Q]: ymean, ysd, c, running prediction
    double *vl = ys + 4 * Q;          // [16]   dominant right singular vector
    double *G0 = vl + 16;             // 5 x [Q][lda] Gram / eigen-solver work + csl [2 (Q + 2) + 8]
    double *A0 = G0 + Q * lda, *A1 = A0 + Q * lda, *V0 = A1 + Q * lda, *V1 = V0 + Q * lda, *csl = V1 + Q * lda;
    double *red = csl + 2 * (Q + 2) + 8;   // [4 * max(Q (Q + 1) / 2, Q + 1)]
    int *idx = reinterpret_cast<int *>(red + 4 * (Q * (Q + 1) / 2 + Q + 1));  // [k]
    double *Xg = g.scratch + (size_t)blockIdx.x * g.slab;
    double *Pm = Xg + (size_t)k * ldr;
    double *Rm = Pm + (size_t)g.nlv_hi * ldr;
    const int le = g.nlv_hi - g.nlv_lo + 1;
    const int nlvloc = min(min(k, p), g.nlv_hi);
    double *ymean = ys, *ysd = ys + Q, *prun = ys + 3 * Q;

    for (int qi = blockIdx.x; qi < g.m; qi += gridDim.x) {
        __syncthreads();
        // ---- weights (mweight), neighbour ids, Y rows
        double s = 0.0;
        for (int e = tid; e < k; e += 256) { idx[e] = g.ind[(size_t)qi * k + e]; s += g.w[(size_t)qi * k + e]; }
        const double sw = jch_block_sum<256>(s, sc);
        double sy[Q];
#pragma unroll
        for (int y = 0; y < Q; ++y) sy[y] = 0.0;
        double ymin = __builtin_inf(), ymax = -__builtin_inf();
        for (int e = tid; e < k; e += 256) {
            const double d = g.w[(size_t)qi * k + e] / sw;
            dl[e] = d;
#pragma unroll
            for (int y = 0; y < Q; ++y) {
                const double yv = y < q ? g.Y[(size_t)idx[e] + (size_t)y * (size_t)g.ldy] : 0.0;
                yc[(size_t)e * Q + y] = yv;
                sy[y] += d * yv;
                if (y == 0) { ymin = fmin(ymin, yv); ymax = fmax(ymax, yv); }
            }
        }
        locw_block_sums<Q>(sy, red);
        if (tid < Q) ymean[tid] = sy[tid];
        // constant-y shortcut, univariate y only (src/locwlv.jl:25-28)
        __syncthreads();
        for (int o = 32; o > 0; o >>= 1) { ymin = fmin(ymin, __shfl_xor(ymin, o, 64)); ymax = fmax(ymax, __shfl_xor(ymax, o, 64)); }
        if (lane == 0) { sc[8 + wv] = ymin; sc[12 + wv] = ymax; }
        __syncthreads();
        const double gmin = fmin(fmin(sc[8], sc[9]), fmin(sc[10], sc[11])), gmax = fmax(fmax(sc[12], sc[13]), fmax(sc[14], sc[15]));
        __syncthreads();
        if (q == 1 && gmin == gmax) {
            for (int a = tid; a < le; a += 256) g.pred[(size_t)qi * le + a] = gmin;
            continue;
        }
        // ---- local weighted means / stds of X: wave per neighbour row, SR rows per trip with all loads first (random rows of
        // the row-major copy: HBM latency), column sums in registers like zp in the sweep, 4-wave combine through LDS
        constexpr int SR = KC <= 4 ? 4 : (KC == 8 ? 2 : 1);
        auto load_rows = [&](v2f64 (&x)[SR][KC], int e0) {
#pragma unroll
            for (int rr = 0; rr < SR; ++rr) {
                const int e = min(e0 + 4 * rr, k - 1);          // rows past k: re-read the last one (weight zero below)
                const v2f64 *rp = reinterpret_cast<const v2f64 *>(g.Xrm + (size_t)idx[e] * ldr) + lane;
#pragma unroll
                for (int c = 0; c < KC; ++c) x[rr][c] = (2 * lane + 128 * c < ldr) ? rp[64 * c] : v2f64{0.0, 0.0};
            }
        };
        auto combine_cols = [&](const v2f64 (&acc)[KC], double *out, bool root) {   // out[j] = sum over the 4 waves (sqrt if root)
            __syncthreads();
#pragma unroll
            for (int c = 0; c < KC; ++c) *reinterpret_cast<v2f64 *>(zred + wv * (KC * 128) + 2 * lane + 128 * c) = acc[c];
            __syncthreads();
            for (int j = tid; j < ldr; j += 256) {
                const double v = ((zred[j] + zred[KC * 128 + j]) + zred[2 * KC * 128 + j]) + zred[3 * KC * 128 + j];
                out[j] = root ? (j < p ? sqrt(v) : 1.0) : v;
            }
            __syncthreads();
        };
        {
            v2f64 acc[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) acc[c] = v2f64{0.0, 0.0};
            for (int e0 = wv; e0 < k; e0 += 4 * SR) {
                v2f64 x[SR][KC];
                load_rows(x, e0);
#pragma unroll
                for (int rr = 0; rr < SR; ++rr) {
                    const double d = e0 + 4 * rr < k ? dl[e0 + 4 * rr] : 0.0;
#pragma unroll
                    for (int c = 0; c < KC; ++c) { acc[c].x += d * x[rr][c].x; acc[c].y += d * x[rr][c].y; }
                }
            }
            combine_cols(acc, mu, false);
            if (g.scal) {
                v2f64 mf[KC];
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    mf[c] = (2 * lane + 128 * c < ldr) ? *reinterpret_cast<const v2f64 *>(mu + 2 * lane + 128 * c) : v2f64{0.0, 0.0};
                    acc[c] = v2f64{0.0, 0.0};
                }
                for (int e0 = wv; e0 < k; e0 += 4 * SR) {
                    v2f64 x[SR][KC];
                    load_rows(x, e0);
#pragma unroll
                    for (int rr = 0; rr < SR; ++rr) {
                        const double d = e0 + 4 * rr < k ? dl[e0 + 4 * rr] : 0.0;
#pragma unroll
                        for (int c = 0; c < KC; ++c) {
                            const double zx = x[rr][c].x - mf[c].x, zy = x[rr][c].y - mf[c].y;
                            acc[c].x += d * zx * zx; acc[c].y += d * zy * zy;
                        }
                    }
                }
                combine_cols(acc, sg, true);
            } else {
                for (int j = tid; j < ldr; j += 256) sg[j] = 1.0;
            }
        }
        {
            double vv[Q];
#pragma unroll
            for (int y = 0; y < Q; ++y) vv[y] = 0.0;
            if (g.scal)
                for (int e = tid; e < k; e += 256)
#pragma unroll
                    for (int y = 0; y < Q; ++y) { const double z = yc[(size_t)e * Q + y] - ymean[y]; vv[y] += dl[e] * z * z; }
            locw_block_sums<Q>(vv, red);
            if (tid < Q) ysd[tid] = (g.scal && tid < q) ? sqrt(vv[tid]) : 1.0;
        }
        __syncthreads();
        for (int e = tid; e < k; e += 256)
#pragma unroll
            for (int y = 0; y < Q; ++y) {
                const double z = yc[(size_t)e * Q + y] - ymean[y];
                yc[(size_t)e * Q + y] = y < q ? (g.scal ? z / ysd[y] : z) : 0.0;
            }
        // ---- gather + centre/scale into the slab; centred query; K = X' D Y (for Q * KC <= 16 accumulated in the same pass)
        __syncthreads();
        for (int j = tid; j < ldr; j += 256)
            xq[j] = j < p ? (g.scal ? (g.Xq[(size_t)qi + (size_t)j * (size_t)g.ldxq] - mu[j]) / sg[j]
                                    : g.Xq[(size_t)qi + (size_t)j * (size_t)g.ldxq] - mu[j]) : 0.0;
        {
            constexpr bool FUSEK = Q * KC <= 16;
            v2f64 mf[KC], sf[KC], kacc[FUSEK ? Q : 1][KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const bool in = 2 * lane + 128 * c < ldr;
                mf[c] = in ? *reinterpret_cast<const v2f64 *>(mu + 2 * lane + 128 * c) : v2f64{0.0, 0.0};
                sf[c] = in ? *reinterpret_cast<const v2f64 *>(sg + 2 * lane + 128 * c) : v2f64{1.0, 1.0};
#pragma unroll
                for (int y = 0; y < (FUSEK ? Q : 1); ++y) kacc[y][c] = v2f64{0.0, 0.0};
            }
            for (int e0 = wv; e0 < k; e0 += 4 * SR) {
                v2f64 x[SR][KC];
                load_rows(x, e0);
#pragma unroll
                for (int rr = 0; rr < SR; ++rr) {
                    const int e = e0 + 4 * rr;
                    if (e < k) {                                 // wave-uniform
                        double *dst = Xg + (size_t)e * ldr;
#pragma unroll
                        for (int c = 0; c < KC; ++c) {
                            const int col = 2 * lane + 128 * c;
                            v2f64 z;
                            z.x = col < p ? (g.scal ? (x[rr][c].x - mf[c].x) / sf[c].x : x[rr][c].x - mf[c].x) : 0.0;
                            z.y = col + 1 < p ? (g.scal ? (x[rr][c].y - mf[c].y) / sf[c].y : x[rr][c].y - mf[c].y) : 0.0;
                            if (col < ldr) *reinterpret_cast<v2f64 *>(dst + col) = z;
                            if constexpr (FUSEK) {
#pragma unroll
                                for (int y = 0; y < Q; ++y) {
                                    const double sy_ = dl[e] * yc[(size_t)e * Q + y];
                                    kacc[y][c].x += sy_ * z.x; kacc[y][c].y += sy_ * z.y;
                                }
                            }
                        }
                    }
                }
            }
            if constexpr (FUSEK) {
#pragma unroll
                for (int y = 0; y < Q; ++y) combine_cols(kacc[y], Kv + (size_t)y * ldr, false);
            } else {
                __syncthreads();
                for (int j = tid; j < ldr; j += 256) {
                    double s2[Q];
#pragma unroll
                    for (int y = 0; y < Q; ++y) s2[y] = 0.0;
                    if (j < p)
                        for (int e = 0; e < k; ++e) {
                            const double xv = dl[e] * Xg[(size_t)e * ldr + j];
#pragma unroll
                            for (int y = 0; y < Q; ++y) s2[y] += xv * yc[(size_t)e * Q + y];
                        }
#pragma unroll
                    for (int y = 0; y < Q; ++y) Kv[(size_t)y * ldr + j] = s2[y];
                }
            }
        }
        if (tid < Q) prun[tid] = ymean[tid];   // nlv = 0: the intercept alone (src/plskern.jl:207-217 with B = 0)
        __syncthreads();
        if (tid < q && g.nlv_lo == 0) g.pred[((size_t)qi * le) * q + tid] = prun[tid];
        // ---- LV loop (src/plskern.jl:149-175)
        for (int a = 0; a < nlvloc; ++a) {
            // w = dominant left singular vector of K: q == 1 the normalised column (:150-152), else K v / ||K v|| with v the
            // dominant eigenvector of K'K (Gram + repeated squaring / Jacobi in wave 0, same solver and sign rule as the
            // global fit: lv_device.h)
            if constexpr (Q > 1) {
                constexpr int NE = Q * (Q + 1) / 2;
                if constexpr (Q <= 8) {   // Gram entries as per-thread partial sums (36 registers at Q = 8), block-reduced
                    double ge[NE];
#pragma unroll
                    for (int e = 0; e < NE; ++e) ge[e] = 0.0;
                    for (int j = tid; j < p; j += 256) {
                        double kr[Q];
#pragma unroll
                        for (int y = 0; y < Q; ++y) kr[y] = Kv[(size_t)y * ldr + j];
                        int e = 0;
#pragma unroll
                        for (int y1 = 0; y1 < Q; ++y1)
#pragma unroll
                            for (int y2 = y1; y2 < Q; ++y2) ge[e++] += kr[y1] * kr[y2];
                    }
                    locw_block_sums<NE>(ge, red);
                    for (int e = tid; e < 5 * Q * lda; e += 256) G0[e] = 0.0;
                    __syncthreads();
                    if (tid == 0) {
                        int e = 0;
                        for (int y1 = 0; y1 < Q; ++y1)
                            for (int y2 = y1; y2 < Q; ++y2) { G0[y1 * lda + y2] = ge[e]; G0[y2 * lda + y1] = ge[e]; ++e; }
                    }
                } else {                  // Q = 16: 136 entries x 2 column halves = 272 work items over the 256 threads, rows of K from LDS
                    for (int e = tid; e < 5 * Q * lda; e += 256) G0[e] = 0.0;
                    for (int item = tid; item < 2 * NE; item += 256) {
                        int e = item >> 1, y1 = 0;
                        while (e >= Q - y1) { e -= Q - y1; ++y1; }
                        const double *ka = Kv + (size_t)y1 * ldr, *kb = Kv + (size_t)(y1 + e) * ldr;
                        double g0 = 0.0, g1 = 0.0;
                        int j = item & 1;
                        for (; j + 2 < p; j += 4) { g0 += ka[j] * kb[j]; g1 += ka[j + 2] * kb[j + 2]; }
                        for (; j < p; j += 2) g0 += ka[j] * kb[j];
                        red[item] = g0 + g1;
                    }
                    __syncthreads();      // G0 zeroed, partial sums published
                    for (int ent = tid; ent < NE; ent += 256) {
                        int e = ent, y1 = 0;
                        while (e >= Q - y1) { e -= Q - y1; ++y1; }
                        const double v = red[2 * ent] + red[2 * ent + 1];
                        G0[y1 * lda + y1 + e] = v; G0[(y1 + e) * lda + y1] = v;
                    }
                }
                __syncthreads();
                if (wv == 0) {
                    if (!dominant_by_squaring<Q>(q, lda, G0, A0, A1, vl, nullptr)) {
                        for (int e = lane; e < Q * lda; e += 64) A0[e] = G0[e];
                        wavesync();
                        jacobi_wave(q, lda, A0, A1, V0, V1, csl, vl, nullptr);
                    }
                }
                __syncthreads();
            }
            double s2 = 0.0;
            for (int j = tid; j < ldr; j += 256) {
                double wj = 0.0;
                if (j < p) {
                    if (Q == 1) wj = Kv[j];
                    else
#pragma unroll
                        for (int y = 0; y < Q; ++y) wj += Kv[(size_t)y * ldr + j] * vl[y];
                }
                wv_[j] = wj;
                s2 += wj * wj;
            }
            const double nrm = sqrt(jch_block_sum<256>(s2, sc));
            for (int j = tid; j < ldr; j += 256) wv_[j] = wv_[j] / nrm;   // (own entries only)
            __syncthreads();
            for (int l = wv; l < a; l += 4) {   // dots w . P_l
                double s3 = 0.0;
                for (int j = lane; j < p; j += 64) s3 += wv_[j] * Pm[(size_t)l * ldr + j];
                s3 = jch_wave_sum(s3);
                if (lane == 0) sc[64 + l] = s3;
            }
            __syncthreads();
            for (int j = tid; j < ldr; j += 256) {
                double rj = wv_[j];
                for (int l = 0; l < a; ++l) rj -= sc[64 + l] * Rm[(size_t)l * ldr + j];
                rv[j] = j < p ? rj : 0.0;
            }
            __syncthreads();
            // fused sweep over the gathered rows (same structure as k_sweep, sweep.hip)
            v2f64 rf[KC], zp[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int col = 2 * lane + 128 * c;
                rf[c] = col < ldr ? *reinterpret_cast<const v2f64 *>(rv + col) : v2f64{0.0, 0.0};
                zp[c] = v2f64{0.0, 0.0};
            }
            double tt = 0.0;
            // LR rows of the wave per trip, all loads first (the slab sits in L2: with 4-8 waves per CU the loop is bound by
            // that latency, one trip = one round trip), row sums by the transposing permlane reduction of the K4 sweep
            constexpr int LR = KC <= 4 ? 4 : (KC == 8 ? 2 : 1);   // rows per wave and trip (register budget: LR x KC pairs)
            for (int e0 = wv; e0 < k; e0 += 4 * LR) {
                v2f64 x[LR][KC];
                double s4[LR];
#pragma unroll
                for (int rr = 0; rr < LR; ++rr) {
                    const int e = min(e0 + 4 * rr, k - 1);      // rows past k: re-read the last one, weight zero
                    const v2f64 *rp = reinterpret_cast<const v2f64 *>(Xg + (size_t)e * ldr) + lane;
#pragma unroll
                    for (int c = 0; c < KC; ++c) x[rr][c] = (2 * lane + 128 * c < ldr) ? rp[64 * c] : v2f64{0.0, 0.0};
                }
#pragma unroll
                for (int rr = 0; rr < LR; ++rr) {
                    double a4 = 0.0;
#pragma unroll
                    for (int c = 0; c < KC; ++c) a4 += x[rr][c].x * rf[c].x + x[rr][c].y * rf[c].y;
                    s4[rr] = a4;
                }
                double tv[LR];
                if constexpr (LR == 4) {
                    const double hsum = jch_rowsums<4>(s4, lane);
#pragma unroll
                    for (int rr = 0; rr < LR; ++rr) tv[rr] = jch_readlane(hsum, jch_rowsum_lane<4>(rr));
                } else {
#pragma unroll
                    for (int rr = 0; rr < LR; ++rr) tv[rr] = jch_wave_sum(s4[rr]);
                }
#pragma unroll
                for (int rr = 0; rr < LR; ++rr) {
                    const int e = e0 + 4 * rr;
                    const double t = tv[rr];
                    const double dt = e < k ? dl[e] * t : 0.0;
                    tt += dt * t;
#pragma unroll
                    for (int c = 0; c < KC; ++c) { zp[c].x += dt * x[rr][c].x; zp[c].y += dt * x[rr][c].y; }
                }
            }
#pragma unroll
            for (int c = 0; c < KC; ++c) *reinterpret_cast<v2f64 *>(zred + wv * (KC * 128) + 2 * lane + 128 * c) = zp[c];
            if (lane == 0) sc[wv] = tt;
            __syncthreads();
            const double ttot = ((sc[0] + sc[1]) + sc[2]) + sc[3];
            // c = K'r / tt (one entry per response) ; tq = xq . r
            double cs[Q + 1];
#pragma unroll
            for (int y = 0; y <= Q; ++y) cs[y] = 0.0;
            for (int j = tid; j < p; j += 256) {
                const double rj = rv[j];
#pragma unroll
                for (int y = 0; y < Q; ++y) cs[y] += Kv[(size_t)y * ldr + j] * rj;
                cs[Q] += xq[j] * rj;
            }
            locw_block_sums<Q + 1>(cs, red);
            const double tq = cs[Q];
            for (int j = tid; j < ldr; j += 256) {
                const double z = ((zred[j] + zred[KC * 128 + j]) + zred[2 * KC * 128 + j]) + zred[3 * KC * 128 + j];
#pragma unroll
                for (int y = 0; y < Q; ++y) Kv[(size_t)y * ldr + j] -= z * (cs[y] / ttot);
                Pm[(size_t)a * ldr + j] = z / ttot;
                Rm[(size_t)a * ldr + j] = rv[j];
            }
            const int kk = a + 1;
            if (tid < Q) {
                const double pr = prun[tid] + tq * (cs[tid] / ttot) * ysd[tid];
                prun[tid] = pr;
                if (tid < q && kk >= g.nlv_lo && kk <= g.nlv_hi) g.pred[((size_t)qi * le + (kk - g.nlv_lo)) * q + tid] = pr;
            }
            __syncthreads();
        }
        // requested nlv beyond what the local model has: predict clamps to the model's nlv (src/plskern.jl:228-229)
        for (int e = tid; e < (g.nlv_hi - nlvloc) * q; e += 256) {
            const int kk = nlvloc + 1 + e / q, y = e % q;
            if (kk >= g.nlv_lo) g.pred[((size_t)qi * le + (kk - g.nlv_lo)) * q + y] = prun[y];
        }
    }
}

template <int KC, int Q>
static int32_t launch_locw_q(jch_ctx *ctx, locw_args &g)
{
    const size_t lds = sizeof(double) * ((1 + (size_t)Q) * g.k + (5 + (size_t)Q) * g.ldr + 4 * KC * 128 + 128 + 4 * Q + 16 + 5 * Q * (Q + 2) +
                                         2 * (Q + 2) + 8 + 4 * (Q * (Q + 1) / 2 + Q + 1)) + sizeof(int) * (size_t)g.k + 64;
    static jch_per_device_once attr;
    if (!attr.done(ctx->device)) {
        JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_locw_plskern<KC, Q>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr.mark(ctx->device);
    }
    if (lds > 150 * 1024) {   // the slab bookkeeping of this kernel does not fit: the neighbour-space kernel has no such limit in p
        if (jch_locw_kspace_feasible(g)) return jch_launch_locw_kspace(ctx, g);
        return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_predict: k / p / q too large for the batched local-PLS kernels (k = %d, p = %d, q = %d)", g.k, g.p, g.q);
    }
    const char *e_bpc = getenv("JCH_LOCW_BPC");   // (measurement knob) blocks per CU of the local-fit kernel
    int nb = std::min(g.m, ctx->cus * ((e_bpc && atoi(e_bpc) > 0) ? atoi(e_bpc) : 2));
    g.slab = ((size_t)g.k * g.ldr + 2 * (size_t)g.nlv_hi * g.ldr + 31) & ~(size_t)31;
    if (ctx->xcopy.host) ctx->xcopy_valid = false;   // (the staging buffer of a host-array fit is re-used for the slabs)
    JCH_TRY(jch_reserve(ctx, ctx->xstage, sizeof(double) * g.slab * nb));
    g.scratch = (double *)ctx->xstage.ptr;
    hipLaunchKernelGGL((k_locw_plskern<KC, Q>), dim3(nb), dim3(256), lds, ctx->stream, g);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

template <int KC>
static int32_t launch_locw(jch_ctx *ctx, locw_args &g)
{
    if (g.q <= 1) return launch_locw_q<KC, 1>(ctx, g);
    if (g.q <= 2) return launch_locw_q<KC, 2>(ctx, g);
    if (g.q <= 4) return launch_locw_q<KC, 4>(ctx, g);
    if (g.q <= 8) return launch_locw_q<KC, 8>(ctx, g);
    return launch_locw_q<KC, 16>(ctx, g);
}

// the shape fits one of the batched local-fit kernels (the conditions of launch_locw_q / jch_locw_kspace_feasible)
static bool locw_batched_fits(const locw_args &g)
{
    if (g.q > 16 || g.nlv_hi > 48 || g.ldr > JCH_SWEEP_MAXP) return false;
    if (jch_locw_kspace_feasible(g)) return true;
    const int KC = g.ldr <= 128 ? 1 : g.ldr <= 256 ? 2 : g.ldr <= 512 ? 4 : g.ldr <= 1024 ? 8 : 16;
    const size_t Q = g.q <= 1 ? 1 : g.q <= 2 ? 2 : g.q <= 4 ? 4 : g.q <= 8 ? 8 : 16;
    const size_t lds = sizeof(double) * ((1 + Q) * g.k + (5 + Q) * g.ldr + 4 * KC * 128 + 128 + 4 * Q + 16 + 5 * Q * (Q + 2) +
                                         2 * (Q + 2) + 8 + 4 * (Q * (Q + 1) / 2 + Q + 1)) + sizeof(int) * (size_t)g.k + 64;
    return lds <= 150 * 1024;
}

// ---------------------------------------------------------------- C ABI
// The model-constant part of a prediction: what `lwplsr(X, Y; ...)` holds (src/lwplsr.jl:1-12, 114-131) in the form the
// kernels want it — the row-major copy of Xtrain for the 4 KB-contiguous neighbour gathers, device copies of Ytrain and of
// the (whitened) training scores.  Built once by jch_lwplsr_prepare and reused by every jch_lwplsr_predict_prepared call;
// the one-shot jch_lwplsr_predict builds the same thing in the ctx workspace on every call.
struct jch_lwplsr_model {
    int device = 0;
    int64_t n = 0, p = 0, q = 0, dd = 0;
    int ldr = 0;
    double *Xrm = nullptr;   // [n][ldr]
    double *Y = nullptr;     // n x q, column-major, ld n
    double *Zt = nullptr;    // n x dd, column-major, ld n
    // optional: the map that takes a query block to the neighbour-search space, as a chain of affine maps (the reference's
    // transform(fm, X) followed by the whitening of getknn: two stages) kept on the device — jch_lwplsr_add_query_map
    struct qmap { int p_in = 0, k_out = 0, kpad = 0; double *dB = nullptr; };   // dB: [p_in][kpad] folded B, then kpad folded biases
    std::vector<qmap> qmaps;
    // the operand-ordered f32 copy of the scores for the screened kNN (lwplsr_screen.hip); absent when the score space is too wide
    bool has_screen = false;
    mutable bool screen_off = false;   // set when a call had a quarter of its queries redone by the exact scan
    knn_screen screen;
    void *screen_mem = nullptr;
};

void jch_lw_to_rowmajor(jch_ctx *ctx, const double *dX, int64_t ldxd, int64_t n, int p, double *Xrm, int ldr)
{
    const int ptiles = (ldr + 63) / 64;
    const int64_t nchunks = (n + 63) / 64;
    int nbx = (int)std::max<int64_t>(1, std::min<int64_t>(nchunks, (ctx->cus * 4 + ptiles - 1) / ptiles));
    hipLaunchKernelGGL(k_to_rowmajor, dim3(nbx, ptiles), dim3(256), 0, ctx->stream, dX, ldxd, n, p, Xrm, ldr);
}

// LDS bytes of the scan for this shape (its envelope: <= 150 KB); wide_out: the JCH_KNN_WIDE=1 measurement variant applies
size_t jch_knn_scan_lds(int k, int dd, int m, bool *wide_out)
{
    // JCH_KNN_WIDE=1 (measurement knob, round 3): eight queries per 512-thread workgroup — half the L2 traffic, the same waves per
    // CU.  Measured at cfg5: 1.62 ms with 3 row segments, 1.05 with 2, against 0.87 for four queries per 256-thread workgroup:
    // the scan is not bound by the L2 bytes but by its dependent steps (threshold tests, LDS appends, barriers, sorts), which
    // eight waves share one candidate bookkeeping for.  Default: off.
    const char *e_w = getenv("JCH_KNN_WIDE");
    const bool wide = k <= KNN_CAP - 512 && e_w && atoi(e_w) == 1 && m > 4;
    if (wide_out) *wide_out = wide;
    const int qb = wide ? 8 : KNN_QB;
    return (sizeof(double) + sizeof(int)) * qb * KNN_CAP + sizeof(double) * (qb * (size_t)dd + qb) + sizeof(int) * qb + 64;
}

// The exact scan + finish (K9 / K9b) for a.m queries; a.only_flags: null, or device flags — only the groups of KNN_QB queries with a
// flagged member are done (the queries the screened search could not settle).  cbuf: the buffer the segments' candidates go to.
int32_t jch_launch_knn_scan(jch_ctx *ctx, knn_args a, jch_buf &cbuf)
{
    const int64_t n = a.n;
    const int m = a.m, k = a.k;
    bool wide = false;
    const size_t lds = jch_knn_scan_lds(k, a.dd, m, &wide);
    if (a.only_flags) wide = false;
    const int nt = wide ? 512 : 256;
    {
        // row segments: as many as keep every segment at >= 4 trips and the merged candidate lists inside one sort (nseg * k <= KNN_CAP)
        // (measured at cfg5, 1000 queries: 1 segment 1.38 ms, 2: 1.00, 3: 0.88, 5: 1.21 — every (query group, segment) block pays
        // its own compaction sorts)
        // (round 3, with the sort-free compactions: 2 segments 0.60 ms, 3: 0.51, 4: 0.59, 5: 0.57, 6: 0.56, 8: 0.59)
        int nseg = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(3, KNN_FCAP / k), n / (4 * nt * KNN_RB)));
        if (const char *e = getenv("JCH_KNN_SEGMENTS")) nseg = std::max(1, std::min(atoi(e), KNN_FCAP / k));
        // behind the screened search: ONE segment and no separate finishing kernel — two launches whose workgroups find nothing
        // flagged and leave cost 9.8 us per call at cfg5, one launch of a third as many workgroups 3
        if (a.only_flags) nseg = 1;
        a.nseg = nseg;
        JCH_TRY(jch_reserve(ctx, cbuf, (sizeof(double) + sizeof(int)) * (size_t)m * nseg * k + 256));
        a.ckey = (double *)cbuf.ptr; a.cidx = (int *)(a.ckey + (size_t)m * nseg * k);
        static jch_per_device_once attr;
        if (!attr.done(ctx->device)) {
            JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_knn_scan<256, KNN_QB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_knn_scan<512, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr.mark(ctx->device);
        }
        // (round 4, measured and removed: eight queries per 256-thread workgroup with 512-entry candidate buffers — half the score
        // traffic per query — took 1.1 ms for the scan at cfg5 with half as many workgroups, i.e. the same time per workgroup-query:
        // the scan is NOT bound by its L2 / Infinity Cache traffic; with a dot-product screen |z|^2 + |zq|^2 - 2 z.zq in front of the
        // exact distance — half the arithmetic per pair — 1.78 ms: the ~1000 survivors per query and segment each pay a divergent
        // 20-load recomputation.  Results were identical in both.  What bounds the scan is the dependent chain of a wave's trip.)
        if (wide) hipLaunchKernelGGL((k_knn_scan<512, 8>), dim3((unsigned)((m + 7) / 8) * nseg), dim3(512), lds, ctx->stream, a);
        else hipLaunchKernelGGL((k_knn_scan<256, KNN_QB>), dim3((unsigned)((m + KNN_QB - 1) / KNN_QB) * nseg), dim3(256), lds + (a.only_flags ? sizeof(double) * (KNN_CAP + 16) : 0), ctx->stream, a);
        if (!a.only_flags) hipLaunchKernelGGL(k_knn_finish, dim3((unsigned)m), dim3(256), 0, ctx->stream, a);
    }
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

// kNN + weights + batched local fits on device-resident pieces; results to the host.  Xrm / dY / dZt: the model; dZq, dXq: the queries.
static int32_t lw_run(jch_ctx *ctx, const double *Xrm, int ldr, int64_t n, int64_t p, const double *dY, int64_t q, int64_t ldyd,
                      const double *dZt, int64_t ldztd, const double *dZq, int64_t ldzqd, int64_t dd, const double *dXq, int64_t m,
                      int64_t ldxqd, int32_t k, double h, double tol, int32_t scal, int32_t nlv_lo, int32_t nlv_hi, double *pred,
                      int32_t *ind_out, double *dist_out, double *w_out, hipEvent_t ev0, const knn_screen *scr, bool *screen_off)
{
    const int le = nlv_hi - nlv_lo + 1;
    JCH_TRY(jch_reserve(ctx, ctx->gemm_out, sizeof(double) * ((size_t)m * k * 2 + (size_t)m * le * q) + sizeof(int) * (size_t)m * k + 256));
    double *ddist = (double *)ctx->gemm_out.ptr, *dw = ddist + (size_t)m * k, *dpred = dw + (size_t)m * k;
    int *dind = (int *)(dpred + (size_t)m * le * q);
    hipEvent_t ev1 = jch_ev(ctx), ev2 = nullptr, ev3 = nullptr;   // profiling: (copy) | kNN + weights | local fits
    bool generic_fits = false;
    int *kflags = nullptr;          // k-space kernel: per-query pivot flags (device)
    locw_args kargs{};
    int *sflags = nullptr;          // screened kNN: per-query "redone by the exact selection" flags (device; read back with the predictions)
    {
        knn_args a;
        a.Zt = dZt; a.ldzt = ldztd; a.n = n; a.Zq = dZq; a.ldzq = ldzqd; a.m = (int)m; a.dd = (int)dd; a.k = k;
        a.h = h; a.cri = 4.0; a.tol = tol; a.ind = dind; a.dist = ddist; a.w = dw;
        { const char *e = getenv("JCH_KNN_DBG"); a.dbg = e ? atoi(e) : 0; }
        // JCH_KNN_WIDE=1 (measurement knob, round 3): eight queries per 512-thread workgroup — half the L2 traffic, the same waves per
        // CU.  Measured at cfg5: 1.62 ms with 3 row segments, 1.05 with 2, against 0.87 for four queries per 256-thread workgroup:
        // the scan is not bound by the L2 bytes but by its dependent steps (threshold tests, LDS appends, barriers, sorts), which
        // eight waves share one candidate bookkeeping for.  Default: off.
        const size_t lds = jch_knn_scan_lds(k, (int)dd, (int)m, nullptr);
        // outside the scan's envelope (k beyond the candidate buffers, a search space too wide for its LDS, 2^29 rows): the generic
        // selection, one workgroup per query (lwplsr_generic.hip); JCH_KNN_GENERIC=1 forces it (tests)
        const char *e_g = getenv("JCH_KNN_GENERIC");
        // the screened kNN (lwplsr_screen.hip: all pairs in f32 on the matrix cores, exact distances for the survivors only) when the
        // shape is inside its envelope; JCH_KNN_SCREEN=0 selects the exact scan below (A/B runs, tests)
        const char *e_s = getenv("JCH_KNN_SCREEN");
        const bool screen = !(e_g && atoi(e_g) == 1) && !(e_s && atoi(e_s) == 0) && !a.dbg && !(screen_off && *screen_off) && jch_knn_screen_shape_ok(n, (int)dd, k);
        if (screen) {
            knn_screen local;
            if (!scr) {   // one-shot call: the model-constant operand copy is rebuilt in the ctx workspace
                JCH_TRY(jch_reserve(ctx, ctx->lw_screen, jch_knn_screen_model_bytes(n, (int)dd)));
                JCH_TRY(jch_knn_screen_build(ctx, dZt, ldztd, n, (int)dd, ctx->lw_screen.ptr, &local));
                scr = &local;
            }
            // (one reservation for both flag arrays of a call: the neighbour-space kernel's pivot flags [0, m), the screen's [m, 2 m))
            JCH_TRY(jch_reserve(ctx, ctx->lw_flags, sizeof(int) * 2 * (size_t)m + 256));
            sflags = (int *)ctx->lw_flags.ptr + (size_t)m;
            JCH_TRY(jch_launch_knn_screen(ctx, a, *scr, sflags));
        } else
        if (k > KNN_CAP - 256 || lds > 150 * 1024 || n >= ((int64_t)1 << 29) || (e_g && atoi(e_g) == 1)) {
            JCH_TRY(jch_launch_knn_generic(ctx, a));
        } else {
        JCH_TRY(jch_launch_knn_scan(ctx, a, ctx->gemm_b));
        }
    }
    ev2 = jch_ev(ctx);
    // neighbours, distances and weights are final here: their copies to the host (3.2 MB at cfg5, into pages of the caller's
    // fresh arrays that have never been touched) run on a second stream BESIDE the local fits instead of after them
    // JCH_LW_LISTS_STAGED=1 (round 4, measured, NOT the default): the neighbour lists go to pinned staging through copies queued on
    // the main stream IN FRONT of the local fits, and the host moves them on to the caller's arrays while the local fits run.  Why
    // it was tried: under rocprofv3 two of the three list copies of the default arrangement (second stream, beside the local fits)
    // show up AFTER the local-fit launch.  Without the tracer they do not (JCH_LW_HOST_DBG=1 host stamps: the lists are on the
    // host 250 us after the kNN launches, the main stream's wait ends at 1036 us; staged: 1100-1115 us, the copies delay the local
    // fits by their 75 us): 1.20-1.25 ms per call against 1.17-1.18 on the same box.  Kept for stacks where the copies do starve.
    const size_t list_bytes = (ind_out ? sizeof(int) : 0) * (size_t)m * k + ((dist_out ? 1 : 0) + (w_out ? 1 : 0)) * sizeof(double) * (size_t)m * k;
    const size_t pred_bytes_ = sizeof(double) * (size_t)m * le * q + sizeof(int) * 2 * (size_t)m + 64;
    const char *e_ls = getenv("JCH_LW_LISTS_STAGED");
    const bool lists_staged = e_ls && atoi(e_ls) == 1 && list_bytes > 0 && list_bytes + pred_bytes_ <= ((size_t)64 << 20);
    char *hl_d = nullptr, *hl_w = nullptr, *hl_i = nullptr;   // pinned staging of dist / w / ind
    if (lists_staged) {
        JCH_TRY(jch_reserve_host(ctx, list_bytes + pred_bytes_ + 256));
        char *hb = (char *)ctx->hstage + ((pred_bytes_ + 63) & ~(size_t)63);
        if (dist_out) { hl_d = hb; hb += sizeof(double) * (size_t)m * k; }
        if (w_out) { hl_w = hb; hb += sizeof(double) * (size_t)m * k; }
        if (ind_out) { hl_i = hb; }
        if (dist_out && w_out) JCH_HIP(ctx, hipMemcpyAsync(hl_d, ddist, 2 * sizeof(double) * (size_t)m * k, hipMemcpyDeviceToHost, ctx->stream));   // (adjacent on both sides)
        else {
            if (dist_out) JCH_HIP(ctx, hipMemcpyAsync(hl_d, ddist, sizeof(double) * (size_t)m * k, hipMemcpyDeviceToHost, ctx->stream));
            if (w_out) JCH_HIP(ctx, hipMemcpyAsync(hl_w, dw, sizeof(double) * (size_t)m * k, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (ind_out) JCH_HIP(ctx, hipMemcpyAsync(hl_i, dind, sizeof(int) * (size_t)m * k, hipMemcpyDeviceToHost, ctx->stream));
        if (!ctx->aux_event && hipEventCreateWithFlags(&ctx->aux_event, hipEventDisableTiming) != hipSuccess) { ctx->aux_event = nullptr; return jch_fail(ctx, JCH_EHIP, "jch_lwplsr_predict: event creation failed"); }
        JCH_HIP(ctx, hipEventRecord(ctx->aux_event, ctx->stream));
    }
    // the lists are on the host as soon as their event is: on to the caller's arrays (beside the batched local fits; BEFORE per-query
    // fits, which use the pinned staging themselves)
    bool lists_drained = !lists_staged;
    const bool host_dbg = getenv("JCH_LW_HOST_DBG") != nullptr;
    const auto t_host0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_host0).count(); };
    auto drain_lists = [&]() -> int32_t {
        if (lists_drained) return JCH_OK;
        lists_drained = true;
        const double t0 = since();
        JCH_HIP(ctx, hipEventSynchronize(ctx->aux_event));
        const double t1 = since();
        if (dist_out) memcpy(dist_out, hl_d, sizeof(double) * (size_t)m * k);
        if (w_out) memcpy(w_out, hl_w, sizeof(double) * (size_t)m * k);
        if (ind_out) memcpy(ind_out, hl_i, sizeof(int) * (size_t)m * k);
        if (host_dbg) fprintf(stderr, "[jch] lists staged: wait for the event %.0f .. %.0f us after the kNN launches, host copy until %.0f\n", t0, t1, since());
        return JCH_OK;
    };
    bool side_copies = !lists_staged && (ind_out || dist_out || w_out) && !getenv("JCH_LW_SIDE_COPY_OFF");
    if (side_copies) {
        if (!ctx->aux_stream && hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess) { ctx->aux_stream = nullptr; side_copies = false; }
        if (side_copies && !ctx->aux_event && hipEventCreateWithFlags(&ctx->aux_event, hipEventDisableTiming) != hipSuccess) { ctx->aux_event = nullptr; side_copies = false; }
        if (side_copies) {
            JCH_HIP(ctx, hipEventRecord(ctx->aux_event, ctx->stream));
            JCH_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->aux_event, 0));
        }
    }
    {
        locw_args g;
        g.Xrm = Xrm; g.ldr = ldr; g.p = (int)p; g.Y = dY; g.ldy = ldyd; g.q = (int)q; g.Xq = dXq; g.ldxq = ldxqd; g.m = (int)m;
        g.ind = dind; g.w = dw; g.k = k; g.scal = scal; g.nlv_lo = nlv_lo; g.nlv_hi = nlv_hi; g.pred = dpred;
        g.scratch = nullptr; g.slab = 0; g.flags = nullptr;
        { const char *e = getenv("JCH_LOCW_DBG"); g.dbg = e ? atoi(e) : 0; }
        // neighbour-space kernel (lwplsr_kspace.hip: the gathered rows are read ONCE, the fit runs on their Gram matrix held in
        // registers) when the shape fits it; the p-space kernel (one sweep of the slab per LV) otherwise
        const char *e_lg = getenv("JCH_LOCW_GENERIC");
        if (!locw_batched_fits(g) || (e_lg && atoi(e_lg) == 1)) {   // outside the batched kernels' envelope: one fit per query (lwplsr_generic.hip)
            const hipError_t pe = hipMemsetAsync(dpred, 0, sizeof(double) * (size_t)m * le * q, ctx->stream);
            if (pe != hipSuccess) return jch_fail(ctx, JCH_EHIP, "jch_lwplsr_predict: %s", hipGetErrorString(pe));
            JCH_TRY(drain_lists());
            JCH_TRY(jch_lw_generic_fits(ctx, g, n));
            generic_fits = true;
        } else
        if (jch_locw_kspace_supported(g)) {
            // pivot check of the neighbour-space kernel: queries far from their neighbours in p-space are flagged and refitted below
            JCH_TRY(jch_reserve(ctx, ctx->lw_flags, sizeof(int) * 2 * (size_t)m + 256));
            g.flags = (int *)ctx->lw_flags.ptr;
            JCH_TRY(jch_launch_locw_kspace(ctx, g));
            kflags = g.flags; kargs = g;
        } else
        if (ldr <= 128) JCH_TRY(launch_locw<1>(ctx, g));
        else if (ldr <= 256) JCH_TRY(launch_locw<2>(ctx, g));
        else if (ldr <= 512) JCH_TRY(launch_locw<4>(ctx, g));
        else if (ldr <= 1024) JCH_TRY(launch_locw<8>(ctx, g));
        else JCH_TRY(launch_locw<16>(ctx, g));
    }
    JCH_HIP(ctx, hipGetLastError());
    ev3 = generic_fits ? nullptr : jch_ev(ctx);
    // (the local fits are in the queue: whatever the host does from here on runs beside them)
    // The small results — predictions, the two flag arrays — go to PINNED staging through copies queued right behind the local fits,
    // before the host waits for anything (round 4; three copies into pageable memory issued after the wait for the neighbour lists
    // left the device idle for 65 us and cost 30 + 18 + 34 us of turn-around: 0.15 of a 1.17 ms call at cfg5)
    const size_t pred_bytes = sizeof(double) * (size_t)m * le * q, flag_bytes = (kflags || sflags) ? sizeof(int) * 2 * (size_t)m : 0;
    const bool staged = lists_staged || pred_bytes + flag_bytes <= ((size_t)64 << 20);
    char *hs = nullptr;
    if (staged) {
        if (lists_drained) JCH_TRY(jch_reserve_host(ctx, pred_bytes + flag_bytes + 64));   // (lists still in the staging: reserved above, they lie behind this part)
        hs = (char *)ctx->hstage;
        if (flag_bytes) JCH_HIP(ctx, hipMemcpyAsync(hs, ctx->lw_flags.ptr, flag_bytes, hipMemcpyDeviceToHost, ctx->stream));   // [pivot flags m][screen flags m]
        JCH_HIP(ctx, hipMemcpyAsync(hs + flag_bytes, dpred, pred_bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    JCH_TRY(drain_lists());
    hipStream_t cs = side_copies ? ctx->aux_stream : ctx->stream;
    hipError_t ce = hipSuccess;
    if (lists_staged) { /* done */ } else
    if (ind_out && ce == hipSuccess) ce = hipMemcpyAsync(ind_out, dind, sizeof(int) * (size_t)m * k, hipMemcpyDeviceToHost, cs);
    if (!lists_staged && dist_out && ce == hipSuccess) ce = hipMemcpyAsync(dist_out, ddist, sizeof(double) * (size_t)m * k, hipMemcpyDeviceToHost, cs);
    if (!lists_staged && w_out && ce == hipSuccess) ce = hipMemcpyAsync(w_out, dw, sizeof(double) * (size_t)m * k, hipMemcpyDeviceToHost, cs);
    if (side_copies) { const hipError_t se = hipStreamSynchronize(cs); if (ce == hipSuccess) ce = se; }   // (before any return: the buffers are the caller's)
    if (ce != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); return jch_fail(ctx, JCH_EHIP, "jch_lwplsr_predict: copy of the neighbour lists failed: %s", hipGetErrorString(ce)); }
    std::vector<int> hf, hsf;
    if (staged) {
        const double ts0 = since();
        JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (host_dbg) fprintf(stderr, "[jch] stream wait %.0f .. %.0f us\n", ts0, since());
        memcpy(pred, hs + flag_bytes, pred_bytes);
        const int *hfl = reinterpret_cast<const int *>(hs);
        if (kflags) hf.assign(hfl, hfl + m);
        if (sflags) hsf.assign(hfl + m, hfl + 2 * m);
    } else {
        if (kflags) { hf.resize((size_t)m); JCH_HIP(ctx, hipMemcpyAsync(hf.data(), kflags, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream)); }
        if (sflags) { hsf.resize((size_t)m); JCH_HIP(ctx, hipMemcpyAsync(hsf.data(), sflags, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream)); }
        JCH_HIP(ctx, hipMemcpyAsync(pred, dpred, pred_bytes, hipMemcpyDeviceToHost, ctx->stream));
        JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (sflags) {
        long long redone = 0;
        for (int i = 0; i < (int)m; ++i) redone += hsf[(size_t)i] ? 1 : 0;
        ctx->knn_screened += (long long)m;
        ctx->knn_screen_redone += redone;
        // a training set whose geometry defeats the screen's error bound (lwplsr_screen.hip): the model stops screening
        if (screen_off && m >= 8 && redone * 4 > (long long)m) *screen_off = true;
    }
    if (kflags) {   // flagged queries (the exception): refitted by the per-query path, their predictions fetched again
        std::vector<int> only;
        for (int i = 0; i < (int)m; ++i) if (hf[(size_t)i]) only.push_back(i);
        if (!only.empty()) {
            ctx->locw_refits += (long long)only.size();
            JCH_TRY(jch_lw_generic_fits(ctx, kargs, n, only.data(), (int)only.size()));
            JCH_HIP(ctx, hipMemcpyAsync(pred, dpred, sizeof(double) * (size_t)m * le * q, hipMemcpyDeviceToHost, ctx->stream));
            JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ev3 = nullptr;   // (the refits recycled the event pool: no stage times for this call)
        }
    }
    if (ctx->profiling && ev0 && ev1 && ev2 && ev3) {
        // jch_profile of a prediction call: fit_ms = device time of the three stages, prologue_ms = row-major copy + kNN +
        // weights, sweep_ms = the batched local fits (ONE launch), sweep_bytes = the gathered neighbour rows m k ldr 8
        // (algorithmic bytes per query = k p 8), smallstate_ms = what precedes the kNN scan: the row-major copy (one-shot call) or the handle's query map (prepared model, Zq = NULL)
        float a = 0.f, b = 0.f, c = 0.f;
        (void)hipEventElapsedTime(&a, ev0, ev1); (void)hipEventElapsedTime(&b, ev1, ev2); (void)hipEventElapsedTime(&c, ev2, ev3);
        jch_profile &pr = ctx->prof;
        pr.fit_ms = a + b + c; pr.prologue_ms = a + b; pr.sweep_ms = c; pr.smallstate_ms = a;
        pr.sweep_launches = 1; pr.nlv = nlv_hi; pr.sweep_bytes = (double)m * k * ldr * 8.0;
    }
    return JCH_OK;
}

static int32_t lw_check(jch_ctx *ctx, const char *who, int64_t n, int64_t p, int64_t q, int64_t dd, int64_t m, int32_t &k, int32_t nlv_lo, int32_t nlv_hi)
{
    if (n < 1 || p < 1 || m < 1 || dd < 1 || k < 1 || q < 1 || nlv_lo < 0 || nlv_hi < nlv_lo) return jch_fail(ctx, JCH_EINVAL, "%s: bad arguments", who);
    if (n >= ((int64_t)1 << 31) - 1 || p > (1 << 20) || m >= ((int64_t)1 << 31) - 1) return jch_fail(ctx, JCH_EINVAL, "%s: n, p or m too large", who);
    if (k > n) k = (int32_t)n;                                    // src/getknn.jl:33
    // (no envelope limits: shapes outside the batched kernels' — k > 768; local fits with p > 2048, q > 16, nlv > 48 or beyond
    // their LDS budget — run the generic paths of lwplsr_generic.hip)
    return JCH_OK;
}

extern "C" int32_t jch_lwplsr_predict(jch_ctx *ctx, int32_t loc, const double *Xtrain, int64_t n, int64_t p, int64_t ldx,
                                      const double *Ytrain, int64_t q, int64_t ldy, const double *Ztrain, int64_t ldzt,
                                      const double *Zq, int64_t ldzq, int64_t dd, const double *Xq, int64_t m, int64_t ldxq,
                                      int32_t k, double h, double tol, int32_t scal, int32_t nlv_lo, int32_t nlv_hi,
                                      double *pred, int32_t *ind_out, double *dist_out, double *w_out)
{
    if (!ctx) return JCH_EINVAL;
    if (!Xtrain || !Ytrain || !Ztrain || !Zq || !Xq || !pred || ldx < n || ldy < n || ldzt < n || ldzq < m || ldxq < m)
        return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_predict: bad arguments");
    JCH_TRY(lw_check(ctx, "jch_lwplsr_predict", n, p, q, dd, m, k, nlv_lo, nlv_hi));
    if (loc != JCH_LOC_HOST && loc != JCH_LOC_DEVICE) return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_predict: bad loc");
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    const int ldr = ((int)p + 1) & ~1;
    // ---- stage host inputs
    const double *dX = Xtrain, *dY = Ytrain, *dZt = Ztrain, *dZq = Zq, *dXq = Xq;
    int64_t ldxd = ldx, ldyd = ldy, ldztd = ldzt, ldzqd = ldzq, ldxqd = ldxq;
    if (loc == JCH_LOC_HOST) {
        const size_t need = sizeof(double) * ((size_t)n * p + (size_t)n * q + (size_t)n * dd + (size_t)m * dd + (size_t)m * p);
        JCH_TRY(jch_reserve(ctx, ctx->xq, need));
        double *b = (double *)ctx->xq.ptr;
        auto up = [&](const double *src, int64_t rows, int64_t cols, int64_t ld, const double *&dst, int64_t &ldd) -> int32_t {
            if (ld == rows) JCH_HIP(ctx, hipMemcpyAsync(b, src, sizeof(double) * (size_t)rows * cols, hipMemcpyHostToDevice, ctx->stream));
            else JCH_HIP(ctx, hipMemcpy2DAsync(b, sizeof(double) * rows, src, sizeof(double) * ld, sizeof(double) * rows, cols, hipMemcpyHostToDevice, ctx->stream));
            dst = b; ldd = rows; b += (size_t)rows * cols;
            return JCH_OK;
        };
        JCH_TRY(up(Xtrain, n, p, ldx, dX, ldxd)); JCH_TRY(up(Ytrain, n, q, ldy, dY, ldyd)); JCH_TRY(up(Ztrain, n, dd, ldzt, dZt, ldztd));
        JCH_TRY(up(Zq, m, dd, ldzq, dZq, ldzqd)); JCH_TRY(up(Xq, m, p, ldxq, dXq, ldxqd));
    }
    // ---- the model-constant piece, in the ctx workspace (rebuilt on every call: use jch_lwplsr_prepare to keep it)
    // (a buffer of its own: the per-query fits of the generic path use the fit workspace, ctx->xr included)
    JCH_TRY(jch_reserve(ctx, ctx->lw_xrm, sizeof(double) * (size_t)n * ldr));
    double *Xrm = (double *)ctx->lw_xrm.ptr;
    ctx->ev_used = 0;
    ctx->prof = jch_profile{};
    hipEvent_t ev0 = jch_ev(ctx);
    jch_lw_to_rowmajor(ctx, dX, ldxd, n, (int)p, Xrm, ldr);
    return lw_run(ctx, Xrm, ldr, n, p, dY, q, ldyd, dZt, ldztd, dZq, ldzqd, dd, dXq, m, ldxqd, k, h, tol, scal, nlv_lo, nlv_hi, pred,
                  ind_out, dist_out, w_out, ev0, nullptr, nullptr);
}

extern "C" int32_t jch_lwplsr_prepare(jch_ctx *ctx, int32_t loc, const double *Xtrain, int64_t n, int64_t p, int64_t ldx,
                                      const double *Ytrain, int64_t q, int64_t ldy, const double *Ztrain, int64_t ldzt, int64_t dd,
                                      jch_lwplsr_model **model_out)
{
    if (!ctx) return JCH_EINVAL;
    if (!model_out) return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_prepare: model_out is NULL");
    *model_out = nullptr;
    if (!Xtrain || !Ytrain || !Ztrain || n < 1 || p < 1 || q < 1 || dd < 1 || ldx < n || ldy < n || ldzt < n || p > (1 << 20))
        return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_prepare: bad arguments");
    if (loc != JCH_LOC_HOST && loc != JCH_LOC_DEVICE) return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_prepare: bad loc");
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    jch_lwplsr_model *mo = new (std::nothrow) jch_lwplsr_model();
    if (!mo) return jch_fail(ctx, JCH_ENOMEM, "jch_lwplsr_prepare: host allocation failed");
    mo->device = ctx->device; mo->n = n; mo->p = p; mo->q = q; mo->dd = dd; mo->ldr = ((int)p + 1) & ~1;
    auto fail = [&](int32_t st) { (void)hipFree(mo->Xrm); (void)hipFree(mo->Y); (void)hipFree(mo->Zt); (void)hipFree(mo->screen_mem); delete mo; return st; };
    if (hipMalloc((void **)&mo->Xrm, sizeof(double) * (size_t)n * mo->ldr) != hipSuccess || hipMalloc((void **)&mo->Y, sizeof(double) * (size_t)n * q) != hipSuccess ||
        hipMalloc((void **)&mo->Zt, sizeof(double) * (size_t)n * dd) != hipSuccess)
        return fail(jch_fail(ctx, JCH_ENOMEM, "jch_lwplsr_prepare: device allocation failed (%lld x %lld training rows)", (long long)n, (long long)p));
    const hipMemcpyKind kind = loc == JCH_LOC_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    auto copy2d = [&](double *dst, const double *src, int64_t cols, int64_t ld) {
        return hipMemcpy2DAsync(dst, sizeof(double) * n, src, sizeof(double) * ld, sizeof(double) * n, cols, kind, ctx->stream);
    };
    if (copy2d(mo->Y, Ytrain, q, ldy) != hipSuccess || copy2d(mo->Zt, Ztrain, dd, ldzt) != hipSuccess)
        return fail(jch_fail(ctx, JCH_EHIP, "jch_lwplsr_prepare: copy of Y / scores failed"));
    const double *dX = Xtrain;
    int64_t ldxd = ldx;
    if (loc == JCH_LOC_HOST) {   // stage the column-major X once, transpose on the device
        int32_t st = jch_reserve(ctx, ctx->xstage, sizeof(double) * (size_t)n * p);
        if (st != JCH_OK) return fail(st);
        if (hipMemcpy2DAsync(ctx->xstage.ptr, sizeof(double) * n, Xtrain, sizeof(double) * ldx, sizeof(double) * n, p, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            return fail(jch_fail(ctx, JCH_EHIP, "jch_lwplsr_prepare: copy of X failed"));
        dX = (const double *)ctx->xstage.ptr; ldxd = n;
    }
    jch_lw_to_rowmajor(ctx, dX, ldxd, n, (int)p, mo->Xrm, mo->ldr);
    if (dd <= 62 && n < ((int64_t)1 << 26)) {   // (whether a call is screened also depends on its k: jch_knn_screen_shape_ok)
        // (no memory for the operand copy: the model works without it — every call then builds it in the ctx workspace, or scans)
        if (hipMalloc(&mo->screen_mem, jch_knn_screen_model_bytes(n, (int)dd)) != hipSuccess) { mo->screen_mem = nullptr; (void)hipGetLastError(); }
        else {
            const int32_t st = jch_knn_screen_build(ctx, mo->Zt, n, n, (int)dd, mo->screen_mem, &mo->screen);
            if (st != JCH_OK) return fail(st);
            mo->has_screen = true;
        }
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
        return fail(jch_fail(ctx, JCH_EHIP, "jch_lwplsr_prepare: row-major copy failed"));
    *model_out = mo;
    return JCH_OK;
}

extern "C" int32_t jch_lwplsr_release(jch_ctx *ctx, jch_lwplsr_model *model)
{
    if (!model) return JCH_OK;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(model->Xrm); (void)hipFree(model->Y); (void)hipFree(model->Zt); (void)hipFree(model->screen_mem);
    for (auto &qm : model->qmaps) (void)hipFree(qm.dB);
    delete model;
    return JCH_OK;
}

// Appends one affine stage Z <- ((Z - shift) ./ scale) B + bias (the arithmetic of jch_affine_gemm, folded the same way) to the
// model's query map.  With a map in place jch_lwplsr_predict_prepared may be called with Zq = NULL: the query scores are then
// computed from Xq on the device, on the ctx stream, with no upload and no host synchronisation per stage (the two
// jch_affine_gemm calls per predict they replace cost 0.15 ms of a 1.9 ms call at cfg5).  Stage 1 takes p columns (the
// model's), every later stage the previous one's k; the last stage must deliver the model's dd columns.
extern "C" int32_t jch_lwplsr_add_query_map(jch_ctx *ctx, jch_lwplsr_model *model, const double *shift, const double *scale, const double *B,
                                            int64_t p_in, int64_t k_out, const double *bias)
{
    if (!ctx) return JCH_EINVAL;
    if (!model || !B || p_in < 1 || k_out < 1) return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_add_query_map: bad arguments");
    if (model->device != ctx->device) return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_add_query_map: the model lives on device %d, the ctx on %d", model->device, ctx->device);
    const int64_t expect = model->qmaps.empty() ? model->p : model->qmaps.back().k_out;
    if (p_in != expect) return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_add_query_map: this stage takes %lld columns, the previous one delivers %lld", (long long)p_in, (long long)expect);
    if (model->qmaps.size() >= 4) return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_add_query_map: at most 4 stages");
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    const int kpad = (int)((k_out + 15) / 16 * 16);
    std::vector<double> hb((size_t)p_in * kpad + kpad, 0.0);      // the fold of jch_affine_gemm: Bs = diag(1 / scale) B, bias' = bias - shift' Bs
    double *Bs = hb.data(), *b2 = hb.data() + (size_t)p_in * kpad;
    for (int64_t c = 0; c < k_out; ++c) {
        double acc = bias ? bias[c] : 0.0;
        for (int64_t j = 0; j < p_in; ++j) {
            const double v = B[j + c * p_in] / (scale ? scale[j] : 1.0);
            Bs[j * kpad + c] = v;
            if (shift) acc -= shift[j] * v;
        }
        b2[c] = acc;
    }
    jch_lwplsr_model::qmap qm;
    qm.p_in = (int)p_in; qm.k_out = (int)k_out; qm.kpad = kpad;
    if (hipMalloc((void **)&qm.dB, sizeof(double) * hb.size()) != hipSuccess) return jch_fail(ctx, JCH_ENOMEM, "jch_lwplsr_add_query_map: device allocation failed");
    if (hipMemcpy(qm.dB, hb.data(), sizeof(double) * hb.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(qm.dB); return jch_fail(ctx, JCH_EHIP, "jch_lwplsr_add_query_map: upload failed"); }
    model->qmaps.push_back(qm);
    return JCH_OK;
}

extern "C" int32_t jch_lwplsr_predict_prepared(jch_ctx *ctx, const jch_lwplsr_model *model, int32_t loc, const double *Zq, int64_t ldzq,
                                               const double *Xq, int64_t m, int64_t ldxq, int32_t k, double h, double tol, int32_t scal,
                                               int32_t nlv_lo, int32_t nlv_hi, double *pred, int32_t *ind_out, double *dist_out, double *w_out)
{
    if (!ctx) return JCH_EINVAL;
    if (!model || !Xq || !pred || ldxq < m || (Zq && ldzq < m)) return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_predict_prepared: bad arguments");
    if (!Zq && (model->qmaps.empty() || model->qmaps.back().k_out != model->dd))
        return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_predict_prepared: Zq is NULL and the model has no query map ending in %lld columns (jch_lwplsr_add_query_map)", (long long)model->dd);
    if (model->device != ctx->device) return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_predict_prepared: the model lives on device %d, the ctx on %d", model->device, ctx->device);
    JCH_TRY(lw_check(ctx, "jch_lwplsr_predict_prepared", model->n, model->p, model->q, model->dd, m, k, nlv_lo, nlv_hi));
    if (loc != JCH_LOC_HOST && loc != JCH_LOC_DEVICE) return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_predict_prepared: bad loc");
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t p = model->p, dd = model->dd;
    const double *dZq = Zq, *dXq = Xq;
    int64_t ldzqd = ldzq, ldxqd = ldxq;
    if (loc == JCH_LOC_HOST) {
        JCH_TRY(jch_reserve(ctx, ctx->xq, sizeof(double) * ((size_t)m * dd + (size_t)m * p)));
        double *b = (double *)ctx->xq.ptr;
        if (Zq) JCH_HIP(ctx, hipMemcpy2DAsync(b, sizeof(double) * m, Zq, sizeof(double) * ldzq, sizeof(double) * m, dd, hipMemcpyHostToDevice, ctx->stream));
        JCH_HIP(ctx, hipMemcpy2DAsync(b + (size_t)m * dd, sizeof(double) * m, Xq, sizeof(double) * ldxq, sizeof(double) * m, p, hipMemcpyHostToDevice, ctx->stream));
        dZq = Zq ? b : nullptr; ldzqd = m; dXq = b + (size_t)m * dd; ldxqd = m;
    }
    ctx->ev_used = 0;
    ctx->prof = jch_profile{};
    hipEvent_t ev0 = jch_ev(ctx);
    if (!Zq) {   // the model's query map: Xq -> the neighbour-search space, stage by stage in two alternating device buffers
        size_t wmax = 0;
        for (const auto &qm : model->qmaps) wmax = std::max(wmax, (size_t)qm.k_out);
        JCH_TRY(jch_reserve(ctx, ctx->qz, sizeof(double) * 2 * (size_t)m * wmax));
        double *zb[2] = {(double *)ctx->qz.ptr, (double *)ctx->qz.ptr + (size_t)m * wmax};
        const double *src = dXq;
        int64_t lds_ = ldxqd;
        int w = 0;
        for (const auto &qm : model->qmaps) {
            JCH_TRY(jch_launch_affine_gemm(ctx, src, m, qm.p_in, lds_, qm.dB, qm.k_out, qm.kpad, qm.dB + (size_t)qm.p_in * qm.kpad, zb[w], m));
            src = zb[w]; lds_ = m; w ^= 1;
        }
        dZq = src; ldzqd = m;
    }
    return lw_run(ctx, model->Xrm, model->ldr, model->n, p, model->Y, model->q, model->n, model->Zt, model->n, dZq, ldzqd, dd, dXq, m, ldxqd, k, h, tol,
                  scal, nlv_lo, nlv_hi, pred, ind_out, dist_out, w_out, ev0, model->has_screen ? &model->screen : nullptr, &model->screen_off);
}

// Weighted (uncorrected) covariance of the columns of A (n x d, d <= 64): S = (A - 1 mu')' D (A - 1 mu'), the
// `Statistics.cov(Xtrain, corrected = false)` of getknn's Mahalanobis branch (src/getknn.jl:38).  Reuses K0/K1/K2
// with Y = A.  S (d x d, column-major) and mu (d) on the HOST.
extern "C" int32_t jch_weighted_cov(jch_ctx *ctx, int32_t loc, const double *A, int64_t n, int64_t d, int64_t lda,
                                    const double *weights, double *S, double *mu)
{
    if (!ctx) return JCH_EINVAL;
    if (!A || !S || n < 1 || d < 1 || d > (1 << 15) || lda < n) return jch_fail(ctx, JCH_EINVAL, "jch_weighted_cov: bad arguments");
    if (loc != JCH_LOC_HOST && loc != JCH_LOC_DEVICE) return jch_fail(ctx, JCH_EINVAL, "jch_weighted_cov: bad loc");
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    const int dd = (int)d, ldr = (dd + 1) & ~1, qpad = ((dd + 15) / 16) * 16;
    const double *dA = A, *dw = weights;
    int64_t ldad = lda;
    if (loc == JCH_LOC_HOST) {
        JCH_TRY(jch_reserve(ctx, ctx->xq, sizeof(double) * ((size_t)n * d + (size_t)n)));
        double *b = (double *)ctx->xq.ptr;
        if (lda == n) JCH_HIP(ctx, hipMemcpyAsync(b, A, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, ctx->stream));
        else JCH_HIP(ctx, hipMemcpy2DAsync(b, sizeof(double) * n, A, sizeof(double) * lda, sizeof(double) * n, d, hipMemcpyHostToDevice, ctx->stream));
        dA = b; ldad = n;
        if (weights) {
            JCH_HIP(ctx, hipMemcpyAsync(b + (size_t)n * d, weights, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
            dw = b + (size_t)n * d;
        }
    }
    JCH_TRY(jch_reserve(ctx, ctx->dnorm, sizeof(double) * (size_t)n));
    ctx->xcopy_valid = false;   // (the fit workspace's row-major copy is overwritten here)
    JCH_TRY(jch_reserve(ctx, ctx->xr, sizeof(double) * (size_t)n * ldr));
    if (d > JCH_MAXQ) {
        // wide A (round 4; the Mahalanobis branch of getknn on raw spectra, nlvdis = 0: src/getknn.jl:37-49, src/lwplsr.jl:21): the
        // centred row-major copy from K2 (with A's first column standing in for Y), then the tiled MFMA SYRK of kern2.hip on it
        JCH_TRY(jch_reserve(ctx, ctx->yr, sizeof(double) * (size_t)n * 16));
        JCH_TRY(jch_reserve(ctx, ctx->small, sizeof(double) * ((size_t)dd * 16 + 4 * (size_t)dd + 64) + 4096));
        JCH_TRY(jch_reserve(ctx, ctx->gram, sizeof(double) * (size_t)dd * ldr));
        double *K1 = (double *)ctx->small.ptr, *mom1 = K1 + (size_t)dd * 16, *scl1 = mom1 + dd + 1, *hdr1 = scl1 + dd + 2;
        double *dn1 = (double *)ctx->dnorm.ptr, *G = (double *)ctx->gram.ptr;
        JCH_TRY(jch_launch_weights(ctx, dw, n, dn1, hdr1));
        JCH_TRY(jch_launch_moments(ctx, dA, ldad, dA, ldad, dn1, n, dd, 1, nullptr, mom1));
        std::vector<double> ones1((size_t)dd + 1, 1.0);
        JCH_HIP(ctx, hipMemcpyAsync(scl1, ones1.data(), sizeof(double) * ((size_t)dd + 1), hipMemcpyHostToDevice, ctx->stream));
        JCH_TRY(jch_launch_center_xty(ctx, const_cast<double *>(dA), ldad, const_cast<double *>(dA), ldad, dn1, n, dd, 1, mom1, scl1, false,
                                      (double *)ctx->xr.ptr, ldr, (double *)ctx->yr.ptr, 16, K1, false));
        JCH_TRY(jch_launch_syrk(ctx, (const double *)ctx->xr.ptr, n, dd, ldr, dn1, G, ldr));
        std::vector<double> hG((size_t)dd * ldr), hm1(dd);
        JCH_HIP(ctx, hipMemcpyAsync(hG.data(), G, sizeof(double) * hG.size(), hipMemcpyDeviceToHost, ctx->stream));
        JCH_HIP(ctx, hipMemcpyAsync(hm1.data(), mom1, sizeof(double) * dd, hipMemcpyDeviceToHost, ctx->stream));
        JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < dd; ++i)
            for (int j = 0; j < dd; ++j) S[i + (size_t)j * dd] = hG[(size_t)i * ldr + j];
        if (mu) for (int i = 0; i < dd; ++i) mu[i] = hm1[i];
        return JCH_OK;
    }
    JCH_TRY(jch_reserve(ctx, ctx->yr, sizeof(double) * (size_t)n * qpad));
    JCH_TRY(jch_reserve(ctx, ctx->small, sizeof(double) * ((size_t)dd * qpad + 4 * (size_t)dd + 64) + 4096));
    double *K = (double *)ctx->small.ptr, *mom = K + (size_t)dd * qpad, *scl = mom + 2 * dd, *hdr = scl + 2 * dd;
    double *dn = (double *)ctx->dnorm.ptr;
    JCH_TRY(jch_launch_weights(ctx, dw, n, dn, hdr));
    JCH_TRY(jch_launch_moments(ctx, dA, ldad, dA, ldad, dn, n, dd, dd, nullptr, mom));
    std::vector<double> ones(2 * (size_t)dd, 1.0);
    JCH_HIP(ctx, hipMemcpyAsync(scl, ones.data(), sizeof(double) * 2 * dd, hipMemcpyHostToDevice, ctx->stream));
    JCH_TRY(jch_launch_center_xty(ctx, const_cast<double *>(dA), ldad, const_cast<double *>(dA), ldad, dn, n, dd, dd, mom, scl, false,
                                  (double *)ctx->xr.ptr, ldr, (double *)ctx->yr.ptr, qpad, K, false));
    std::vector<double> hK((size_t)dd * qpad), hm(dd);
    JCH_HIP(ctx, hipMemcpyAsync(hK.data(), K, sizeof(double) * hK.size(), hipMemcpyDeviceToHost, ctx->stream));
    JCH_HIP(ctx, hipMemcpyAsync(hm.data(), mom, sizeof(double) * dd, hipMemcpyDeviceToHost, ctx->stream));
    JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < dd; ++i)
        for (int j = 0; j < dd; ++j) S[i + (size_t)j * dd] = hK[(size_t)i * qpad + j];
    if (mu) for (int i = 0; i < dd; ++i) mu[i] = hm[i];
    return JCH_OK;
}
