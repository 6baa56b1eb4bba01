// Arguments shared by the batched local-fit kernels of the kNN-LWPLSR prediction path (lwplsr.hip: k_locw_plskern, the
// p-space kernel; lwplsr_kspace.hip: k_locw_kspace, the Gram / k-space kernel) — src/locwlv.jl:9-48.
#pragma once
#include "jch_internal.h"

struct locw_args {
    const double *Xrm; int ldr; int p;      // row-major training X (uncentred)
    const double *Y; int64_t ldy; int q;    // training Y (column-major n x q)
    const double *Xq; int64_t ldxq; int m;  // queries, column-major m x p
    const int *ind; const double *w; int k; // neighbours / weights [m][k]
    int scal, nlv_lo, nlv_hi;
    double *scratch; size_t slab;           // per-block global scratch (p-space kernel: Xg [k][ldr], P [nlv][ldr], R [nlv][ldr]; k-space kernel: [ldr] column stds)
    double *pred;                           // [m][le][q], le = nlv_hi - nlv_lo + 1
    int dbg;                                // measurement switches (JCH_LOCW_DBG; results then wrong by design): 1 = every query gathers rows 0 .. k-1
    int *flags;                             // [m] or null.  k-space kernel: flags[i] = 1 when query i lies too far from its neighbours for the
                                            // Gram matrix about the query row (pivot check, lwplsr_kspace.hip): the caller refits those queries
};

// lwplsr_kspace.hip: _feasible: the k-space kernel can take this shape (k <= 208, q <= 8, nlv <= 48, p <= 2048);
// _supported: ... and is expected to beat the p-space kernel there
bool jch_locw_kspace_feasible(const locw_args &g);
bool jch_locw_kspace_supported(const locw_args &g);
int32_t jch_launch_locw_kspace(jch_ctx *ctx, locw_args &g);

// Arguments of the kNN + weights stage (lwplsr.hip: k_knn_scan / k_knn_finish; lwplsr_generic.hip: k_knn_generic) — src/getknn.jl:29-57,
// src/wdist.jl:64-75.
struct knn_args {
    const double *Zt; int64_t ldzt; int64_t n;   // train scores, column-major n x dd
    const double *Zq; int64_t ldzq; int m;       // query scores, column-major m x dd
    int dd, k;
    double h, cri, tol;
    int *ind;      // [m][k]
    double *dist;  // [m][k]
    double *w;     // [m][k]
    int nseg;      // the training rows are scanned in nseg segments by different workgroups (block b: segment b % nseg, query group b / nseg)
    double *ckey;  // [m][nseg][k] squared distances of every segment's k best (ascending; +inf beyond the segment's rows)
    int *cidx;     // [m][nseg][k]
    int dbg;       // measurement switch (JCH_KNN_DBG; results then wrong by design): 1 = the bar starts at -inf (no candidate is ever kept: the bare scan)
    const int *only_flags = nullptr;   // null, or [m] device flags — only the queries with a non-zero flag are done (k_knn_scan: only their groups of qb)
};

// lwplsr_screen.hip: the screened kNN (round 4).  Squared distances of ALL (row, query) pairs on the matrix cores
// (v_mfma_f32_32x32x16_bf16 on two-piece bf16, norm-augmented operands), an error-bounded bar per query from the k-th smallest GROUP minimum, exact
// f64 distances only for the survivors (k .. ~1.2 k rows per query); queries the screen cannot settle (non-finite scores, more
// survivors than the candidate list holds) are flagged and done by k_knn_generic.  Results identical to k_knn_scan's.
struct knn_screen {          // the model-constant part: built once per prepared model, or per call in the ctx workspace
    uint4 *Zs = nullptr;     // [ntiles][KS][64 lanes] operand-ordered two-piece bf16 copy of the centred training scores (+ the |z|^2 and 1 slots)
    double *Zr = nullptr;    // [n][ldzr] row-major f64 copy of the scores (uncentred, the caller's values): one 8 dd-byte read per exact distance
    int ldzr = 0;
    double *mu = nullptr;    // [dd] column means the copy is centred on (distances do not depend on them)
    unsigned *hdr = nullptr; // [0] bits of max |z|^2 (f32) over the rows, [1] non-zero: a training score is not finite
    int KS = 0;              // k-steps of 16 operand slots: 16 KS >= 3 dd + 4
    int64_t ntiles = 0;      // 32-row tiles
};
bool jch_knn_screen_shape_ok(int64_t n, int dd, int k);
size_t jch_knn_screen_model_bytes(int64_t n, int dd);
// mem: jch_knn_screen_model_bytes(n, dd) bytes of device memory (256-B aligned) that `out` is carved from
int32_t jch_knn_screen_build(jch_ctx *ctx, const double *dZt, int64_t ldzt, int64_t n, int dd, void *mem, knn_screen *out);
// flags: [m] device ints, set to 1 for the queries handed to the exact selection (0 otherwise)
int32_t jch_launch_knn_screen(jch_ctx *ctx, const knn_args &a, const knn_screen &sc, int *flags);

// lwplsr.hip: row-major copy Xrm [n][ldr] of a column-major n x p matrix (columns p .. ldr - 1 zero), on the ctx stream
void jch_lw_to_rowmajor(jch_ctx *ctx, const double *dX, int64_t ldxd, int64_t n, int p, double *Xrm, int ldr);
// lwplsr.hip: the exact scan (k <= 768, LDS for the score space: jch_knn_scan_lds <= 150 KB)
size_t jch_knn_scan_lds(int k, int dd, int m, bool *wide_out);
int32_t jch_launch_knn_scan(jch_ctx *ctx, knn_args a, jch_buf &cbuf);

// lwplsr_generic.hip: the paths WITHOUT shape limits (any k <= n, any p, q, nlv) behind the batched kernels' envelope.
// kNN + weights of all m queries: exact selection of the k smallest distances per query, (distance, index) order, wdist weights
int32_t jch_launch_knn_generic(jch_ctx *ctx, const knn_args &a);
// local fits one query at a time: gather the neighbour rows, jch_plskern_fit on them, jch_predict on the query row (the
// reference's own schedule, src/locwlv.jl:18-39); dpred [m][le][q] device
int32_t jch_lw_generic_fits(jch_ctx *ctx, const locw_args &g, int64_t n, const int *only = nullptr /*host: query indices to fit (null: all)*/, int n_only = 0);

#ifdef __HIPCC__
#define KNN_CAP 1024   // candidate buffer per query (LDS) of the scan; the finishing kernels order at most this many (distance, index) pairs

// bitonic sort of `cap` (a power of two <= KNN_CAP) (key, idx) pairs in LDS, ascending by (key, idx); NT threads
template <int NT>
__device__ static void bitonic_sort_n(double *key, int *idx, int cap)
{
    const int tid = threadIdx.x;
    for (int size = 2; size <= cap; size <<= 1) {
        for (int stride = size >> 1, ls = 31 - __builtin_clz(size >> 1); stride > 0; stride >>= 1, --ls) {
            __syncthreads();
            for (int t = tid; t < cap / 2; t += NT) {
                // (shifts, not t / stride and t % stride: a runtime integer division is ~40 instructions on this ISA and was
                // 3/4 of the sort's time)
                const int lo = ((t >> ls) << (ls + 1)) | (t & (stride - 1)), hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const double a = key[lo], b = key[hi];
                const int ia = idx[lo], ib = idx[hi];
                const bool gt = (a > b) || (a == b && ia > ib) || (a != a && b == b);   // NaN sorts last
                if (gt == up) { key[lo] = b; key[hi] = a; idx[lo] = ib; idx[hi] = ia; }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void knn_wavesync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// The same network run by ONE wave (no workgroup barrier between the passes: a wave's LDS operations complete in order): the four
// waves of a scan workgroup sort the four queries' buffers side by side — 7 us per 1024 entries against 14 us x 4 queries with
// the workgroup-wide sort, which was half of the scan's time (JCH_KNN_DBG=1 measures the scan without any candidate kept).
template <int PP>   // PP = pairs per lane and pass = cap / 128 (1 for cap <= 128): all of a pass's loads go out together
__device__ __forceinline__ void bitonic_sort_wave_pp(double *key, int *idx, int cap)
{
    const int lane = threadIdx.x & 63;
    for (int size = 2; size <= cap; size <<= 1) {
        for (int stride = size >> 1, ls = 31 - __builtin_clz(size >> 1); stride > 0; stride >>= 1, --ls) {
            knn_wavesync();
            double a[PP], b[PP];
            int ia[PP], ib[PP], lo[PP];
#pragma unroll
            for (int u = 0; u < PP; ++u) {
                const int t = lane + 64 * u;
                lo[u] = ((t >> ls) << (ls + 1)) | (t & (stride - 1));
                const int l = t < cap / 2 ? lo[u] : 0;
                a[u] = key[l]; b[u] = key[l + stride]; ia[u] = idx[l]; ib[u] = idx[l + stride];
            }
#pragma unroll
            for (int u = 0; u < PP; ++u) {
                const bool up = ((lo[u] & size) == 0);
                const bool gt = (a[u] > b[u]) || (a[u] == b[u] && ia[u] > ib[u]) || (a[u] != a[u] && b[u] == b[u]);   // NaN sorts last
                if (lane + 64 * u < cap / 2 && gt == up) { key[lo[u]] = b[u]; key[lo[u] + stride] = a[u]; idx[lo[u]] = ib[u]; idx[lo[u] + stride] = ia[u]; }
            }
        }
    }
    knn_wavesync();
}
__device__ static void bitonic_sort_wave(double *key, int *idx, int cap)
{
    if (cap <= 128) bitonic_sort_wave_pp<1>(key, idx, cap);
    else if (cap == 256) bitonic_sort_wave_pp<2>(key, idx, cap);
    else if (cap == 512) bitonic_sort_wave_pp<4>(key, idx, cap);
    else bitonic_sort_wave_pp<8>(key, idx, cap);
}

// 64 (key, idx) pairs, one per lane, into ascending (key, idx) order across the lanes of a wave: the bitonic network on registers
// (21 exchange steps through the lanes, no LDS, no barrier)
__device__ __forceinline__ void knn_sort64_lanes(double &kv, int &iv)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const double pk = __shfl_xor(kv, stride, 64);
            const int pi = __shfl_xor(iv, stride, 64);
            const bool keepmin = ((lane & stride) == 0) == ((lane & size) == 0);
            const bool p_first = (pk < kv) || (pk == kv && pi < iv);      // the partner's pair comes before mine
            const bool m_first = (kv < pk) || (kv == pk && iv < pi);
            const bool take = keepmin ? p_first : m_first;
            kv = take ? pk : kv;
            iv = take ? pi : iv;
        }
}

// The tail shared by the finishing kernels (k_knn_finish, k_knn_finish_screen): okey / oidx [kk] hold the query's nearest squared
// distances and rows in (distance, index) order (sentinels +inf / 0x7fffffff where fewer than kk were found); writes the neighbour
// list, the distances and the wdist weights (src/wdist.jl:64-75).  256 threads; key: scratch of >= kk doubles; every thread calls.
__device__ static void knn_finish_tail(const knn_args &g, int qi, int kk, double *key, double *okey, int *oidx, double *sred, double *smed, int *snn)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int k = g.k;
    int *oi = g.ind + (size_t)qi * k;
    double *od = g.dist + (size_t)qi * k, *ow = g.w + (size_t)qi * k;
    // fewer than k candidates with a finite distance (NaN / Inf in the query's or the training scores: no comparison
    // against the bar ever holds): the empty places keep the sentinel index.  They are given the in-range row `e` and a
    // NaN distance — wdist then yields weights 1 for the query exactly as the reference's arithmetic does (every
    // comparison with NaN is false, 0 / 0 -> NaN -> 1, src/wdist.jl:64-75) and nothing downstream reads out of bounds.
    for (int e = tid; e < kk; e += 256) {
        const bool hole = oidx[e] < 0 || (int64_t)oidx[e] >= g.n;
        oi[e] = hole ? e : oidx[e];
        const double dv = hole ? __builtin_nan("") : sqrt(okey[e]);
        okey[e] = dv;
        od[e] = dv;
    }
    __syncthreads();
    // wdist (src/wdist.jl:64-75): median, MAD, cutoff, exp weights, / max, NaN -> 1, then the tol clamp
    const double med = (kk & 1) ? okey[kk / 2] : 0.5 * (okey[kk / 2 - 1] + okey[kk / 2]);
    // the median of |d - med|: the values at the places t0, t1 of the kk deviations in ascending order, NaN last.
    const int t1 = kk / 2, t0 = (kk & 1) ? -1 : kk / 2 - 1;
    const double dlast = okey[kk - 1];
    if (med == med && fabs(med) < __builtin_inf() && dlast == dlast && fabs(dlast) < __builtin_inf()) {   // (block-uniform)
        // All distances finite (they ascend; holes, NaN, would be last).  The deviations are then two monotone runs — descending up
        // to the first distance >= med, ascending from there — and an entry's place in their merged order is its place in its own
        // run plus a binary search of the other (round 4; was a count over all kk entries per entry: kk^2 steps, 10 of the
        // finishing kernel's 55 us at cfg5).  Ties: the descending run's entries first — any order of equal values puts the same
        // VALUE at a place.
        for (int e = tid; e < kk; e += 256) key[e] = fabs(okey[e] - med);
        if (tid == 0) snn[0] = kk;
        if (tid < 2) smed[tid] = __builtin_nan("");
        __syncthreads();
        for (int e = tid; e < kk; e += 256)
            if (okey[e] >= med && (e == 0 || okey[e - 1] < med)) snn[0] = e;
        __syncthreads();
        const int p = snn[0];
        for (int e = tid; e < kk; e += 256) {
            const double v = key[e];
            int r;
            if (e < p) {            // descending run, read backwards: ascending; entries of the other run strictly below v come first
                int lo = 0, hi = kk - p;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (key[p + mid] < v) lo = mid + 1; else hi = mid; }
                r = (p - 1 - e) + lo;
            } else {                // ascending run; entries of the other run at or below v come first
                int lo = 0, hi = p;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (key[p - 1 - mid] <= v) lo = mid + 1; else hi = mid; }
                r = (e - p) + lo;
            }
            if (r == t1) smed[1] = v;
            if (r == t0) smed[0] = v;
        }
    } else {
        // (non-finite distances) by RANK COUNTING: entry e's place — by value, then by position, NaN last — is the number of entries
        // that come before it
        int nn = 0;
        for (int e = tid; e < kk; e += 256) { const double v = fabs(okey[e] - med); key[e] = v; nn += v == v ? 1 : 0; }
        for (int o = 32; o > 0; o >>= 1) nn += __shfl_xor(nn, o, 64);
        if (lane == 0) snn[wv] = nn;
        if (tid < 2) smed[tid] = __builtin_nan("");                // (a target place among the NaNs stays NaN)
        __syncthreads();
        nn = snn[0] + snn[1] + snn[2] + snn[3];
        for (int e = tid; e < kk; e += 256) {
            const double v = key[e];
            if (v != v) continue;
            int r = 0;
            for (int f = 0; f < kk; ++f) { const double u = key[f]; r += (u < v || (u == v && f < e)) ? 1 : 0; }
            if (r == t1) smed[1] = v;
            if (r == t0) smed[0] = v;
        }
    }
    __syncthreads();
    const double zmad = 1.4826 * ((kk & 1) ? smed[1] : 0.5 * (smed[0] + smed[1]));
    const double cutoff = med + g.cri * zmad;
    // weights; max with NaN propagation (Julia's `maximum` returns NaN if any NaN is present)
    double wmax = -__builtin_inf();
    int anynan = 0;
    __syncthreads();                                           // (key: the deviations are done with, the weights go there)
    for (int e = tid; e < kk; e += 256) {
        const double dv = okey[e];
        const double wv_ = dv <= cutoff ? exp(-dv / (g.h * zmad)) : 0.0;
        key[e] = wv_;
        if (wv_ != wv_) anynan = 1;
        else if (wv_ > wmax) wmax = wv_;
    }
    for (int o = 32; o > 0; o >>= 1) { wmax = fmax(wmax, __shfl_xor(wmax, o, 64)); anynan |= __shfl_xor(anynan, o, 64); }
    if (lane == 0) { sred[wv] = wmax; sred[4 + wv] = (double)anynan; }
    __syncthreads();
    wmax = fmax(fmax(sred[0], sred[1]), fmax(sred[2], sred[3]));
    if (sred[4] + sred[5] + sred[6] + sred[7] > 0.0) wmax = __builtin_nan("");
    for (int e = tid; e < kk; e += 256) {
        double wv_ = key[e] / wmax;
        if (wv_ != wv_) wv_ = 1.0;
        if (wv_ < g.tol) wv_ = g.tol;
        ow[e] = wv_;
    }
}
#endif
