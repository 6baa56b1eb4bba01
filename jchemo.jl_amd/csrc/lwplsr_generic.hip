// kNN-LWPLSR prediction path WITHOUT shape limits (round 4): what runs when a call is outside the envelope of the batched kernels
// of lwplsr.hip / lwplsr_kspace.hip (k <= 768 neighbours; local fits with p <= 2048, q <= 16, nlv <= 48 and the LDS budget).  The
// reference has no such limits (src/getknn.jl:29-57, src/locwlv.jl:9-48), so neither may a drop-in: these paths are slower, never
// absent.
//   k_knn_generic      one workgroup per query: all n squared distances into a global scratch row (the same expression, in the same
//                      column order, as k_knn_scan: identical bits), the k-th smallest by an 8-pass radix select on the bit
//                      patterns, the k selected rows (ties at the k-th distance: lowest row indices first, as the batched path and
//                      the oracle do), a bitonic sort of the k (distance, index) pairs in global memory, then the wdist weights
//                      (src/wdist.jl:64-75) with the median of the absolute deviations from a second sort.  Any k <= n, any
//                      score dimension.
//   jch_lw_generic_fits   the reference's own schedule (src/locwlv.jl:18-39): per query, gather the k neighbour rows into a
//                      column-major slab on the device and run jch_plskern_fit (weights = the query's kNN weights) and
//                      jch_predict (the query row, the whole nlv range) on it — every kernel of the global fit applies, so any
//                      p (wide sweep), q (generic small state) and nlv work; the constant-y shortcut (:25-28) is decided for all
//                      queries by one kernel up front.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "jch_internal.h"
#include "lwplsr_dev.h"

#define KG_NT 256

// ascending bitonic sort of cap (power of two) (key, idx) pairs by (key, idx), NaN last; any address space; KG_NT threads
__device__ static void kg_bitonic(double *key, int *idx, int cap)
{
    const int tid = threadIdx.x;
    for (int size = 2; size <= cap; size <<= 1) {
        for (int stride = size >> 1, ls = 31 - __builtin_clz(size >> 1); stride > 0; stride >>= 1, --ls) {
            __syncthreads();
            for (int t = tid; t < cap / 2; t += KG_NT) {
                const int lo = ((t >> ls) << (ls + 1)) | (t & (stride - 1)), hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const double a = key[lo], b = key[hi];
                const int ia = idx[lo], ib = idx[hi];
                const bool gt = (a > b) || (a == b && ia > ib) || (a != a && b == b);
                if (gt == up) { key[lo] = b; key[hi] = a; idx[lo] = ib; idx[hi] = ia; }
            }
        }
    }
    __syncthreads();
}

struct kg_scratch {
    double *d2;     // [nblk][n]
    double *skey;   // [nblk][K2]
    int *sidx;      // [nblk][K2]
    double *dkey;   // [nblk][K2]
    int *didx;      // [nblk][K2]
    int K2;
};

__global__ __launch_bounds__(KG_NT) void k_knn_generic(knn_args g, kg_scratch sc)
{
    extern __shared__ __attribute__((aligned(16))) double zq[];   // [dd]
    __shared__ int hist[256];
    __shared__ unsigned long long s_prefix, s_mask;
    __shared__ int s_remaining;
    __shared__ int wl[KG_NT / 64], we[KG_NT / 64];
    __shared__ double sred[8];
    __shared__ int snn[KG_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t n = g.n;
    const int k = g.k, K2 = sc.K2;
    double *d2 = sc.d2 + (size_t)blockIdx.x * (size_t)n;
    double *skey = sc.skey + (size_t)blockIdx.x * K2, *dkey = sc.dkey + (size_t)blockIdx.x * K2;
    int *sidx = sc.sidx + (size_t)blockIdx.x * K2, *didx = sc.didx + (size_t)blockIdx.x * K2;
    if (g.only_flags) {   // behind the screened search: nothing flagged among this block's queries (the rule) -> one round trip and out
        int any = 0;
        for (int qi = blockIdx.x + tid * (int)gridDim.x; qi < g.m; qi += KG_NT * (int)gridDim.x) any |= g.only_flags[qi];
        if (!__syncthreads_or(any)) return;
    }
    for (int qi = blockIdx.x; qi < g.m; qi += gridDim.x) {
        if (g.only_flags && !g.only_flags[qi]) continue;       // (block-uniform)
        __syncthreads();
        for (int c = tid; c < g.dd; c += KG_NT) zq[c] = g.Zq[(size_t)qi + (size_t)c * (size_t)g.ldzq];
        __syncthreads();
        // ---- squared distances, columns in order (the expression of k_knn_scan)
        for (int64_t i = tid; i < n; i += KG_NT) {
            double acc = 0.0;
            for (int c0 = 0; c0 < g.dd; c0 += 8) {
                double x[8];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) x[cc] = g.Zt[(size_t)i + (size_t)min(c0 + cc, g.dd - 1) * (size_t)g.ldzt];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc)
                    if (c0 + cc < g.dd) { const double e = x[cc] - zq[c0 + cc]; acc += e * e; }
            }
            d2[i] = acc;
        }
        // ---- k-th smallest distance: radix select on the bit patterns (non-negative doubles order like their bits; NaN above +inf)
        if (tid == 0) { s_prefix = 0ull; s_mask = 0ull; s_remaining = k; }
        for (int pass = 7; pass >= 0; --pass) {
            const int shift = pass * 8;
            hist[tid] = 0;                                   // (KG_NT == 256)
            __syncthreads();
            const unsigned long long prefix = s_prefix, mask = s_mask;
            for (int64_t i = tid; i < n; i += KG_NT) {
                const unsigned long long key = (unsigned long long)__double_as_longlong(d2[i]);
                if ((key & mask) == prefix) atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int cum = 0, b = 0;
                const int rem = s_remaining;
                for (; b < 255; ++b) { if (cum + hist[b] >= rem) break; cum += hist[b]; }
                s_remaining = rem - cum;
                s_prefix = prefix | ((unsigned long long)b << shift);
                s_mask = mask | (255ull << shift);
            }
            __syncthreads();
        }
        const unsigned long long kth = s_prefix;
        const int need_eq = s_remaining, n_less = k - need_eq;   // rows strictly below the k-th key; of the rows AT it, the first need_eq
        // ---- the k selected rows: everything below the k-th key, then the ties in row order
        int base_less = 0, base_eq = 0;        // running counts, the same in every thread (each adds the four waves' totals itself)
        for (int64_t base = 0; base < n; base += KG_NT) {
            const int64_t i = base + tid;
            const double v = i < n ? d2[i] : 0.0;
            const unsigned long long key = (unsigned long long)__double_as_longlong(v);
            const bool isl = i < n && key < kth, ise = i < n && key == kth;
            const unsigned long long ml = __ballot(isl), me = __ballot(ise);
            if (lane == 0) { wl[wv] = __popcll(ml); we[wv] = __popcll(me); }
            __syncthreads();
            int ol = base_less, oe = base_eq, tl = 0, te = 0;
            for (int w = 0; w < KG_NT / 64; ++w) {
                if (w < wv) { ol += wl[w]; oe += we[w]; }
                tl += wl[w]; te += we[w];
            }
            const unsigned long long below = (1ull << lane) - 1ull;
            if (isl) { const int pos = ol + __popcll(ml & below); skey[pos] = v; sidx[pos] = (int)i; }
            if (ise) { const int rk = oe + __popcll(me & below); if (rk < need_eq) { skey[n_less + rk] = v; sidx[n_less + rk] = (int)i; } }
            base_less += tl; base_eq += te;
            __syncthreads();                       // (wl / we are rewritten by the next trip)
        }
        for (int e = k + tid; e < K2; e += KG_NT) { skey[e] = __builtin_inf(); sidx[e] = 0x7fffffff; }
        __syncthreads();
        kg_bitonic(skey, sidx, K2);
        // ---- neighbours, distances; wdist weights (the arithmetic of k_knn_finish)
        int *oi = g.ind + (size_t)qi * k;
        double *od = g.dist + (size_t)qi * k, *ow = g.w + (size_t)qi * k;
        // a NaN distance (NaN in the query's or in a selected row's scores) sorts BEHIND the padding of the sort buffer, so either
        // may sit among the first k entries: such a place is a hole and gets what k_knn_finish gives its holes — the in-range row
        // `e` and a NaN distance (knn_finish_tail, lwplsr_dev.h); nothing downstream gathers from the sentinel index
        for (int e = tid; e < k; e += KG_NT) {
            const int si = sidx[e];
            const double kv = skey[e];
            const bool hole = si < 0 || (int64_t)si >= n || kv != kv;
            const double dv = hole ? __builtin_nan("") : sqrt(kv);
            oi[e] = hole ? e : si;
            od[e] = dv;
            skey[e] = dv;
        }
        __syncthreads();
        const double med = (k & 1) ? skey[k / 2] : 0.5 * (skey[k / 2 - 1] + skey[k / 2]);
        for (int e = tid; e < K2; e += KG_NT) { dkey[e] = e < k ? fabs(skey[e] - med) : __builtin_nan(""); didx[e] = e; }
        __syncthreads();
        kg_bitonic(dkey, didx, K2);                            // NaN deviations (and the padding) last
        const double m1 = dkey[k / 2], m0 = (k & 1) ? m1 : dkey[k / 2 - 1];
        const double zmad = 1.4826 * ((k & 1) ? m1 : 0.5 * (m0 + m1));
        const double cutoff = med + g.cri * zmad;
        double wmax = -__builtin_inf();
        int anynan = 0;
        __syncthreads();
        for (int e = tid; e < k; e += KG_NT) {
            const double dv = skey[e];
            const double wv_ = dv <= cutoff ? exp(-dv / (g.h * zmad)) : 0.0;
            dkey[e] = wv_;
            if (wv_ != wv_) anynan = 1;
            else if (wv_ > wmax) wmax = wv_;
        }
        for (int o = 32; o > 0; o >>= 1) { wmax = fmax(wmax, __shfl_xor(wmax, o, 64)); anynan |= __shfl_xor(anynan, o, 64); }
        if (lane == 0) { sred[wv] = wmax; sred[4 + wv] = (double)anynan; }
        __syncthreads();
        wmax = fmax(fmax(sred[0], sred[1]), fmax(sred[2], sred[3]));
        if (sred[4] + sred[5] + sred[6] + sred[7] > 0.0) wmax = __builtin_nan("");
        for (int e = tid; e < k; e += KG_NT) {
            double wv_ = dkey[e] / wmax;
            if (wv_ != wv_) wv_ = 1.0;
            if (wv_ < g.tol) wv_ = g.tol;
            ow[e] = wv_;
        }
        (void)snn;
    }
}

int32_t jch_launch_knn_generic(jch_ctx *ctx, const knn_args &a)
{
    if (a.k < 1 || (int64_t)a.k > a.n || a.n >= ((int64_t)1 << 31) - 1) return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_predict: bad k / n for the kNN selection");
    if ((size_t)a.dd * sizeof(double) > 60 * 1024) return jch_fail(ctx, JCH_EINVAL, "jch_lwplsr_predict: neighbour-search space with %d dimensions (at most 7680)", a.dd);
    int K2 = 64;
    while (K2 < a.k) K2 <<= 1;
    const size_t per = sizeof(double) * ((size_t)a.n + 2 * (size_t)K2) + sizeof(int) * 2 * (size_t)K2;
    const size_t budget = (size_t)4 << 30;
    // (behind the screened kNN only the flagged queries — the exception — are done: a small grid, a small scratch)
    const size_t want = a.only_flags ? 64 : (size_t)ctx->cus * 2;
    int nblk = (int)std::min<size_t>(std::min<size_t>((size_t)a.m, want), std::max<size_t>(1, budget / per));
    JCH_TRY(jch_reserve(ctx, ctx->lw_work, per * nblk + 1024));
    kg_scratch sc;
    char *b = (char *)ctx->lw_work.ptr;
    sc.d2 = (double *)b; b += sizeof(double) * (size_t)a.n * nblk;
    sc.skey = (double *)b; b += sizeof(double) * (size_t)K2 * nblk;
    sc.dkey = (double *)b; b += sizeof(double) * (size_t)K2 * nblk;
    sc.sidx = (int *)b; b += sizeof(int) * (size_t)K2 * nblk;
    sc.didx = (int *)b;
    sc.K2 = K2;
    hipLaunchKernelGGL(k_knn_generic, dim3(nblk), dim3(KG_NT), sizeof(double) * (size_t)a.dd, ctx->stream, a, sc);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

// ---------------------------------------------------------------- local fits, one query at a time
// rows ind[0 .. k) of the row-major Xrm -> column-major slab Xl (k x p, ld k); 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void k_lw_gather_x(const double *__restrict__ Xrm, int ldr, int p, const int *__restrict__ ind, int k,
                                                     double *__restrict__ Xl)
{
    __shared__ double tile[64][65];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    for (int rr = wv; rr < 64; rr += 4) {
        const int row = r0 + rr, j = j0 + lane;
        tile[rr][lane] = (row < k && j < p) ? Xrm[(size_t)ind[row] * ldr + j] : 0.0;
    }
    __syncthreads();
    for (int cc = wv; cc < 64; cc += 4) {
        const int col = j0 + cc, r = r0 + lane;
        if (col < p && r < k) Xl[(size_t)r + (size_t)col * k] = tile[lane][cc];
    }
}
__global__ __launch_bounds__(256) void k_lw_gather_y(const double *__restrict__ Y, int64_t ldy, int q, const int *__restrict__ ind, int k,
                                                     double *__restrict__ Yl)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= k * q) return;
    const int y = e / k, r = e - y * k;
    Yl[e] = Y[(size_t)ind[r] + (size_t)y * (size_t)ldy];
}
// q == 1: flag[i] = 1 when all neighbour responses of query i are equal (src/locwlv.jl:25), val[i] = that response
__global__ __launch_bounds__(256) void k_lw_const_y(const double *__restrict__ Y, const int *__restrict__ ind, int k, int m, double *__restrict__ out)
{
    __shared__ int diff;
    const int qi = blockIdx.x;
    if (threadIdx.x == 0) diff = 0;
    __syncthreads();
    const double y0 = Y[ind[(size_t)qi * k]];
    int d = 0;
    for (int e = threadIdx.x; e < k; e += 256) d |= (Y[ind[(size_t)qi * k + e]] != y0) ? 1 : 0;   // (`unique`: isequal — NaN handled below)
    if (d) atomicOr(&diff, 1);
    __syncthreads();
    if (threadIdx.x == 0) { out[2 * qi] = (diff == 0 && y0 == y0) ? 1.0 : 0.0; out[2 * qi + 1] = y0; }
}
__global__ __launch_bounds__(256) void k_lw_fill(double *__restrict__ v, int count, double c)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < count) v[e] = c;
}
// block `src_blk` of a 1 x (nb q) prediction row -> places a_lo .. a_hi of the query's [le][q] slot
__global__ __launch_bounds__(64) void k_lw_place(const double *__restrict__ src, int q, int lo_fit, int hi_fit, int nlv_lo, int nlv_hi, double *__restrict__ dst)
{
    for (int e = threadIdx.x; e < (nlv_hi - nlv_lo + 1) * q; e += 64) {
        const int a = nlv_lo + e / q, y = e % q;
        const int af = min(max(a, lo_fit), hi_fit);      // beyond the local model's LVs predict clamps (src/plskern.jl:228-229)
        dst[e] = src[(af - lo_fit) * q + y];
    }
}

int32_t jch_lw_generic_fits(jch_ctx *ctx, const locw_args &g, int64_t n, const int *only, int n_only)
{
    (void)n;
    const int p = g.p, q = g.q, k = g.k, m = g.m, le = g.nlv_hi - g.nlv_lo + 1;
    const int nlv_req = std::max(g.nlv_hi, 1);
    // scratch: slab k x p, Y k x q, a prediction row, the constant-y flags
    const size_t need = sizeof(double) * ((size_t)k * p + (size_t)k * q + (size_t)(le + 1) * q + 2 * (size_t)m + 64);
    JCH_TRY(jch_reserve(ctx, ctx->lw_work, need));
    double *Xl = (double *)ctx->lw_work.ptr, *Yl = Xl + (size_t)k * p, *prow = Yl + (size_t)k * q, *flags = prow + (size_t)(le + 1) * q;
    std::vector<double> hflags(2 * (size_t)m, 0.0);
    if (q == 1) {
        hipLaunchKernelGGL(k_lw_const_y, dim3((unsigned)m), dim3(256), 0, ctx->stream, g.Y, g.ind, k, m, flags);
        JCH_HIP(ctx, hipMemcpyAsync(hflags.data(), flags, sizeof(double) * 2 * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
        JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    const int nlv_cap = std::min(std::min(nlv_req, p), k);
    std::vector<double> P((size_t)p * nlv_cap), R((size_t)p * nlv_cap), W((size_t)p * nlv_cap), C((size_t)q * nlv_cap), TT(nlv_cap), xm(p), xs(p), ym(q), ys(q);
    const bool prof = ctx->profiling;
    ctx->profiling = false;                                  // (the per-query fits would recycle the event pool of the enclosing call)
    int32_t st = JCH_OK;
    const int nloop = only ? n_only : m;
    for (int ii = 0; ii < nloop && st == JCH_OK; ++ii) {
        const int i = only ? only[ii] : ii;
        double *dst = g.pred + (size_t)i * le * q;
        if (q == 1 && hflags[2 * (size_t)i] != 0.0) {
            hipLaunchKernelGGL(k_lw_fill, dim3((le * q + 255) / 256), dim3(256), 0, ctx->stream, dst, le * q, hflags[2 * (size_t)i + 1]);
            continue;
        }
        const int *ind = g.ind + (size_t)i * k;
        hipLaunchKernelGGL(k_lw_gather_x, dim3((k + 63) / 64, (p + 63) / 64), dim3(256), 0, ctx->stream, g.Xrm, g.ldr, p, ind, k, Xl);
        hipLaunchKernelGGL(k_lw_gather_y, dim3((k * q + 255) / 256), dim3(256), 0, ctx->stream, g.Y, g.ldy, q, ind, k, Yl);
        jch_pls_desc d{};
        d.n = k; d.p = p; d.q = q; d.nlv = nlv_req; d.scal = g.scal; d.dtype = JCH_F64; d.loc = JCH_LOC_DEVICE; d.inplace = 0; d.reserved = 0;
        int32_t nlv_fit = 0;
        st = jch_plskern_fit(ctx, &d, Xl, k, Yl, k, g.w + (size_t)i * k, nullptr, P.data(), R.data(), W.data(), C.data(), TT.data(), xm.data(), xs.data(),
                             ym.data(), ys.data(), nullptr, &nlv_fit);
        if (st != JCH_OK) break;
        const int hi_fit = std::min(g.nlv_hi, (int)nlv_fit), lo_fit = std::min(g.nlv_lo, hi_fit);
        st = jch_predict(ctx, JCH_LOC_DEVICE, g.Xq + i, 1, p, g.ldxq, xm.data(), xs.data(), ym.data(), ys.data(), R.data(), C.data(), q, lo_fit, hi_fit, prow, 1);
        if (st != JCH_OK) break;
        hipLaunchKernelGGL(k_lw_place, dim3(1), dim3(64), 0, ctx->stream, prow, q, lo_fit, hi_fit, g.nlv_lo, g.nlv_hi, dst);
    }
    ctx->profiling = prof;
    if (st != JCH_OK) return st;
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}
