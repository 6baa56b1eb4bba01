// Screened kNN of the kNN-LWPLSR prediction path (round 4) — src/getknn.jl:29-57 (the k nearest training rows of every query in
// the score space, ties in row order), src/wdist.jl:64-75 (weights).
//
// k_knn_scan (lwplsr.hip) evaluates the exact f64 distance of every (row, query) pair on the vector pipe — 40 instructions per
// pair — and keeps a running bar per query; it is bound by the dependent chain of its trips (0.51 ms per 1000 queries at cfg5,
// 1e8 pairs).  The f64 matrix pipe runs at the vector rate on this chip, so it cannot screen; the f32 one runs at twice that and
// takes the whole inner product off the vector pipe.  This path:
//
//   pack      (once per model)  the training scores, centred on their column means, rounded to f32 and laid out in the operand
//             order of v_mfma_f32_32x32x2_f32, with two extra operand columns: |z|^2 and 1.  The queries get the matching columns
//             -2 zq ... , 1, |zq|^2, so that ONE chain of matrix instructions delivers  a_ij ~ |z_i - zq_j|^2  for a 32 x 32 tile of
//             pairs.
//   k_knn_gmin    pass 1 over all pairs: every lane keeps the minimum of a_ij over the rows of a GROUP (16 rows of each of T
//             tiles).  The k-th smallest of a query's G group minima is an upper bound of its k-th smallest a_ij (k groups hold
//             k different rows at or below it), and a tight one: with G = 4.5 k it is the ~1.13 k-th smallest.
//   k_knn_bar     that k-th smallest per query (radix descent on the bit patterns, one wave per query), widened by the error
//             bound: |a_ij - d_ij^2| <= eps_j = c (max_i |z_i|^2 + |zq_j|^2) for every row, c = (4 K + 16) 2^-23, K = dd + 2
//             operand columns — twice what K truncating f32 accumulations + the f32 rounding of the operands can lose.  A row
//             among the exact k nearest (ties included) has a_ij <= tau_j + 2 eps_j.
//   k_knn_survive pass 2: the same products, every a_ij against the bar; survivors (1.13 k + the ties + the few within 2 eps) go
//             through a wave-private LDS list to the query's candidate list.
//   k_knn_finish_screen  one workgroup per query: EXACT distances of the candidates (the expression and column order of
//             k_knn_scan: the same bits), (distance, index) order, the k nearest, the weights (the tail shared with k_knn_finish).
//             A query whose list overflowed or came up short (non-finite scores) is flagged ...
//   k_knn_generic ... and done by the exact selection of lwplsr_generic.hip (flagged queries only).
//
// Neighbours, their order, distances and weights are identical to k_knn_scan's: the screen only decides WHICH rows get an exact
// distance, and it never drops a row at or below the exact k-th distance.
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "jch_internal.h"
#include "lv_device.h"
#include "lwplsr_dev.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define KS_CCAP 2048        // candidate rows per query (global); more survivors: the query is flagged for the generic selection
#define KS_LCAP 256         // wave-private LDS list of survivors (row << 5 | query column)
#define KS_PAD 1.0e30f      // |z|^2 of the pad rows of the last tile: never below a bar, never a group minimum that counts
#define KS_MAXSLOTS 1024    // group slots per query tile: G = 2 x slots <= 2048 values per query in k_knn_bar (32 per lane)

struct ks_args {
    knn_args a;
    knn_screen sc;
    float *Qs;          // [nqt][KG][64][4] query operand
    double *nq;         // [nqt * 32] |zq - mu|^2 of the f32-rounded query (f64 sum of the rounded values)
    float *gmin;        // [nqt][nslots][64]
    float *bar;         // [nqt * 32]
    int *cnt;           // [nqt * 32]
    int *cand;          // [m][KS_CCAP]
    int *flags;         // [m]
    int nqt, nslots, T, gpw, nchunks;
    double cfac;
};

// ---------------------------------------------------------------- model-constant part
__global__ __launch_bounds__(1024) void k_ks_colmean(const double *__restrict__ Zt, int64_t ldzt, int64_t n, double *__restrict__ mu, unsigned *__restrict__ hdr)
{
    __shared__ double red[16];
    const int c = blockIdx.x, tid = threadIdx.x;
    const double *col = Zt + (size_t)c * (size_t)ldzt;
    double s = 0.0;
    for (int64_t i = tid; i < n; i += 1024) s += col[i];
    s = jch_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        t /= (double)n;
        mu[c] = (t == t && fabs(t) < 1e300) ? t : 0.0;       // (non-finite scores: the pack kernel raises hdr[1])
        if (c == 0) { hdr[0] = 0u; hdr[1] = 0u; }
    }
}

// one thread per (tile, lane): A operand of v_mfma_f32_32x32x2_f32 for k-step j is A[lane % 32][2 j + lane / 32]; four steps per 16-B load
template <int KG>
__global__ __launch_bounds__(256) void k_ks_pack_rows(const double *__restrict__ Zt, int64_t ldzt, int64_t n, int dd, const double *__restrict__ mu,
                                                      float *__restrict__ Zs, int64_t ntiles, unsigned *__restrict__ hdr)
{
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tile = gid >> 6;
    const int lane = (int)(gid & 63);
    if (tile >= ntiles) return;
    const int64_t row = tile * 32 + (lane & 31);
    const int h = lane >> 5;
    const bool live = row < n;
    float v[KG * 4];
    double nz = 0.0;
    for (int c = 0; c < dd; ++c) {
        const float f = live ? (float)(Zt[(size_t)row + (size_t)c * (size_t)ldzt] - mu[c]) : 0.0f;
        nz += (double)f * (double)f;
#pragma unroll
        for (int s = 0; s < KG * 4; ++s)
            if (c == 2 * s + h) v[s] = f;
    }
    const float nzf = live ? (float)nz : KS_PAD;
#pragma unroll
    for (int s = 0; s < KG * 4; ++s) {
        const int c = 2 * s + h;
        if (c == dd) v[s] = nzf;
        else if (c == dd + 1) v[s] = 1.0f;
        else if (c > dd + 1) v[s] = 0.0f;
    }
#pragma unroll
    for (int kg = 0; kg < KG; ++kg)
        reinterpret_cast<float4 *>(Zs)[((size_t)tile * KG + kg) * 64 + lane] = make_float4(v[4 * kg], v[4 * kg + 1], v[4 * kg + 2], v[4 * kg + 3]);
    if (live && h == 0) {
        if (!(nzf < 1.0e29f)) atomicOr(hdr + 1, 1u);              // NaN / Inf / beyond the pad value: no screen for this model
        else atomicMax(hdr, __float_as_uint(nzf));                // (non-negative floats order like their bits)
    }
}

// B operand: B[2 j + lane / 32][lane % 32]; one thread per (query tile, lane)
template <int KG>
__global__ __launch_bounds__(256) void k_ks_pack_queries(ks_args g)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int qt = gid >> 6, lane = gid & 63;
    if (qt >= g.nqt) return;
    const int j = qt * 32 + (lane & 31), h = lane >> 5, dd = g.a.dd;
    const bool live = j < g.a.m;
    float v[KG * 4];
    double nq = 0.0;
    for (int c = 0; c < dd; ++c) {
        const float f = live ? (float)(g.a.Zq[(size_t)j + (size_t)c * (size_t)g.a.ldzq] - g.sc.mu[c]) : 0.0f;
        nq += (double)f * (double)f;
#pragma unroll
        for (int s = 0; s < KG * 4; ++s)
            if (c == 2 * s + h) v[s] = -2.0f * f;
    }
    const float nqf = (float)nq;
#pragma unroll
    for (int s = 0; s < KG * 4; ++s) {
        const int c = 2 * s + h;
        if (c == dd) v[s] = 1.0f;
        else if (c == dd + 1) v[s] = nqf;
        else if (c > dd + 1) v[s] = 0.0f;
    }
#pragma unroll
    for (int kg = 0; kg < KG; ++kg)
        reinterpret_cast<float4 *>(g.Qs)[((size_t)qt * KG + kg) * 64 + lane] = make_float4(v[4 * kg], v[4 * kg + 1], v[4 * kg + 2], v[4 * kg + 3]);
    if (h == 0) g.nq[j] = (double)nqf;
}

// ---------------------------------------------------------------- the two passes over all pairs
template <int KG>
__device__ __forceinline__ void ks_load_b(const ks_args &g, int qt, int lane, float (&b)[KG * 4])
{
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        const float4 t = reinterpret_cast<const float4 *>(g.Qs)[((size_t)qt * KG + kg) * 64 + lane];
        b[4 * kg] = t.x; b[4 * kg + 1] = t.y; b[4 * kg + 2] = t.z; b[4 * kg + 3] = t.w;
    }
}
template <int KG>
__device__ __forceinline__ void ks_load_a(const float *__restrict__ Zs, int64_t tile, int lane, float4 (&a)[KG])
{
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) a[kg] = reinterpret_cast<const float4 *>(Zs)[((size_t)tile * KG + kg) * 64 + lane];
}
// a_ij of one 32 x 32 tile of pairs: acc[r] of lane l <-> row 8 (r / 4) + 4 (l / 32) + r % 4, query column l % 32
template <int KG>
__device__ __forceinline__ f32x16 ks_tile(const float4 (&a)[KG], const float (&b)[KG * 4])
{
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kg].x, b[4 * kg], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kg].y, b[4 * kg + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kg].z, b[4 * kg + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kg].w, b[4 * kg + 3], acc, 0, 0, 0);
    }
    return acc;
}

// wave item w: query tile w % nqt, row chunk w / nqt (gpw groups of T tiles); the four waves of a workgroup share the chunk
template <int KG>
__global__ __launch_bounds__(256) void k_knn_gmin(ks_args g)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (int64_t)g.nqt * g.nchunks) return;
    const int qt = (int)(w % g.nqt), chunk = (int)(w / g.nqt);
    float b[KG * 4];
    ks_load_b<KG>(g, qt, lane, b);
    for (int gi = 0; gi < g.gpw; ++gi) {
        const int slot = chunk * g.gpw + gi;
        if (slot >= g.nslots) break;
        const int64_t t0 = (int64_t)slot * g.T, t1 = min(g.sc.ntiles, t0 + g.T);
        float mn = __builtin_inff();
        float4 a[KG], an[KG];
        ks_load_a<KG>(g.sc.Zs, t0, lane, a);
        for (int64_t t = t0; t < t1; ++t) {
            ks_load_a<KG>(g.sc.Zs, min(t + 1, t1 - 1), lane, an);           // next tile's operands behind this tile's products
            const f32x16 acc = ks_tile<KG>(a, b);
#pragma unroll
            for (int r = 0; r < 16; ++r) mn = fminf(mn, acc[r]);            // (a NaN never replaces a number)
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) a[kg] = an[kg];
        }
        g.gmin[((size_t)qt * g.nslots + slot) * 64 + lane] = mn;
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
    unsigned prefix = 0u;
    int need = g.a.k;
    for (int bit = 31; bit >= 0; --bit) {                                   // (wave-uniform)
        const unsigned hi = bit == 31 ? 0u : (0xffffffffu << (bit + 1));
        int cnt0 = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) cnt0 += __popcll(__ballot((u[i] & (hi | (1u << bit))) == prefix));
        if (need > cnt0) { need -= cnt0; prefix |= 1u << bit; }
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

template <int KG>
__global__ __launch_bounds__(256) void k_knn_survive(ks_args g)
{
    __shared__ unsigned list[4][KS_LCAP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * 4 + wv;
    if (w >= (int64_t)g.nqt * g.nchunks) return;
    const int qt = (int)(w % g.nqt), chunk = (int)(w / g.nqt);
    float b[KG * 4];
    ks_load_b<KG>(g, qt, lane, b);
    const float barv = g.bar[qt * 32 + (lane & 31)];
    unsigned *mine = list[wv];
    int nl = 0;                                                             // (wave-uniform)
    const unsigned long long below = (1ull << lane) - 1ull;
    auto flush = [&]() {
        wavesync();
        for (int e = lane; e < nl; e += 64) {
            const unsigned ent = mine[e];
            const int j = qt * 32 + (int)(ent & 31u);
            const int pos = atomicAdd(&g.cnt[j], 1);
            if (pos < KS_CCAP && j < g.a.m) g.cand[(size_t)j * KS_CCAP + pos] = (int)(ent >> 5);
        }
        wavesync();
        nl = 0;
    };
    const int64_t t0 = (int64_t)chunk * g.gpw * g.T, t1 = min(g.sc.ntiles, t0 + (int64_t)g.gpw * g.T);
    if (t0 >= t1) return;
    float4 a[KG], an[KG];
    ks_load_a<KG>(g.sc.Zs, t0, lane, a);
    for (int64_t t = t0; t < t1; ++t) {
        ks_load_a<KG>(g.sc.Zs, min(t + 1, t1 - 1), lane, an);
        const f32x16 acc = ks_tile<KG>(a, b);
        const unsigned rowbase = (unsigned)(t * 32) + 4u * (unsigned)(lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool pass = acc[r] <= barv;
            const unsigned long long m = __ballot(pass);
            if (m) {                                                        // (wave-uniform; 1 register in 7 at cfg5)
                if (nl + 64 > KS_LCAP) flush();
                if (pass) mine[nl + __popcll(m & below)] = ((rowbase + 8u * (r >> 2) + (r & 3)) << 5) | (unsigned)(lane & 31);
                nl += __popcll(m);
            }
        }
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) a[kg] = an[kg];
    }
    if (nl) flush();
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
    const int tid = threadIdx.x;
    const int qi = blockIdx.x, k = g.a.k, dd = g.a.dd;
    const int c = g.cnt[qi];
    if (c > KS_CCAP || c < k) {                                   // (block-uniform) overflow, or a query / model with non-finite scores
        if (tid == 0) g.flags[qi] = 1;
        return;
    }
    if (tid == 0) g.flags[qi] = 0;
    for (int e = tid; e < dd; e += 256) zq[e] = g.a.Zq[(size_t)qi + (size_t)e * (size_t)g.a.ldzq];
    __syncthreads();
    int cap = 64;
    while (cap < c) cap <<= 1;
    const int *cd = g.cand + (size_t)qi * KS_CCAP;
    for (int e = tid; e < cap; e += 256) {
        double acc = __builtin_inf();
        int row = 0x7fffffff;
        if (e < c) {
            row = cd[e];
            acc = 0.0;
            // (the expression and column order of k_knn_scan: the same bits)
            for (int c0 = 0; c0 < dd; c0 += 8) {
                double x[8];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) x[cc] = g.a.Zt[(size_t)row + (size_t)min(c0 + cc, dd - 1) * (size_t)g.a.ldzt];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc)
                    if (c0 + cc < dd) { const double d = x[cc] - zq[c0 + cc]; acc += d * d; }
            }
        }
        key[e] = acc; idx[e] = row;
    }
    bitonic_sort_n<256>(key, idx, cap);
    for (int e = tid; e < k; e += 256) { okey[e] = key[e]; oidx[e] = idx[e]; }
    __syncthreads();
    knn_finish_tail(g.a, qi, k, key, okey, oidx, sred, smed, snn);
}

// ---------------------------------------------------------------- host
static int ks_kg(int dd) { return (dd + 2 + 7) / 8; }

// the shapes the screen takes: score space of at most 62 dimensions (8 operand-column groups), k inside the finishing kernel's
// buffers, and enough rows for the groups to give a bar the candidate list can hold the survivors of
static void ks_plan(int64_t n, int k, int64_t &ntiles, int &T, int &nslots)
{
    ntiles = (n + 31) / 32;
    const int64_t target = std::min<int64_t>(KS_MAXSLOTS, std::max<int64_t>(64, ((int64_t)5 * k + 1) / 2));
    T = (int)std::max<int64_t>(1, (ntiles + target - 1) / target);
    nslots = (int)((ntiles + T - 1) / T);
}
bool jch_knn_screen_shape_ok(int64_t n, int dd, int k)
{
    if (dd < 1 || dd > 62 || k < 1 || k > KNN_CAP - 256 || n >= ((int64_t)1 << 26)) return false;
    int64_t ntiles; int T, nslots;
    ks_plan(n, k, ntiles, T, nslots);
    // the bar is the k-th smallest of G group minima, i.e. about the (-G ln(1 - k / G))-th smallest distance: that many survivors
    // (+ 15 % and the ties) must fit the candidate list
    const double G = 2.0 * (double)(n / (32 * (int64_t)T));
    if (G < 1.25 * k) return false;
    const double est = -G * log(1.0 - (double)k / G);
    return 1.15 * est + 32.0 <= 0.9 * KS_CCAP;
}
size_t jch_knn_screen_model_bytes(int64_t n, int dd)
{
    const size_t ntiles = (size_t)((n + 31) / 32);
    return ntiles * (size_t)ks_kg(dd) * 64 * 16 + (((size_t)dd * sizeof(double) + 255) & ~(size_t)255) + 256;
}
int32_t jch_knn_screen_build(jch_ctx *ctx, const double *dZt, int64_t ldzt, int64_t n, int dd, void *mem, knn_screen *out)
{
    knn_screen sc;
    sc.KG = ks_kg(dd);
    sc.ntiles = (n + 31) / 32;
    char *b = (char *)mem;
    sc.Zs = (float *)b; b += (size_t)sc.ntiles * sc.KG * 64 * 16;
    sc.mu = (double *)b; b += ((size_t)dd * sizeof(double) + 255) & ~(size_t)255;
    sc.hdr = (unsigned *)b;
    hipLaunchKernelGGL(k_ks_colmean, dim3(dd), dim3(1024), 0, ctx->stream, dZt, ldzt, n, sc.mu, sc.hdr);
    const unsigned nb = (unsigned)((sc.ntiles * 64 + 255) / 256);
#define KS_PACK(KGv) case KGv: hipLaunchKernelGGL((k_ks_pack_rows<KGv>), dim3(nb), dim3(256), 0, ctx->stream, dZt, ldzt, n, dd, sc.mu, sc.Zs, sc.ntiles, sc.hdr); break
    switch (sc.KG) { KS_PACK(1); KS_PACK(2); KS_PACK(3); KS_PACK(4); KS_PACK(5); KS_PACK(6); KS_PACK(7); KS_PACK(8);
    default: return jch_fail(ctx, JCH_EINVAL, "internal: screened kNN: %d score dimensions", dd); }
#undef KS_PACK
    JCH_HIP(ctx, hipGetLastError());
    *out = sc;
    return JCH_OK;
}

template <int KG>
static void ks_launch_passes(jch_ctx *ctx, const ks_args &g)
{
    const int64_t waves = (int64_t)g.nqt * g.nchunks;
    const unsigned nb = (unsigned)((waves + 3) / 4);
    hipLaunchKernelGGL((k_ks_pack_queries<KG>), dim3((unsigned)((g.nqt * 64 + 255) / 256)), dim3(256), 0, ctx->stream, g);
    (void)jch_ev(ctx);
    hipLaunchKernelGGL((k_knn_gmin<KG>), dim3(nb), dim3(256), 0, ctx->stream, g);
    const int G = 2 * g.nslots, nbq = (g.nqt * 32 + 3) / 4;
    if (G <= 256) hipLaunchKernelGGL((k_knn_bar<4>), dim3(nbq), dim3(256), 0, ctx->stream, g);
    else if (G <= 512) hipLaunchKernelGGL((k_knn_bar<8>), dim3(nbq), dim3(256), 0, ctx->stream, g);
    else if (G <= 1024) hipLaunchKernelGGL((k_knn_bar<16>), dim3(nbq), dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL((k_knn_bar<32>), dim3(nbq), dim3(256), 0, ctx->stream, g);
    hipLaunchKernelGGL((k_knn_survive<KG>), dim3(nb), dim3(256), 0, ctx->stream, g);
}

int32_t jch_launch_knn_screen(jch_ctx *ctx, const knn_args &a, const knn_screen &sc, int *flags)
{
    if (!jch_knn_screen_shape_ok(a.n, a.dd, a.k) || sc.KG != ks_kg(a.dd)) return jch_fail(ctx, JCH_EINVAL, "internal: screened kNN: shape outside its envelope");
    ks_args g;
    g.a = a; g.sc = sc;
    int64_t ntiles;
    ks_plan(a.n, a.k, ntiles, g.T, g.nslots);
    g.nqt = (a.m + 31) / 32;
    // groups per wave: ~6 wave items per SIMD of the chip
    g.gpw = (int)std::max<int64_t>(1, ((int64_t)g.nslots * g.nqt) / ((int64_t)ctx->cus * 4 * 6));
    if (const char *e = getenv("JCH_KNN_SCREEN_GPW")) g.gpw = std::max(1, atoi(e));
    g.nchunks = (g.nslots + g.gpw - 1) / g.gpw;
    const int K = a.dd + 2;
    g.cfac = (4.0 * K + 16.0) * 1.1920928955078125e-07;   // x 2^-23
    const size_t mpad = (size_t)g.nqt * 32;
    const size_t b_qs = (size_t)g.nqt * sc.KG * 64 * 16, b_nq = mpad * sizeof(double), b_gmin = (size_t)g.nqt * g.nslots * 64 * sizeof(float),
                 b_bar = mpad * sizeof(float), b_cnt = mpad * sizeof(int), b_cand = (size_t)a.m * KS_CCAP * sizeof(int);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    JCH_TRY(jch_reserve(ctx, ctx->gemm_b, up(b_qs) + up(b_nq) + up(b_gmin) + up(b_bar) + up(b_cnt) + up(b_cand) + 256));
    char *b = (char *)ctx->gemm_b.ptr;
    g.Qs = (float *)b; b += up(b_qs);
    g.nq = (double *)b; b += up(b_nq);
    g.gmin = (float *)b; b += up(b_gmin);
    g.bar = (float *)b; b += up(b_bar);
    g.cnt = (int *)b; b += up(b_cnt);
    g.cand = (int *)b;
    g.flags = flags;
    switch (sc.KG) {
    case 1: ks_launch_passes<1>(ctx, g); break;
    case 2: ks_launch_passes<2>(ctx, g); break;
    case 3: ks_launch_passes<3>(ctx, g); break;
    case 4: ks_launch_passes<4>(ctx, g); break;
    case 5: ks_launch_passes<5>(ctx, g); break;
    case 6: ks_launch_passes<6>(ctx, g); break;
    case 7: ks_launch_passes<7>(ctx, g); break;
    default: ks_launch_passes<8>(ctx, g); break;
    }
    hipLaunchKernelGGL(k_knn_finish_screen, dim3((unsigned)a.m), dim3(256), sizeof(double) * (size_t)a.dd, ctx->stream, g);
    JCH_HIP(ctx, hipGetLastError());
    // the flagged queries (the exception; every workgroup of a 64-block grid looks at its queries' flags and leaves)
    knn_args ag = a;
    ag.only_flags = g.flags;
    return jch_launch_knn_generic(ctx, ag);
}
