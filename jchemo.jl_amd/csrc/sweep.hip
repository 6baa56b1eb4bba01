// K4 — the fused per-latent-variable sweep: ONE read of X per LV.
//
// Replaces the two dgemv of the reference loop (src/plskern.jl:162 `t = X r` and :167 `zp = X' D t`, with
// :163-164 `tt = t' D t` and the T column store :170; src/plsnipals.jl:78-83 for the NIPALS variant, which
// also needs c_raw = Y' D t).  The reference reads X twice per LV; t_i = x_i . r is row-local, so tt and
// zp = sum_i d_i t_i x_i are accumulated while the row is still in registers.
//
// Layout: Xr is the ROW-major working copy (ld = ldr, even; pad column zero).  One wave owns R consecutive
// rows at a time; lane l holds columns {2l, 2l+1} + 128k (k < KC) of each row as double2 (16-B loads, one
// wave-instruction = 1 KiB contiguous).  Phase 1: per-lane partial dot with the r fragment (registers),
// 64-lane butterfly -> t_i in every lane.  Phase 2: zp fragment += (d_i t_i) * x fragment.  No LDS in the
// loop; LDS only for the 4-wave combine at the end.  Per-block partials are written to `part` and summed in
// a fixed order by k_reduce_part: bit-reproducible run to run (no float atomics).
//
// Bound: HBM.  Algorithmic bytes per launch = n*ldr*8 (X) + 16 n (weights read + T column write).
#include <stdlib.h>

#include <algorithm>

#include "jch_internal.h"

typedef double v2f64 __attribute__((ext_vector_type(2)));

template <int KC, int R, bool NIPALS, bool NT, bool PF>
__global__ __launch_bounds__(256) void k_sweep(const double *__restrict__ Xr, int64_t n, int ldr,
                                               const double *__restrict__ dw, const double *__restrict__ rvec,
                                               const double *__restrict__ Yr, int qpad, double *__restrict__ tcol,
                                               double *__restrict__ part, int ldpart, const double *__restrict__ mu)
{
    extern __shared__ __attribute__((aligned(16))) double red[];  // [nw][KC*128] + [16] tt, st + [nw][64] c
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v2f64 rf[KC], zp[KC];
    bool in[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int col = 2 * lane + 128 * k;
        in[k] = col < ldr;
        rf[k] = in[k] ? *reinterpret_cast<const v2f64 *>(rvec + col) : v2f64{0.0, 0.0};
        zp[k] = v2f64{0.0, 0.0};
    }
    double tt = 0.0, cacc = 0.0, st = 0.0;
    // raw mode (mu != null, plskern-shaped fits only): the rows are NOT centred; t = x.r - mu.r, and the caller turns
    // zp_raw = sum_i d_i t_i x_i into zp = zp_raw - mu * st with st = sum_i d_i t_i (== 0 up to rounding).  Every wave
    // derives the same offset from the replicated r and means.
    double off = 0.0;
    if (mu) {
        double o = 0.0;
#pragma unroll
        for (int k = 0; k < KC; ++k)
            if (in[k]) {
                const v2f64 m2 = *reinterpret_cast<const v2f64 *>(mu + 2 * lane + 128 * k);
                o += m2.x * rf[k].x + m2.y * rf[k].y;
            }
        off = jch_wave_sum(o);
    }
    const int64_t ngroups = (n + R - 1) / R;
    const int64_t gstride = (int64_t)gridDim.x * nw;
    // PF: the rows (and weights) of the wave's next group are requested before the current group is reduced
    v2f64 xn[R][KC];
    double dwn[R];
    auto fetch = [&](int64_t gg) {
        const int64_t r0 = gg * R;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const bool live = r0 + rr < n;  // wave-uniform
            const v2f64 *rp = reinterpret_cast<const v2f64 *>(Xr + (size_t)(r0 + rr) * (size_t)ldr) + lane;
#pragma unroll
            for (int k = 0; k < KC; ++k)
                xn[rr][k] = (live && in[k]) ? (NT ? __builtin_nontemporal_load(rp + 64 * k) : rp[64 * k]) : v2f64{0.0, 0.0};
            dwn[rr] = live ? dw[r0 + rr] : 0.0;
        }
    };
    int64_t g = (int64_t)blockIdx.x * nw + wv;
    if (g < ngroups) fetch(g);
    for (; g < ngroups; g += gstride) {
        const int64_t row0 = g * R;
        v2f64 x[R][KC];
        double dwc[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            dwc[rr] = dwn[rr];
#pragma unroll
            for (int k = 0; k < KC; ++k) x[rr][k] = xn[rr][k];
        }
        if (PF && g + gstride < ngroups) fetch(g + gstride);
        double tsel = 0.0;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < KC; ++k) s += x[rr][k].x * rf[k].x + x[rr][k].y * rf[k].y;
            const double t = jch_wave_sum(s) - off;
            const bool live = row0 + rr < n;
            const double dt = live ? dwc[rr] * t : 0.0;
            tt += dt * t;
            st += dt;
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                zp[k].x += dt * x[rr][k].x;
                zp[k].y += dt * x[rr][k].y;
            }
            if (NIPALS) {
                const double yv = (live && lane < qpad) ? Yr[(size_t)(row0 + rr) * qpad + lane] : 0.0;
                cacc += dt * yv;
            }
            if (lane == rr) tsel = t;
        }
        if (lane < R && row0 + lane < n) tcol[row0 + lane] = tsel;
        if (!PF && g + gstride < ngroups) fetch(g + gstride);
    }
    // ---- combine the waves of the block in wave order, then one partial row per block
    double *zred = red;                     // [nw][KC*128]
    double *tred = red + nw * KC * 128;     // [16]
    double *cred = tred + 16;               // [nw][64]
#pragma unroll
    for (int k = 0; k < KC; ++k)
        *reinterpret_cast<v2f64 *>(zred + wv * (KC * 128) + 2 * lane + 128 * k) = zp[k];
    if (lane == 0) { tred[wv] = tt; tred[8 + wv] = st; }
    if (NIPALS) cred[wv * 64 + lane] = cacc;
    __syncthreads();
    double *prow = part + (size_t)blockIdx.x * ldpart;
    for (int c = threadIdx.x; c < ldr; c += blockDim.x) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += zred[w * (KC * 128) + c];
        prow[c] = s;
    }
    if (threadIdx.x == 0) {
        double s = 0.0, s2 = 0.0;
        for (int w = 0; w < nw; ++w) { s += tred[w]; s2 += tred[8 + w]; }
        prow[ldr] = s;
        if (!NIPALS && mu) prow[ldr + 1] = s2;
    }
    if (NIPALS && threadIdx.x < qpad) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += cred[w * 64 + threadIdx.x];
        prow[ldr + 1 + threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------- K4 v2: the same sweep with a cheaper instruction stream
// Round 2: k_sweep<4, 8> runs ONE block per CU (256 VGPRs + 131 AGPRs: 1 wave per SIMD), so nothing hides its own
// instruction latencies; per 32 KB wave-iteration it issued 96 ds_bpermute + 48 dependent adds for the eight 64-lane
// row sums, 508 v_accvgpr moves for the prefetch buffer and a saveexec/branch pair around every load (ISA listing in
// profiles/r02_sweep_isa.md) — at p = 500 its compute time per iteration was within 20 % of the memory time, which is
// why p = 1000 (half the row sums per byte) streamed 6 % faster.  This version
//   * sums R rows at once by transposing while reducing: v_permlane32_swap / v_permlane16_swap fold two rows per
//     instruction pair (lanes 0-31 keep row a, lanes 32-63 row b; then 16-lane rows), DPP row_ror / half_mirror / quad_perm
//     finish inside 16 lanes — ~32 VALU ops per 8 rows, no LDS crossbar;
//   * reads every row with unconditional loads (lanes past the row end re-read its last pair against a zero coefficient,
//     rows past n re-read row n - 1 with weight zero): straight-line load issue;
//   * rotates NBUF register buffers (loop unrolled NBUF times) instead of copying the prefetch buffer.
// Same partial-row output, same fixed-order combine, bit-reproducible; the row sums associate differently from k_sweep.
#include "rowsum_dev.h"

// Stage 1 of the fixed-order reduction WITHOUT a launch of its own (round 2; `nslice` = 0: off, k_reduce_part does it): every
// block publishes its partial row, takes a ticket in its slice, and the block that draws the last ticket of a slice sums
// the slice's rows — in exactly k_reduce_part's order (16 interleaved block streams, combined in stream order), so zt holds
// the same bits either way.  The ticket counters return to zero for the next launch.
// Visibility across the 8 XCDs (one L2 each) WITHOUT agent-scope fences: a fence makes every wave write back / invalidate
// its whole L2 (measured: +60 us per sweep launch).  Instead the partial rows are stored and re-read with agent-scope
// relaxed atomics (sc1: they go through to / come from the memory side), each thread waits for its own stores to be
// acknowledged (workgroup-scope release = s_waitcnt vmcnt(0)) before the block barrier that precedes the ticket.
__device__ __forceinline__ void jch_publish(double *p, double v, bool coherent)
{
    if (coherent) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ void jch_slice_sum_by_last_block(const double *__restrict__ part, int ldpart, int m, int *__restrict__ tickets,
                                                            double *__restrict__ zt, int ldz, int nslice)
{
    if (nslice <= 0) return;                            // grid-uniform
    __shared__ int s_last;
    const int nb = gridDim.x;
    const int per = (nb + nslice - 1) / nslice;
    const int slice = blockIdx.x / per;
    const int b0 = slice * per, b1 = min(nb, b0 + per);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this thread's published values have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(tickets + slice, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == b1 - b0 - 1;
        if (s_last) __hip_atomic_store(tickets + slice, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;                                // block-uniform
    for (int c = threadIdx.x; c < m; c += blockDim.x) {
        double sg[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) sg[g] = 0.0;
        for (int bb = b0; bb < b1; bb += 16) {
#pragma unroll
            for (int g = 0; g < 16; ++g)
                if (bb + g < b1) sg[g] += __hip_atomic_load(part + (size_t)(bb + g) * ldpart + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += sg[g];
        zt[(size_t)slice * ldz + c] = t;
    }
    if (slice == 0) {                                   // slices without blocks hold zeros (the consumer sums all JCH_ZT_SLICES)
        const int used = (nb + per - 1) / per;
        for (int e = threadIdx.x; e < (JCH_ZT_SLICES - used) * m; e += blockDim.x) {
            const int sl = used + e / m, c = e - (e / m) * m;
            zt[(size_t)sl * ldz + c] = 0.0;
        }
    }
}

// NT = false, rev alternating launch by launch (JCH_SWEEP_ALT=1, round 4): default-policy loads leave the rows in the 256 MiB
// Infinity Cache, and a sweep that walks the row groups in the OPPOSITE order of the previous one starts with what that one read
// last — measured in DESIGN.md §4 / §8.
template <int KC, int R, int NBUF, bool NT = true>
__global__ __launch_bounds__(256) void k_sweep_v2(const double *__restrict__ Xr, int64_t n, int ldr,
                                                  const double *__restrict__ dw, const double *__restrict__ rvec,
                                                  double *__restrict__ tcol, double *__restrict__ part, int ldpart,
                                                  const double *__restrict__ mu, int *__restrict__ tickets,
                                                  double *__restrict__ zt, int ldz, int nslice, int rev = 0)
{
    extern __shared__ __attribute__((aligned(16))) double red[];  // [nw][KC*128] + [16] tt, st
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v2f64 rf[KC], zp[KC];
    int coff[KC];   // element offset of this lane's pair in chunk k (clamped to the row's last pair)
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int col = 2 * lane + 128 * k;
        coff[k] = col < ldr ? col : ldr - 2;
        zp[k] = v2f64{0.0, 0.0};
    }
    const int64_t ngroups = (n + R - 1) / R;
    const int64_t gstride = (int64_t)gridDim.x * nw;
    v2f64 X[NBUF][R][KC];
    double D[NBUF][R];
    const int64_t gbase = rev ? ngroups - 1 : 0, gsign = rev ? -1 : 1;   // group gg of the walk is row group gbase + gsign gg
    auto fetch = [&](v2f64 (&xb)[R][KC], double (&db)[R], int64_t gg) {
        const int64_t r0 = (gbase + gsign * gg) * R;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int64_t row = r0 + rr < n ? r0 + rr : n - 1;         // wave-uniform clamp
            const double *rp = Xr + (size_t)row * (size_t)ldr;
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                if constexpr (NT) xb[rr][k] = __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(rp + coff[k]));
                else xb[rr][k] = *reinterpret_cast<const v2f64 *>(rp + coff[k]);
            }
            db[rr] = dw[row];
        }
    };
    // the first row groups are requested BEFORE the coefficients (r, mu): one memory latency at the head of every launch
    // instead of two (the head and tail of a launch are what does not shrink with the rows per GPU)
    int64_t g = (int64_t)blockIdx.x * nw + wv;
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b)
        if (g + b * gstride < ngroups) fetch(X[b], D[b], g + b * gstride);
#pragma unroll
    for (int k = 0; k < KC; ++k) rf[k] = 2 * lane + 128 * k < ldr ? *reinterpret_cast<const v2f64 *>(rvec + 2 * lane + 128 * k) : v2f64{0.0, 0.0};
    double tt = 0.0, st = 0.0, off = 0.0;
    if (mu) {   // raw mode: t = x.r - mu.r (see k_sweep)
        double o = 0.0;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const v2f64 m2 = *reinterpret_cast<const v2f64 *>(mu + coff[k]);
            o += m2.x * rf[k].x + m2.y * rf[k].y;                       // rf is zero past the row end
        }
        off = jch_wave_sum(o);
    }
    auto process = [&](v2f64 (&x)[R][KC], double (&dv)[R], int64_t gg) {
        const int64_t row0 = (gbase + gsign * gg) * R;
        double s[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < KC; ++k) a += x[rr][k].x * rf[k].x + x[rr][k].y * rf[k].y;
            s[rr] = a;
        }
        const double h = jch_rowsums<R>(s, lane) - off;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const double t = jch_readlane(h, jch_rowsum_lane<R>(rr));
            const double dt = row0 + rr < n ? dv[rr] * t : 0.0;
            tt += dt * t;
            st += dt;
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                zp[k].x += dt * x[rr][k].x;
                zp[k].y += dt * x[rr][k].y;
            }
        }
        // T column: lane l < R stores row row0 + l; it fetches that row's total from a lane that holds it
        {
            const int src = 16 * (((lane & 3) == 1) ? 2 : ((lane & 3) == 2) ? 1 : (lane & 3)) + (R == 8 ? 8 * ((lane >> 2) & 1) : 0);
            const double tl = __shfl(h, src, 64);
            if (lane < R && row0 + lane < n) tcol[row0 + lane] = tl;
        }
    };
    while (g < ngroups) {
#pragma unroll
        for (int b = 0; b < NBUF; ++b) {
            if (g < ngroups) {   // wave-uniform
                const int64_t ahead = g + (NBUF - 1) * gstride;
                if (ahead < ngroups) fetch(X[(b + NBUF - 1) % NBUF], D[(b + NBUF - 1) % NBUF], ahead);
                process(X[b], D[b], g);
                g += gstride;
            }
        }
    }
    // ---- combine the waves of the block in wave order, then one partial row per block
    double *zred = red;                     // [nw][KC*128]
    double *tred = red + nw * KC * 128;     // [16]
#pragma unroll
    for (int k = 0; k < KC; ++k)
        *reinterpret_cast<v2f64 *>(zred + wv * (KC * 128) + 2 * lane + 128 * k) = zp[k];
    if (lane == 0) { tred[wv] = tt; tred[8 + wv] = st; }
    __syncthreads();
    double *prow = part + (size_t)blockIdx.x * ldpart;
    for (int c = threadIdx.x; c < ldr; c += blockDim.x) {
        double a = 0.0;
        for (int w = 0; w < nw; ++w) a += zred[w * (KC * 128) + c];
        jch_publish(prow + c, a, nslice > 0);
    }
    if (threadIdx.x == 0) {
        double a = 0.0, a2 = 0.0;
        for (int w = 0; w < nw; ++w) { a += tred[w]; a2 += tred[8 + w]; }
        jch_publish(prow + ldr, a, nslice > 0);
        if (mu) jch_publish(prow + ldr + 1, a2, nslice > 0);
    }
    jch_slice_sum_by_last_block(part, ldpart, ldr + 1 + (mu ? 1 : 0), tickets, zt, ldz, nslice);
}

// Stage 1: slice s (blockIdx.y) sums its contiguous range of per-block partial rows -> zt[s][c].  Fixed order:
// 16 interleaved row streams per column (threadIdx.x >> 6), combined in stream order.
__global__ __launch_bounds__(1024) void k_reduce_part(const double *__restrict__ part, int nb, int ldpart, int m, int nslice,
                                                      double *__restrict__ zt, int ldz)
{
    __shared__ double sc[16][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int per = (nb + nslice - 1) / nslice;
    const int b0 = blockIdx.y * per, b1 = min(nb, b0 + per);
    double s = 0.0;
    if (c < m)
        for (int b = b0 + g; b < b1; b += 16) s += part[(size_t)b * ldpart + c];
    sc[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < m) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sc[k][cl];
        zt[(size_t)blockIdx.y * ldz + c] = t;
    }
}

// Stage 2 (only when the consumer wants a single vector: multi-GPU all-reduce, generic small-state kernel).
__global__ __launch_bounds__(256) void k_reduce_slices(double *__restrict__ zt, int ldz, int m, int nslice)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= m) return;
    double s = 0.0;
    for (int sl = 0; sl < nslice; ++sl) s += zt[(size_t)sl * ldz + c];
    zt[c] = s;
}

// Fixed-order two-stage sum of `nb` partial rows (ld ldpart, m columns) into one vector `out` (used by the bf16 and
// wide-row paths, which hand a single vector to the all-reduce / small-state kernel).
int32_t jch_launch_reduce_rows(jch_ctx *ctx, const double *part, int nb, int ldpart, int m, double *out)
{
    const int ldz = (m + 7) & ~7;
    JCH_TRY(jch_reserve(ctx, ctx->colpart, sizeof(double) * (size_t)JCH_ZT_SLICES * ldz + 4096 * sizeof(double)));
    double *tmp = (double *)ctx->colpart.ptr;
    const int nslice = std::max(1, std::min(JCH_ZT_SLICES, nb / 8));
    if (nslice == 1) {
        hipLaunchKernelGGL(k_reduce_part, dim3((m + 63) / 64, 1), dim3(1024), 0, ctx->stream, part, nb, ldpart, m, 1, out, ldz);
    } else {
        hipLaunchKernelGGL(k_reduce_part, dim3((m + 63) / 64, JCH_ZT_SLICES), dim3(1024), 0, ctx->stream, part, nb, ldpart, m, nslice, tmp, ldz);
        hipLaunchKernelGGL(k_reduce_slices, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, tmp, ldz, m, JCH_ZT_SLICES);
        JCH_HIP(ctx, hipMemcpyAsync(out, tmp, sizeof(double) * (size_t)m, hipMemcpyDeviceToDevice, ctx->stream));
    }
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

// Raw mode with scaling: weighted second moments about the means of the pivot-shifted row-major copy, one streaming
// pass: out[j] = sum_i d_i (z_ij - m_j)^2 for j < ldr (m = means - pivot), out[ldr + k] = sum_i d_i yc_ik^2 for k < 16
// (two-pass variance like utility.jl:314-323: the means are known).  Same wave/row layout as k_sweep.
template <int KC, int R>
__global__ __launch_bounds__(256) void k_rowvar(const double *__restrict__ Xr, int64_t n, int ldr, const double *__restrict__ dw,
                                                const double *__restrict__ mshift, const double *__restrict__ Yr, int qpad,
                                                double *__restrict__ part, int ldpart)
{
    extern __shared__ __attribute__((aligned(16))) double red[];  // [4][KC*128] + [4][64]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v2f64 mf[KC], acc[KC];
    bool in[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int col = 2 * lane + 128 * k;
        in[k] = col < ldr;
        mf[k] = in[k] ? *reinterpret_cast<const v2f64 *>(mshift + col) : v2f64{0.0, 0.0};
        acc[k] = v2f64{0.0, 0.0};
    }
    double yacc = 0.0;
    const int64_t ngroups = (n + R - 1) / R;
    const int64_t gstride = (int64_t)gridDim.x * 4;
    for (int64_t g = (int64_t)blockIdx.x * 4 + wv; g < ngroups; g += gstride) {
        const int64_t row0 = g * R;
        v2f64 x[R][KC];
        double dv[R], yv[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const bool live = row0 + rr < n;
            const v2f64 *rp = reinterpret_cast<const v2f64 *>(Xr + (size_t)(row0 + rr) * (size_t)ldr) + lane;
#pragma unroll
            for (int k = 0; k < KC; ++k) x[rr][k] = (live && in[k]) ? __builtin_nontemporal_load(rp + 64 * k) : mf[k];
            dv[rr] = live ? dw[row0 + rr] : 0.0;
            yv[rr] = (live && lane < qpad && lane < 16) ? Yr[(size_t)(row0 + rr) * qpad + lane] : 0.0;
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const double a = x[rr][k].x - mf[k].x, b = x[rr][k].y - mf[k].y;
                acc[k].x += dv[rr] * a * a;
                acc[k].y += dv[rr] * b * b;
            }
            yacc += dv[rr] * yv[rr] * yv[rr];
        }
    }
    double *zred = red, *yred = red + 4 * KC * 128;
#pragma unroll
    for (int k = 0; k < KC; ++k) *reinterpret_cast<v2f64 *>(zred + wv * (KC * 128) + 2 * lane + 128 * k) = acc[k];
    yred[wv * 64 + lane] = yacc;
    __syncthreads();
    double *prow = part + (size_t)blockIdx.x * ldpart;
    for (int c = threadIdx.x; c < ldr; c += 256)
        prow[c] = ((zred[c] + zred[KC * 128 + c]) + zred[2 * KC * 128 + c]) + zred[3 * KC * 128 + c];
    if (threadIdx.x < 16) prow[ldr + threadIdx.x] = ((yred[threadIdx.x] + yred[64 + threadIdx.x]) + yred[128 + threadIdx.x]) + yred[192 + threadIdx.x];
}

// sqrt of the reduced second moments -> divisors; out_scl[0..p) x, out_scl[p..p+q) y
__global__ __launch_bounds__(256) void k_var_to_scale(const double *__restrict__ v, int ldr, int p, int q, double *__restrict__ scl)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < p) scl[j] = sqrt(v[j]);
    else if (j < p + q) scl[j] = sqrt(v[ldr + (j - p)]);
}

// K[j][k] /= sx_j * sy_k
__global__ __launch_bounds__(256) void k_scale_K(double *__restrict__ K, int qpad, int p, int q, const double *__restrict__ scl)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= p * qpad) return;
    const int j = e / qpad, k = e - j * qpad;
    if (k < q) K[e] /= scl[j] * scl[p + k];
}

template <int KC, int R>
static int32_t launch_rowvar_t(jch_ctx *ctx, const double *Xr, int64_t n, int ldr, const double *d, const double *mshift,
                               const double *Yr, int qpad, double *out /*[ldr + 16] device*/)
{
    const size_t lds = sizeof(double) * (4 * KC * 128 + 256);
    const int64_t ngroups = (n + R - 1) / R;
    int nb = (int)std::max<int64_t>(1, std::min<int64_t>((ngroups + 3) / 4, (int64_t)ctx->cus * 3));
    const int m = ldr + 16, ldpart = (m + 7) & ~7;
    JCH_TRY(jch_reserve(ctx, ctx->part, sizeof(double) * (size_t)nb * ldpart));
    double *part = (double *)ctx->part.ptr;
    hipLaunchKernelGGL((k_rowvar<KC, R>), dim3(nb), dim3(256), lds, ctx->stream, Xr, n, ldr, d, mshift, Yr, qpad, part, ldpart);
    JCH_HIP(ctx, hipGetLastError());
    return jch_launch_reduce_rows(ctx, part, nb, ldpart, m, out);
}

// raw mode, scal = true: scl <- weighted uncorrected stds of X (about the means) and Y; K <- K / (sx sy')
int32_t jch_launch_raw_scales(jch_ctx *ctx, const double *Xr, int64_t n, int p, int ldr, const double *d, const double *mshift,
                              const double *Yr, int qpad, int q, double *tmp /*[ldr + 16] device*/, double *scl, double *K)
{
    if (ldr <= 128) JCH_TRY((launch_rowvar_t<1, 4>(ctx, Xr, n, ldr, d, mshift, Yr, qpad, tmp)));
    else if (ldr <= 256) JCH_TRY((launch_rowvar_t<2, 4>(ctx, Xr, n, ldr, d, mshift, Yr, qpad, tmp)));
    else if (ldr <= 512) JCH_TRY((launch_rowvar_t<4, 4>(ctx, Xr, n, ldr, d, mshift, Yr, qpad, tmp)));
    else if (ldr <= 1024) JCH_TRY((launch_rowvar_t<8, 2>(ctx, Xr, n, ldr, d, mshift, Yr, qpad, tmp)));
    else if (ldr <= 2048) JCH_TRY((launch_rowvar_t<16, 1>(ctx, Xr, n, ldr, d, mshift, Yr, qpad, tmp)));
    else return jch_fail(ctx, JCH_EINVAL, "internal: raw-mode scales need p <= %d", JCH_SWEEP_MAXP);
    JCH_TRY(jch_allreduce_f64(ctx, tmp, (size_t)ldr + 16));
    hipLaunchKernelGGL(k_var_to_scale, dim3((p + q + 255) / 256), dim3(256), 0, ctx->stream, tmp, ldr, p, q, scl);
    hipLaunchKernelGGL(k_scale_K, dim3((p * qpad + 255) / 256), dim3(256), 0, ctx->stream, K, qpad, p, q, scl);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

// Stage 1 of the fixed-order reduction for callers that keep the slices (bf16 path): always writes JCH_ZT_SLICES slices
// (unused ones zero); *nslice_out = 1 when only slice 0 is populated.
int32_t jch_launch_reduce_part8(jch_ctx *ctx, const double *part, int nb, int ldpart, int m, double *zt, int ldz, int *nslice_out)
{
    int nslice = std::max(1, std::min(JCH_ZT_SLICES, nb / 8));
    hipLaunchKernelGGL(k_reduce_part, dim3((m + 63) / 64, JCH_ZT_SLICES), dim3(1024), 0, ctx->stream, part, nb, ldpart, m, nslice, zt, ldz);
    JCH_HIP(ctx, hipGetLastError());
    *nslice_out = nslice > 1 ? JCH_ZT_SLICES : 1;
    return JCH_OK;
}

template <int KC, int R, bool NT = true, bool PF = false>
static int32_t launch_sweep_t(jch_ctx *ctx, const double *Xr, int64_t n, int ldr, const double *d, const double *rvec,
                              const double *Yr, int qpad, bool nipals, double *tcol, double *zt, int ldz, int max_slices,
                              int *nslice_out, int m, const double *mu)
{
    const int64_t ngroups = (n + R - 1) / R;
    // persistent-style grid: exactly the 256-thread blocks the CUs can hold (register-limited: 146 VGPRs -> 3
    // per CU at KC = 4); rows are interleaved over all waves of the grid.  (One 768-thread block per CU was tried:
    // __launch_bounds__(1024) caps the kernel at 128 VGPRs and the sweep ran 1.55x slower.)
    static int bpc_cache[2] = {0, 0};
    const size_t lds = sizeof(double) * (4 * KC * 128 + 16 + 256);
    if (bpc_cache[nipals] == 0) {
        int nblk = 0;
        hipError_t e = nipals ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_sweep<KC, R, true, NT, PF>, 256, lds)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_sweep<KC, R, false, NT, PF>, 256, lds);
        bpc_cache[nipals] = (e == hipSuccess && nblk > 0) ? nblk : 2;
    }
    const int bpc = ctx->sweep_blocks_per_cu > 0 ? ctx->sweep_blocks_per_cu : bpc_cache[nipals];
    const int wpb = 4;
    int64_t nb64 = (ngroups + wpb - 1) / wpb;
    if (nb64 > (int64_t)ctx->cus * bpc) nb64 = (int64_t)ctx->cus * bpc;
    if (nb64 < 1) nb64 = 1;
    const int nb = (int)nb64;
    const int ldpart = (m + 7) & ~7;
    JCH_TRY(jch_reserve(ctx, ctx->part, sizeof(double) * (size_t)nb * ldpart));
    double *part = (double *)ctx->part.ptr;
    (void)jch_ev(ctx);  // profiling span of the dominant kernel (begin)
    if (nipals)
        hipLaunchKernelGGL((k_sweep<KC, R, true, NT, PF>), dim3(nb), dim3(64 * wpb), lds, ctx->stream, Xr, n, ldr, d, rvec, Yr, qpad,
                           tcol, part, ldpart, mu);
    else
        hipLaunchKernelGGL((k_sweep<KC, R, false, NT, PF>), dim3(nb), dim3(64 * wpb), lds, ctx->stream, Xr, n, ldr, d, rvec, Yr, qpad,
                           tcol, part, ldpart, mu);
    (void)jch_ev(ctx);  // (end)
    int nslice = std::max(1, std::min(JCH_ZT_SLICES, nb / 8));
    hipLaunchKernelGGL(k_reduce_part, dim3((m + 63) / 64, JCH_ZT_SLICES), dim3(1024), 0, ctx->stream, part, nb, ldpart, m, nslice, zt, ldz);
    if (nslice > 1) nslice = JCH_ZT_SLICES;   // slices beyond the used ones hold zeros: the consumer sums all of them
    if (max_slices == 1 && nslice > 1) {  // consumer wants one vector (all-reduce / generic small-state kernel)
        hipLaunchKernelGGL(k_reduce_slices, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, zt, ldz, m, nslice);
        nslice = 1;
    }
    *nslice_out = nslice;
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

// zero-initialised ticket counters of the fused slice sums (the kernels leave them at zero)
int32_t jch_sweep_tickets(jch_ctx *ctx, int **out)
{
    if (!ctx->tickets.ptr) {
        JCH_TRY(jch_reserve(ctx, ctx->tickets, 256));
        JCH_HIP(ctx, hipMemsetAsync(ctx->tickets.ptr, 0, 256, ctx->stream));
    }
    *out = (int *)ctx->tickets.ptr;
    return JCH_OK;
}

template <int KC, int R, int NBUF>
static int32_t launch_sweep_v2_t(jch_ctx *ctx, const double *Xr, int64_t n, int ldr, const double *d, const double *rvec,
                                 double *tcol, double *zt, int ldz, int max_slices, int *nslice_out, int m, const double *mu,
                                 jch_part_view *pv)
{
    const int64_t ngroups = (n + R - 1) / R;
    const size_t lds = sizeof(double) * (4 * KC * 128 + 16);
    static int bpc_cache = 0;
    if (bpc_cache == 0) {
        int nblk = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_sweep_v2<KC, R, NBUF, true>, 256, lds);
        bpc_cache = (e == hipSuccess && nblk > 0) ? nblk : 1;
    }
    const int bpc = ctx->sweep_blocks_per_cu > 0 ? ctx->sweep_blocks_per_cu : bpc_cache;
    int64_t nb64 = (ngroups + 3) / 4;
    // FEWER blocks than CUs (round 4, last session; profiles/r04c_sweep_grid_scan.log): at one 4-wave block per CU the 500-column sweep
    // is faster with 13/16 of the CUs streaming — 208 of 256: 581 us per launch against 593-598 at 1e6 rows (6.91 TB/s against 6.75),
    // 293.5 / 298.9 at 500 k, 147.7 / 151.1 at 250 k — and with 7/8 of them on short shards (224: 77.6 / 79.0 us at 125 k rows; 208: 77.9);
    // 232 / 240 / 248 lie in between, 192 is slower again at 125 k rows, counts that are not a multiple of the 8 XCDs (245, 250, 253)
    // are slower than their neighbours.  The pass is bound by HBM, not by the CUs: fewer concurrent streams reach the stacks in a
    // better order.  Not so for the other streaming kernels (JCH_CUS scan: the 2000-column NIPALS passes, the bf16 sweep, K2p and
    // the local fits of lwplsr want every CU; nor for the narrow sweeps: p = 250 / 120 lose 5-10 % per 32 blocks taken away).  1000-column
    // rows (KC = 8) behave like 500-column ones: 569 us per launch at 192-208 blocks against 582.6 at 256 (n = 500 k).
    // JCH_SWEEP_NB=<blocks> overrides (=256: the former grid).
    int64_t cap = (int64_t)ctx->cus * bpc;
    if ((KC == 4 || KC == 8) && bpc == 1) cap = std::max<int64_t>(8, (((int64_t)ctx->cus * (n < (int64_t)1280 * ctx->cus ? 14 : 13)) / 16) & ~(int64_t)7);
    if (const char *e_nb = getenv("JCH_SWEEP_NB")) { const int v = atoi(e_nb); if (v > 0) cap = std::min<int64_t>(v, (int64_t)ctx->cus * bpc); }
    if (nb64 > cap) nb64 = cap;
    if (nb64 < 1) nb64 = 1;
    const int nb = (int)nb64;
    const int ldpart = (m + 7) & ~7;
    JCH_TRY(jch_reserve(ctx, ctx->part, sizeof(double) * (size_t)nb * ldpart));
    double *part = (double *)ctx->part.ptr;
    int nslice = std::max(1, std::min(JCH_ZT_SLICES, nb / 8));
    // JCH_SWEEP_FUSED_REDUCE=1: the slice sums happen in the sweep's last-arriving blocks instead of k_reduce_part.  Measured
    // and NOT the default: it removes 4 us of small-state time per LV (one launch boundary + the reduce kernel) but the tail it
    // adds to every sweep — store acknowledged at the memory side, ticket, re-read through sc1 loads: three dependent
    // memory-side round trips — costs 16 us (82.9 -> 99.0 us per launch at 125 k rows, 590 -> 601-614 at 1e6 rows); with
    // agent-scope fences instead of sc1 accesses 60 us (every wave writes back / invalidates its XCD's L2).
    const char *e_fr = getenv("JCH_SWEEP_FUSED_REDUCE");
    const bool fused = e_fr && atoi(e_fr) == 1;
    int *tickets = nullptr;
    if (fused) JCH_TRY(jch_sweep_tickets(ctx, &tickets));
    const bool timed = jch_prof_sample(ctx);
    if (timed) (void)jch_ev(ctx);  // profiling span of the dominant kernel (begin)
    // JCH_SWEEP_ALT=1: default-policy loads + the row groups walked in alternating directions, launch by launch (see k_sweep_v2)
    const char *e_alt = getenv("JCH_SWEEP_ALT");
    const int alt = e_alt ? atoi(e_alt) : 0;
    if (alt) {
        const int rev = alt == 2 ? 0 : (int)(ctx->sweep_seq++ & 1u);   // (=2: default-policy loads, one direction — A/B runs)
        hipLaunchKernelGGL((k_sweep_v2<KC, R, NBUF, false>), dim3(nb), dim3(256), lds, ctx->stream, Xr, n, ldr, d, rvec, tcol, part, ldpart, mu,
                           tickets, zt, ldz, fused ? nslice : 0, rev);
    } else {
    // The FIRST sweep of a fit runs right behind the prologue, whose last ~250 MB of the row-major copy still sit (dirty) in the memory-side
    // cache: walked forwards it takes 0.65 ms at cfg2 against 0.573 for the other 24, walked BACKWARDS — it starts with what the prologue wrote
    // last — 0.60 (profiles/r04d_first_sweep_reversed_ab.log: -50 us per cfg2 fit, -3..6 us at 125 k rows).  Same arithmetic, another order of
    // the row groups within a block's partial sums for that one LV; every fit and every rank does the same, so repeated fits and the replicated
    // state stay bit-identical.  JCH_SWEEP_FIRST_REV=0: forwards as before; =2: backwards with default-policy loads (measured SLOWER than either).
    static const int first_rev_mode = [] { const char *e = getenv("JCH_SWEEP_FIRST_REV"); return e ? atoi(e) : 1; }();
    const bool first = ctx->sweep_seq++ == 0u;
    if (first && first_rev_mode == 2)
        hipLaunchKernelGGL((k_sweep_v2<KC, R, NBUF, false>), dim3(nb), dim3(256), lds, ctx->stream, Xr, n, ldr, d, rvec, tcol, part, ldpart, mu,
                           tickets, zt, ldz, fused ? nslice : 0, 1);
    else
    hipLaunchKernelGGL((k_sweep_v2<KC, R, NBUF, true>), dim3(nb), dim3(256), lds, ctx->stream, Xr, n, ldr, d, rvec, tcol, part, ldpart, mu,
                       tickets, zt, ldz, fused ? nslice : 0, first && first_rev_mode == 1 ? 1 : 0);
    }
    if (timed) (void)jch_ev(ctx);  // (end)
    if (pv && !fused) {   // split small-state path: k_lv_spread sums the block partials itself (no k_reduce_part launch)
        pv->part = part; pv->nb = nb; pv->ldpart = ldpart;
        *nslice_out = 1;
        JCH_HIP(ctx, hipGetLastError());
        return JCH_OK;
    }
    if (!fused)
        hipLaunchKernelGGL(k_reduce_part, dim3((m + 63) / 64, JCH_ZT_SLICES), dim3(1024), 0, ctx->stream, part, nb, ldpart, m, nslice, zt, ldz);
    if (nslice > 1) nslice = JCH_ZT_SLICES;
    if (max_slices == 1 && nslice > 1) {
        hipLaunchKernelGGL(k_reduce_slices, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, zt, ldz, m, nslice);
        nslice = 1;
    }
    *nslice_out = nslice;
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

// ---------------------------------------------------------------- NIPALS sweep over a LAZILY deflated working copy
// plsnipals rewrites X after every LV (src/plsnipals.jl:86 `X .-= t * zp'`): 1 read + 1 write of X per LV on top of the two
// reads of the sweep and of the next X'DY.  The rewritten value of an element depends only on its own row and column
// (x_ij - t_i p_j), so the write can be POSTPONED: the working copy keeps the rows of `npend` LVs ago and both passes
// re-apply the pending rank-one corrections in registers, in LV order and with the very expression the eager kernel uses
// (x -= t * p: one fma each) — the same bits reach the dot products as if the rows had been stored and re-read.  The
// rows are written back every m-th LV (k_kpass_lazy, deflate.hip).  Pending loadings p_k live in LDS (one 16-B read per
// lane and chunk), the pending scores t_k[row] of a row group are ONE extra load per wave (lane l holds pending LV l / R,
// row l % R) prefetched with the rows and handed out with v_readlane.
template <int KC, int R>
__global__ __launch_bounds__(256) void k_sweep_lazy(const double *__restrict__ Xr, int64_t n, int ldr,
                                                    const double *__restrict__ dw, const double *__restrict__ wvec,
                                                    const double *__restrict__ Yr, int qpad, double *__restrict__ tcol,
                                                    double *__restrict__ part, int ldpart,
                                                    const double *__restrict__ pend_p, int npend,
                                                    const double *__restrict__ tpend, int64_t tstride)
{
    extern __shared__ __attribute__((aligned(16))) double red[];  // loop: [npend][KC*128] pending loadings; end: combine area
    constexpr int LDP = KC * 128;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v2f64 rf[KC], zp[KC];
    int coff[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int col = 2 * lane + 128 * k;
        coff[k] = col < ldr ? col : ldr - 2;
        zp[k] = v2f64{0.0, 0.0};
    }
    const int64_t ngroups = (n + R - 1) / R;
    const int64_t gstride = (int64_t)gridDim.x * 4;
    // lane -> (pending LV, row of the group) for the score prefetch
    const int pl = npend > 0 ? min(lane, npend * R - 1) : 0;
    const int64_t pk_off = (int64_t)(pl / R) * tstride;
    const int pr = pl % R;
    v2f64 xn[R][KC];
    double dwn[R], tpn = 0.0;
    auto fetch = [&](int64_t gg) {
        const int64_t r0 = gg * R;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int64_t row = r0 + rr < n ? r0 + rr : n - 1;         // wave-uniform clamp
            const double *rp = Xr + (size_t)row * (size_t)ldr;
#pragma unroll
            for (int k = 0; k < KC; ++k) xn[rr][k] = __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(rp + coff[k]));
            dwn[rr] = dw[row];
        }
        if (npend > 0) tpn = tpend[pk_off + (r0 + pr < n ? r0 + pr : n - 1)];
    };
    int64_t g = (int64_t)blockIdx.x * 4 + wv;
    if (g < ngroups) fetch(g);
    for (int e = threadIdx.x; e < npend * LDP; e += 256) red[e] = pend_p[e];
#pragma unroll
    for (int k = 0; k < KC; ++k) rf[k] = 2 * lane + 128 * k < ldr ? *reinterpret_cast<const v2f64 *>(wvec + 2 * lane + 128 * k) : v2f64{0.0, 0.0};
    double tt = 0.0, cacc = 0.0;
    __syncthreads();
    for (; g < ngroups; g += gstride) {
        const int64_t row0 = g * R;
        v2f64 x[R][KC];
        double dwc[R];
        const double tpc = tpn;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            dwc[rr] = dwn[rr];
#pragma unroll
            for (int k = 0; k < KC; ++k) x[rr][k] = xn[rr][k];
        }
        if (g + gstride < ngroups) fetch(g + gstride);
        for (int k = 0; k < npend; ++k) {       // pending deflations, oldest first
            double tk[R];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) tk[rr] = jch_readlane(tpc, k * R + rr);
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
        double tsel = 0.0;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < KC; ++k) s += x[rr][k].x * rf[k].x + x[rr][k].y * rf[k].y;
            const double t = jch_wave_sum(s);
            const bool live = row0 + rr < n;
            const double dt = live ? dwc[rr] * t : 0.0;
            tt += dt * t;
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                zp[k].x += dt * x[rr][k].x;
                zp[k].y += dt * x[rr][k].y;
            }
            const double yv = (live && lane < qpad) ? Yr[(size_t)(row0 + rr) * qpad + lane] : 0.0;
            cacc += dt * yv;
            if (lane == rr) tsel = t;
        }
        if (lane < R && row0 + lane < n) tcol[row0 + lane] = tsel;
    }
    __syncthreads();                            // the pending loadings are dead: the area becomes the combine buffer
    double *zred = red;                     // [4][KC*128]
    double *tred = red + 4 * KC * 128;      // [16]
    double *cred = tred + 16;               // [4][64]
#pragma unroll
    for (int k = 0; k < KC; ++k)
        *reinterpret_cast<v2f64 *>(zred + wv * (KC * 128) + 2 * lane + 128 * k) = zp[k];
    if (lane == 0) tred[wv] = tt;
    cred[wv * 64 + lane] = cacc;
    __syncthreads();
    double *prow = part + (size_t)blockIdx.x * ldpart;
    for (int c = threadIdx.x; c < ldr; c += 256) {
        double s = 0.0;
        for (int w = 0; w < 4; ++w) s += zred[w * (KC * 128) + c];
        prow[c] = s;
    }
    if (threadIdx.x == 0) prow[ldr] = ((tred[0] + tred[1]) + tred[2]) + tred[3];
    if (threadIdx.x < qpad) {
        double s = 0.0;
        for (int w = 0; w < 4; ++w) s += cred[w * 64 + threadIdx.x];
        prow[ldr + 1 + threadIdx.x] = s;
    }
}

template <int KC, int R>
static int32_t launch_sweep_lazy_t(jch_ctx *ctx, const double *Xr, int64_t n, int ldr, const double *d, const double *wvec,
                                   const double *Yr, int qpad, double *tcol, double *zt, int ldz, int max_slices, int *nslice_out,
                                   const double *pend_p, int npend, int npend_max, const double *tpend, int64_t tstride)
{
    const int64_t ngroups = (n + R - 1) / R;
    const size_t lds_red = sizeof(double) * (4 * KC * 128 + 16 + 256);
    const size_t lds = std::max(lds_red, sizeof(double) * (size_t)npend_max * KC * 128);
    static int bpc = 0;
    static jch_per_device_once once;
    if (!once.done(ctx->device)) {
        JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_sweep_lazy<KC, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int nblk = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_sweep_lazy<KC, R>, 256, lds_red);
        bpc = (e == hipSuccess && nblk > 0) ? nblk : 1;
        once.mark(ctx->device);
    }
    // (LDS and grid are sized for the fit's largest pending count, not the current one: one partial-row layout per fit)
    int use_bpc = ctx->sweep_blocks_per_cu > 0 ? ctx->sweep_blocks_per_cu : bpc;
    use_bpc = std::max(1, std::min<int>(use_bpc, (int)((160 * 1024) / std::max<size_t>(lds, 1))));
    int64_t nb64 = std::min<int64_t>((ngroups + 3) / 4, (int64_t)ctx->cus * use_bpc);
    const int nb = (int)std::max<int64_t>(nb64, 1);
    const int m = ldr + 1 + qpad, ldpart = (m + 7) & ~7;
    JCH_TRY(jch_reserve(ctx, ctx->part, sizeof(double) * (size_t)nb * ldpart));
    double *part = (double *)ctx->part.ptr;
    (void)jch_ev(ctx);
    hipLaunchKernelGGL((k_sweep_lazy<KC, R>), dim3(nb), dim3(256), lds, ctx->stream, Xr, n, ldr, d, wvec, Yr, qpad, tcol, part, ldpart,
                       pend_p, npend, tpend, tstride);
    (void)jch_ev(ctx);
    int nslice = std::max(1, std::min(JCH_ZT_SLICES, nb / 8));
    hipLaunchKernelGGL(k_reduce_part, dim3((m + 63) / 64, JCH_ZT_SLICES), dim3(1024), 0, ctx->stream, part, nb, ldpart, m, nslice, zt, ldz);
    if (nslice > 1) nslice = JCH_ZT_SLICES;
    if (max_slices == 1 && nslice > 1) {
        hipLaunchKernelGGL(k_reduce_slices, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, zt, ldz, m, nslice);
        nslice = 1;
    }
    *nslice_out = nslice;
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

// largest number of postponed deflations the lazy kernels can hold for this row pitch (0: not supported)
int jch_nipals_lazy_capacity(int ldr, int q)
{
    if (ldr < 2 || ldr > 2048 || q < 1 || q > 16) return 0;
    const int lds_cap = std::min((int)((144 * 1024) / (sizeof(double) * jch_nipals_lazy_pitch(ldr))), 16);   // 16 x R <= 64 lanes of scores
    const bool stream = q <= 4 && !(ldr > 1024 && q > 2);     // the envelope of the streaming pass; otherwise the MFMA tile pass
    if (stream) return lds_cap;
    if (getenv("JCH_KPASS_TILE") || ldr > 1024) return std::min(lds_cap, 8);   // LDS tile pass: JCH_TILE_MP loadings per lane
    return ldr <= 512 ? 6 : 3;                                // direct-operand MFMA pass: loadings of 2 NS columns in registers
}

int32_t jch_launch_sweep_lazy(jch_ctx *ctx, const double *Xr, int64_t n, int ldr, const double *d, const double *wvec,
                              const double *Yr, int qpad, double *tcol, double *zt, int ldz, int max_slices, int *nslice_out,
                              const double *pend_p, int npend, int npend_max, const double *tpend, int64_t tstride)
{
    if (npend < 0 || npend > npend_max || npend_max > jch_nipals_lazy_capacity(ldr, 1)) return jch_fail(ctx, JCH_EINVAL, "internal: lazy NIPALS sweep: bad pending count");
#define JCH_SL(KC, R) return launch_sweep_lazy_t<KC, R>(ctx, Xr, n, ldr, d, wvec, Yr, qpad, tcol, zt, ldz, max_slices, nslice_out, pend_p, npend, npend_max, tpend, tstride)
    if (ldr <= 128) JCH_SL(1, 4);
    if (ldr <= 256) JCH_SL(2, 4);
    if (ldr <= 512) JCH_SL(4, 4);
    if (ldr <= 1024) JCH_SL(8, 2);
    if (ldr <= 2048) JCH_SL(16, 2);
#undef JCH_SL
    return jch_fail(ctx, JCH_EINVAL, "internal: lazy NIPALS sweep needs p <= 2048");
}

int32_t jch_launch_sweep(jch_ctx *ctx, const double *Xr, int64_t n, int p, int ldr, const double *d, const double *rvec,
                         const double *Yr, int qpad, int q_extra, double *tcol, double *zt, int ldz, int max_slices,
                         int *nslice_out, const double *mu, jch_part_view *pv)
{
    (void)p;
    const bool nip = q_extra > 0;
    if (pv) *pv = jch_part_view{};
    if (mu && (nip || ldr > JCH_SWEEP_MAXP)) return jch_fail(ctx, JCH_EINVAL, "internal: raw-mode sweep is for plskern-shaped fits with p <= %d", JCH_SWEEP_MAXP);
    const int m = ldr + 1 + (nip ? qpad : (mu ? 1 : 0));
    // Default: software-prefetched kernels with 8 rows x 4 KB (32 KB) per wave in flight ahead of the 32 KB being reduced
    // (measured at cfg2 on one box: R = 4 no prefetch 0.629 ms, R = 8 no prefetch 0.633, R = 8 prefetch 0.605).
    // JCH_SWEEP_PF=0 selects the previous kernels, JCH_SWEEP_R / JCH_SWEEP_NT keep working for the p <= 512 shape.
    static int pfsel = -1, rsel = -1, ntsel = -1;
    if (pfsel < 0) { const char *e = getenv("JCH_SWEEP_PF"); pfsel = e ? atoi(e) : 1; }
    if (rsel < 0) { const char *e = getenv("JCH_SWEEP_R"); rsel = e ? atoi(e) : 0; }
    if (ntsel < 0) { const char *e = getenv("JCH_SWEEP_NT"); ntsel = e ? atoi(e) : 1; }
    // v2 kernels (k_sweep_v2: permlane/DPP row sums, branch-free loads, rotating buffers) for the plskern-shaped sweep;
    // JCH_SWEEP_V2=0 selects the round-1 kernels below, JCH_SWEEP_NBUF=3 a three-buffer rotation (read per call: A/B runs)
    {
        const char *e2 = getenv("JCH_SWEEP_V2"), *eb = getenv("JCH_SWEEP_NBUF");
        const int v2 = e2 ? atoi(e2) : 1, nbuf = eb ? atoi(eb) : 2;
        if (v2 && !nip && ldr >= 2) {
#define JCH_SWEEP_V2_CASE(KC, R, NB) return launch_sweep_v2_t<KC, R, NB>(ctx, Xr, n, ldr, d, rvec, tcol, zt, ldz, max_slices, nslice_out, m, mu, pv)
            if (ldr <= 128) JCH_SWEEP_V2_CASE(1, 8, 2);
            if (ldr <= 256) JCH_SWEEP_V2_CASE(2, 8, 2);
            if (ldr <= 512) {
                // short shards (a 1/8 or 1/4 share of cfg2): 4-row groups leave a finer last round, and a third buffer in the
                // rotation keeps two groups in flight behind the one being reduced (round 4, three A/B pairs on one box: 125 k rows
                // 79.0-80.3 against 80.2-81.6 us per launch, 250 k rows 151.4 against 153.9); JCH_SWEEP_NBUF=2 / =3 force either
                const bool shortshard = n < (int64_t)1280 * ctx->cus;
                // (measured, round 4: <4, 8, 3> — two 32 KB row groups in flight per wave, 256 + 254 registers — 589.0-589.4 us per launch
                // at n = 1e6 against 587.6 for <4, 8, 2>: the sweep is not short of bytes in flight)
                if (nbuf == 3 || (!eb && v2 != 8 && v2 != 4 && shortshard)) JCH_SWEEP_V2_CASE(4, 4, 3);
                if (v2 == 4 || (v2 != 8 && n < (int64_t)640 * ctx->cus)) JCH_SWEEP_V2_CASE(4, 4, 2);
                JCH_SWEEP_V2_CASE(4, 8, 2);
            }
            if (ldr <= 1024) { if (nbuf == 3) JCH_SWEEP_V2_CASE(8, 4, 3); JCH_SWEEP_V2_CASE(8, 4, 2); }
#undef JCH_SWEEP_V2_CASE
        }
    }
#define JCH_SWEEP_CASE(KC, R, NT, PF) return launch_sweep_t<KC, R, NT, PF>(ctx, Xr, n, ldr, d, rvec, Yr, qpad, nip, tcol, zt, ldz, max_slices, nslice_out, m, mu)
    // narrow rows keep the plain kernels: 16 rows per wave-iteration with prefetch were measured 1.5-1.8x SLOWER at
    // p = 100 / 200 (the per-row butterfly dominates); p = 1000: +6.8 % (7.06 TB/s), p = 2000: +1.7 %
    if (ldr <= 128) JCH_SWEEP_CASE(1, 4, true, false);
    if (ldr <= 256) JCH_SWEEP_CASE(2, 4, true, false);
    if (ldr <= 512) {
        if (pfsel && rsel == 2) JCH_SWEEP_CASE(4, 2, true, true);
        if (pfsel && rsel == 4) JCH_SWEEP_CASE(4, 4, true, true);
        if (pfsel) JCH_SWEEP_CASE(4, 8, true, true);
        if (rsel == 2 && !ntsel) JCH_SWEEP_CASE(4, 2, false, false);
        if (rsel == 2 && ntsel) JCH_SWEEP_CASE(4, 2, true, false);
        if (rsel == 8 && !ntsel) JCH_SWEEP_CASE(4, 8, false, false);
        if (rsel == 8 && ntsel) JCH_SWEEP_CASE(4, 8, true, false);
        if (!ntsel) JCH_SWEEP_CASE(4, 4, false, false);
        JCH_SWEEP_CASE(4, 4, true, false);
    }
    if (ldr <= 1024) { if (pfsel) JCH_SWEEP_CASE(8, 4, true, true); JCH_SWEEP_CASE(8, 2, true, false); }
    if (ldr <= 2048) { if (pfsel) JCH_SWEEP_CASE(16, 2, true, true); JCH_SWEEP_CASE(16, 1, true, false); }
#undef JCH_SWEEP_CASE
    // wider rows: two-pass fallback (sweep_wide.hip), single reduced vector
    *nslice_out = 1;
    return jch_launch_sweep_wide(ctx, Xr, n, ldr, d, rvec, Yr, qpad, nip, tcol, zt);
}
