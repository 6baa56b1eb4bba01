// K3/K5 split path (round 4) — the per-LV small-state step of the plskern-shaped loop (src/plskern.jl:150-174) as TWO kernels:
//
//   k_lv_spread  (one 256-thread block per 16 rows of the p x q kernel matrix K, i.e. per 16 columns of X; 32 blocks at p = 500)
//       everything of the step that is parallel over p and needs no more than a block-local sum:
//         zp    = sum of the sweep's per-block partial rows        (the former k_reduce_part, now for 16 columns per block)
//         c     = K' r / tt                 (K' r was formed when r was: `kr`, 16 doubles — the only division by tt is here)
//         K    <- K - zp c'                 (16 x 16 entries per block, one per thread)
//         P_a   = zp / tt,  W_a = w,  R_a = r                     (src/plskern.jl:168-174)
//       and the PARTIALS of the three quantities the second kernel needs summed over p:
//         P_i' K_new (i < a, 16 columns)  -> Z_i = their sum over the blocks (Z = P'K, the state of the r-recursion, smallstate_fast.hip) — the DIRECT
//                                          form since the end of round 4: the update Z_i - (P_i . zp) c' kept eps |K_0| of rounding error while K shrinks
//         Z_a   = zp' K_new / tt
//         G     = K_new' K_new             (16 x 16 Gram matrix: the input of the dominant-direction step, :150-155)
//   k_lv_solve   (ONE 512-thread block) sums the 32 partial Gram matrices / Z rows, finds the dominant eigenvector v of G on
//       wave 0 (lv_device.h) WHILE the other seven waves stage K_new into LDS, then w = K v / |K v|, r = (K v - R (Z v)) / |K v|
//       (:156-161) and the next LV's `kr` = K' r.
//
// Why: the single-workgroup kernel (smallstate_fast.hip) pulled K (64 KB), the 8 slice sums (32 KB) and the finished P and R
// rows (200 KB at LV 24) through ONE CU's load path before any arithmetic — 11.9 k of its 38.6 k cycles — and ran the K update,
// the P.zp dots and the Gram build on one CU.  Here that half runs on 32 CUs and the single-workgroup kernel starts at "sum the
// partial Gram matrices"; K_new and the R rows arrive during the eigenvector window, off the critical path.  Same arithmetic as
// the one-kernel path except for the ORDER of the sums over p (block partials): results agree to rounding, not to the bit; the
// replicated state stays bit-identical across ranks (every rank runs the same kernels on the same all-reduced input).
#include <stdlib.h>

#include "jch_internal.h"
#include "lv_device.h"

#define SP_NT 256
#define SP_RB 16          // partial rows per thread and load batch of k_lv_spread
#define SP_PL 32           // finished LVs whose P slice is staged in LDS for the partials of P_i' K_new (later ones: one thread per LV)
#define SP_GP 272          // doubles per block in gpart before the 16-column partials of P_i' K_new: 256 Gram entries + 16 of zp' K_new

struct lvs_args {
    jch_small s;
    int p, q, ldr, a, nlv;
    const double *part;    // [nb][ldpart] partial rows of the sweep output: zp_raw at [0, ldr), tt at [itt], st at [ist]
    int nb, ldpart, itt, ist;   // ist < 0: no st
    int mode;              // 0: centred copy (zp = zp_raw); 1: f64 raw mode (zp = zp_raw - mshift st, / scl with scaling);
                           // 2: bf16 storage mode (zp = (zp_raw - mom st) / scl)
    int nblk, gld;         // blocks of k_lv_spread, doubles per block in s.gpart
    p2p_dev px;            // P2P instantiation: the inbox transport (p2p.hip), one exchange per block
    unsigned *ctr;         // merged kernel: arrival counter of the fit (zeroed at its start); the block that brings it to `ctr_target` solves
    unsigned ctr_target;
};

#define JCH_SSTAMP(k) do { if (g.s.dbg && tid == 0) g.s.dbg[512 + 16 * (g.a + 1) + (k)] = (double)__builtin_readcyclecounter(); } while (0)

// P2P: the cross-GPU all-reduce of the sweep output happens HERE, block by block: every block pushes its 16 column sums + [tt, st]
// into its own 24-double piece of slot [parity][rank] of every rank's inbox, publishes / waits on its OWN flags and adds the ranks'
// pieces in rank order (the same bits on every rank) — the exchange of smallstate_fast.hip's fused kernel, 32 blocks wide.
// WIDE: called from the merged kernel's 512-thread blocks — threads 256 .. 511 only take part in the barriers.  Returns false when the
// exchange has bailed out (uniform over the block).
template <bool P2P, bool WIDE>
__device__ __forceinline__ bool spread_body(const lvs_args &g, const int tid)
{
    __shared__ double sc[16][18];
    __shared__ double tot[18];
    __shared__ double zpl[16], cl[16];
    __shared__ double Knl[16][17];
    __shared__ double Pl[SP_PL][17];   // P_i[j0 .. j0 + 15] of the first SP_PL finished LVs: the Z partials below run one entry per thread
    const bool act = !WIDE || tid < SP_NT;
    const int col = tid & 15, gr = (tid >> 4) & 15;
    const int p = g.p, a = g.a, j0 = blockIdx.x * 16;
    if (g.s.dbg && tid == 0 && blockIdx.x == 0) g.s.dbg[512 + 16 * (a + 1) + 5] = (double)__builtin_readcyclecounter();
    // ---- every load this block needs is issued here (one trip to L2)
    const int jrow = min(j0 + gr, p - 1);
    const double kold = g.s.K[(size_t)jrow * 16 + col];
    const int jc = j0 + col, jcc = min(jc, p - 1);
    double krv = 0.0, wv = 0.0, rv = 0.0, msh = 0.0, scl = 1.0;
    if (tid < 16) {
        krv = g.s.kr[tid];
        wv = g.s.w[jcc]; rv = g.s.r[jcc];
        if (g.mode == 1) { msh = g.s.mshift[jcc]; if (g.s.rs) scl = g.s.scl[jcc]; }
        else if (g.mode == 2) { msh = g.s.mom[jcc]; scl = g.s.scl[jcc]; }
    }
    double pre[16];     // P_i[j0 .. j0 + 15] of the finished LV i = tid - 16 (threads 16 .. 16 + a)
    const int ipre = act ? tid - 16 : -1;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) pre[jj] = (ipre >= 0 && ipre < a) ? g.s.P[(size_t)ipre * p + min(j0 + jj, p - 1)] : 0.0;
    // sums of the partial rows: thread (gr, col) adds rows gr, gr + 16, ... of column j0 + col; fixed order
    const int cidx = min(jc, g.ldr - 1);
    double acc = 0.0, acc2 = 0.0;
    const int xidx = col == 0 ? g.itt : (g.ist >= 0 ? g.ist : g.itt);
    // (16 rows per thread and trip: the 256 partial rows of a full-chip sweep in ONE round trip to L2 instead of two dependent ones)
    for (int b0 = act ? gr : g.nb; b0 < g.nb; b0 += 16 * SP_RB) {
        double v[SP_RB], x[SP_RB];
#pragma unroll
        for (int u = 0; u < SP_RB; ++u) {
            const size_t row = (size_t)min(b0 + 16 * u, g.nb - 1) * g.ldpart;
            v[u] = g.part[row + cidx];
            x[u] = col < 2 ? g.part[row + xidx] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < SP_RB; ++u) {
            const bool live = b0 + 16 * u < g.nb;
            acc += live ? v[u] : 0.0;
            acc2 += live ? x[u] : 0.0;
        }
    }
    if (act) {
        sc[gr][col] = acc;
        if (col < 2) sc[gr][16 + col] = acc2;
    }
    __syncthreads();
    if (tid < 18) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sc[k][tid];
        tot[tid] = t;
    }
    __syncthreads();
    if constexpr (P2P) {
        __shared__ int bail;
        const int par = (int)(g.px.epoch & 1ull), blk = blockIdx.x;
        char *mine = g.px.peer[g.px.rank];
        if (tid == 0) bail = __hip_atomic_load(p2p_status(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
        __syncthreads();
        if (bail) return false;
        const long long ts0 = blk == 0 ? p2p_stat_begin(g.px, tid) : 0;
        if (tid < 18) {
            const double v = tot[tid];
            for (int r = 0; r < g.px.nranks; ++r) p2p_slot(g.px.peer[r], par, g.px.rank, g.px.nranks, g.px.cap)[blk * 24 + tid] = v;
        }
        __threadfence_system();
        __syncthreads();
        p2p_publish_and_wait_block(g.px, tid, blk);
        __syncthreads();
        if (tid == 0) bail = __hip_atomic_load(p2p_status(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
        __syncthreads();
        if (bail) return false;
        if (tid < 18) {
            double s = 0.0;
            for (int r = 0; r < g.px.nranks; ++r) s += p2p_load_slot(p2p_slot(mine, par, r, g.px.nranks, g.px.cap) + blk * 24 + tid);
            tot[tid] = s;
        }
        if (blk == 0) p2p_stat_end(g.px, tid, ts0);
        __syncthreads();
    }
    if (g.s.dbg && tid == 0 && blockIdx.x == 0) g.s.dbg[512 + 16 * (a + 1) + 6] = (double)__builtin_readcyclecounter();
    const double tt = tot[16], st = g.ist >= 0 ? tot[17] : 0.0;
    if (tid < 16) {
        double zp = tot[tid];
        if (g.mode == 1) zp = g.s.rs ? (zp - msh * st) / scl : zp - msh * st;
        else if (g.mode == 2) zp = (zp - msh * st) / scl;
        if (jc >= p) zp = 0.0;
        zpl[tid] = zp;
        const double c = tid < g.q ? krv / tt : 0.0;
        cl[tid] = c;
        if (jc < p) {
            g.s.P[(size_t)a * p + jc] = zp / tt;
            g.s.W[(size_t)a * p + jc] = wv;
            g.s.R[(size_t)a * p + jc] = rv;
        }
        if (blockIdx.x == 0) {
            if (tid < g.q) g.s.C[(size_t)a * g.q + tid] = c;
            if (tid == 0) g.s.TT[a] = tt;
        }
    }
    __syncthreads();
    if (act) {   // K <- K - zp c' : row j0 + gr, column col (pad columns stay exactly zero: c is zero there)
        const bool live = j0 + gr < p;
        const double kn = kold - zpl[gr] * cl[col];
        if (live) g.s.K[(size_t)(j0 + gr) * 16 + col] = kn;
        Knl[gr][col] = live ? kn : 0.0;
        if (ipre >= 0 && ipre < min(a, SP_PL)) {
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) Pl[ipre][jj] = pre[jj];
        }
    }
    __syncthreads();
    double *gp = g.s.gpart + (size_t)blockIdx.x * g.gld;
    if (act) {   // partial Gram matrix of this block's 16 rows of K_new
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int jj = 0; jj < 16; jj += 2) {
            s0 += Knl[jj][gr] * Knl[jj][col];
            s1 += Knl[jj + 1][gr] * Knl[jj + 1][col];
        }
        gp[gr * 16 + col] = s0 + s1;
    }
    if (tid < 16) {   // partial of zp' K_new
        double s = 0.0;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) s += zpl[jj] * Knl[jj][tid];
        gp[256 + tid] = s;
    } else if (act && ipre >= SP_PL && ipre < a) {   // partial of Z_i = P_i' K_new (the DIRECT form: see the header of this file), 16 columns
        double zi[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) zi[k] = 0.0;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
#pragma unroll
            for (int k = 0; k < 16; ++k) zi[k] += pre[jj] * Knl[jj][k];   // (rows beyond p of Knl are zero)
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) gp[SP_GP + 16 * ipre + k] = zi[k];
    }
    if (act) {   // the same partials for the first SP_PL finished LVs, one (i, column) entry per thread: same products in the same order
        const int ne = min(a, SP_PL) * 16;
        for (int e = tid; e < ne; e += SP_NT) {
            double z = 0.0;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) z += Pl[e >> 4][jj] * Knl[jj][e & 15];
            gp[SP_GP + e] = z;
        }
    }
    for (int i = ipre + (SP_NT - 16); ipre >= 0 && i < a; i += SP_NT - 16) {   // (more than 240 finished LVs)
        double zi[16];
        for (int k = 0; k < 16; ++k) zi[k] = 0.0;
        for (int jj = 0; jj < 16; ++jj) {
            const double pj = j0 + jj < p ? g.s.P[(size_t)i * p + j0 + jj] : 0.0;
            for (int k = 0; k < 16; ++k) zi[k] += pj * Knl[jj][k];
        }
        for (int k = 0; k < 16; ++k) gp[SP_GP + 16 * i + k] = zi[k];
    }
    if (g.s.dbg && tid == 0 && blockIdx.x == 0) g.s.dbg[512 + 16 * (a + 1) + 7] = (double)__builtin_readcyclecounter();
    return true;
}

template <bool P2P>
__global__ __launch_bounds__(SP_NT) void k_lv_spread(lvs_args g)
{
    spread_body<P2P, false>(g, threadIdx.x);
}

template <int QP>
__device__ __forceinline__ void solve_body(const lvs_args &g, double *lds, const int tid)
{
    const int p = g.p, q = g.q, ldr = g.ldr, a = g.a;
    const int lane = tid & 63, wv = tid >> 6;
    const int an = a + 1;                          // finished LVs (rows of P / R / Z valid in global memory, row a from k_lv_spread)
    constexpr int ldk = QP | 1, lda = QP + 2;
    double *Kl = lds;                              // [p][ldk]
    double *rnl = Kl + (size_t)p * ldk;            // [ldr]  the new r (for K' r)
    double *scratch = rnl + ldr;                   // [2 FT]
    double *cl = scratch + 2 * FT;                 // [16]
    double *vl = cl + 16;                          // [16]
    double *zal = vl + 16;                         // [16]   zp' K_new
    double *sl = zal + 16;                         // [nlv]  s_i = P_i . zp
    double *ul = sl + ((g.nlv + 1) & ~1);          // [nlv]  u = Z v
    double *Zl = ul + ((g.nlv + 1) & ~1);          // [nlv][QP]
    double *G0 = Zl + (size_t)g.nlv * QP, *A0 = G0 + QP * lda, *A1 = A0 + QP * lda, *V0 = A1 + QP * lda, *V1 = V0 + QP * lda;
    double *csl = V1 + QP * lda;                   // [2 (QP + 2)]
    JCH_SSTAMP(0);
    // ---- loads: what the eigenvector needs (the partial Gram matrices) and the few words the Z update needs; the three small ones
    // go out FIRST (behind the wait for the partials they were a second round trip).  Everything the tail needs (K_new, the R rows)
    // is requested after the barrier and arrives while wave 0 solves.
    // (Measured and rejected, round 4: NO barrier here — wave 0 summing all 256 Gram entries for itself, 128 loads per lane, and every
    // other thread its own Z entry, 33 loads each: the eigenvector started later, 11.0 k cycles against 9.6 k, and the Z updates
    // outlasted it at the late LVs; small state 0.51 -> 0.555 ms per fit.)
    const double cpre = tid < q ? g.s.C[(size_t)a * q + tid] : 0.0;
    const double tt = g.s.TT[a];
    const int nent = SP_GP;
    double gacc = 0.0;
    {
        const int e = min(tid, nent - 1);
        {   // the first 32 blocks' partials in ONE batch of loads (p <= 512: all of them)
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = g.s.gpart[(size_t)min(u, g.nblk - 1) * g.gld + e];
#pragma unroll
            for (int u = 0; u < 32; ++u) gacc += u < g.nblk ? v[u] : 0.0;
        }
        for (int b0 = 32; b0 < g.nblk; b0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = g.s.gpart[(size_t)min(b0 + u, g.nblk - 1) * g.gld + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) gacc += b0 + u < g.nblk ? v[u] : 0.0;
        }
    }
    // (every LDS word is written by exactly ONE thread before the barrier below — no zero-fill pass of its own, one barrier)
    for (int e = tid; e < 4 * QP * lda; e += FT) A0[e] = 0.0;   // A0, A1, V0, V1: zero padding
    if (tid < 16) { cl[tid] = cpre; vl[tid] = 0.0; }
    for (int e = tid; e < g.nlv; e += FT) ul[e] = 0.0;
    if (tid < 256) {
        if ((tid >> 4) < QP && (tid & 15) < QP) G0[(tid >> 4) * lda + (tid & 15)] = gacc;
        if ((tid & 15) == 0 && (tid >> 4) < QP) { G0[(tid >> 4) * lda + QP] = 0.0; G0[(tid >> 4) * lda + QP + 1] = 0.0; }
    } else if (tid < SP_GP) zal[tid - 256] = gacc;
    __syncthreads();
    // the loads the tail needs go out only NOW: a barrier waits for every outstanding load of the wave
    double rreg[32];   // R[i][tid], i < min(an, 32)   (tail)
#pragma unroll
    for (int i = 0; i < 32; ++i) rreg[i] = (i < an && tid < p) ? g.s.R[(size_t)i * p + tid] : 0.0;
    JCH_SSTAMP(1);
    if (wv == 0) {
        if (q > 1) {
            bool solved;
            if constexpr (QP == 16) solved = dominant_by_squaring_mfma16(q, lda, G0, vl, g.s.dbg ? g.s.dbg + an : nullptr,
                                                                      g.s.dbg ? g.s.dbg + 512 + 16 * (a + 1) + 9 : nullptr);
            else solved = dominant_by_squaring<QP>(q, lda, G0, A0, A1, vl, g.s.dbg ? g.s.dbg + an : nullptr);
            if (!solved) {
                for (int e = lane; e < QP * lda; e += 64) A0[e] = G0[e];
                wavesync();
                jacobi_wave(q, lda, A0, A1, V0, V1, csl, vl, g.s.dbg ? g.s.dbg + an : nullptr);
            }
        } else if (lane == 0) vl[0] = 1.0;
    } else {
        // ---- the other seven waves meanwhile: K_new into LDS (flat, coalesced), the Z rows brought up to date
        const int wt = tid - 64, WT = FT - 64, tot = p * 16;
        for (int base = 0; base < tot; base += WT * 8) {
            double kr[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) kr[i] = g.s.K[min(base + wt + WT * i, tot - 1)];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = base + wt + WT * i;
                if (e < tot && (e & 15) < QP) Kl[(e >> 4) * ldk + (e & 15)] = kr[i];
            }
        }
        // Z_i = P_i' K_new (i < a) as the sum of the blocks' partials, fixed order — NOT the update Z_i - (P_i . zp) c': that one keeps
        // rounding errors of size eps |K_0| while K shrinks, and r = (K v - R (Z v)) / |K v| then loses eps |K_0| / |K_a| (six digits of
        // T'DT = diag(TT) on a PLS1 fit whose K falls by 1e8: tools/z_recurrence_drift.py)
        for (int e = wt; e < a * QP; e += WT) {
            const size_t off = (size_t)SP_GP + 16 * (size_t)(e / QP) + (size_t)(e & (QP - 1));
            double z = 0.0;
            if ((e & (QP - 1)) < q) {   // (pad columns of K are exactly zero: no loads for them)
                double v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = g.s.gpart[(size_t)min(u, g.nblk - 1) * g.gld + off];
#pragma unroll
                for (int u = 0; u < 32; ++u) z += u < g.nblk ? v[u] : 0.0;
            }
            for (int b = 32; b < g.nblk && (e & (QP - 1)) < q; ++b) z += g.s.gpart[(size_t)b * g.gld + off];
            Zl[e] = z;
            g.s.Z[e] = z;
        }
        if (wt < QP) {                            // new row Z_a = zp' K_new / tt
            const double z = zal[wt] / tt;
            Zl[a * QP + wt] = z;
            g.s.Z[a * QP + wt] = z;
        }
    }
    __syncthreads();
    JCH_SSTAMP(2);
    if (tid < an) {   // u = Z v
        double u = 0.0;
#pragma unroll
        for (int k = 0; k < QP; ++k) u += Zl[tid * QP + k] * vl[k];
        ul[tid] = u;
    }
    for (int i = tid + FT; i < an; i += FT) {
        double u = 0.0;
        for (int k = 0; k < QP; ++k) u += Zl[i * QP + k] * vl[k];
        ul[i] = u;
    }
    // w_raw = K v ; ||w_raw||
    double wr[JCH_SWEEP_MAXP / FT];
    double ssq = 0.0;
#pragma unroll
    for (int it = 0; it < JCH_SWEEP_MAXP / FT; ++it) {
        wr[it] = 0.0;
        if (it * FT >= p) continue;
        const int j = min(tid + it * FT, p - 1);
        double wv_ = 0.0;
#pragma unroll
        for (int k = 0; k < QP; ++k) wv_ += Kl[j * ldk + k] * vl[k];
        if (tid + it * FT >= p) wv_ = 0.0;
        wr[it] = wv_;
        ssq += wv_ * wv_;
    }
    const double inv = 1.0 / sqrt(jch_block_sum<FT>(ssq, scratch));   // (its barriers also publish ul)
    JCH_SSTAMP(3);
    // w = w_raw / ||.|| ;  r = (w_raw - R (Z v)) / ||.||   ==  w - sum_i (w . P_i) R_i   (src/plskern.jl:156-161)
#pragma unroll
    for (int it = 0; it < JCH_SWEEP_MAXP / FT; ++it) {
        const int j = tid + it * FT;
        if (j < ldr) {
            double wn = 0.0, rn = 0.0;
            if (j < p) {
                wn = wr[it] * inv;
                rn = wr[it];
                double r0 = 0.0, r1 = 0.0;
                int i0 = 0;
                if (it == 0) {
#pragma unroll
                    for (int i = 0; i < 32; i += 2) {
                        r0 += rreg[i] * ul[min(i, g.nlv - 1)];          // rreg is 0 beyond an
                        r1 += rreg[i + 1] * ul[min(i + 1, g.nlv - 1)];
                    }
                    i0 = an < 32 ? an : 32;
                }
                for (int i = i0; i < an; ++i) r0 += g.s.R[(size_t)i * p + j] * ul[i];
                rn -= r0 + r1;
                rn *= inv;
            }
            g.s.w[j] = wn;
            g.s.r[j] = rn;
            if (g.s.rs) g.s.rs[j] = j < p ? rn / g.s.scl[j] : 0.0;
            rnl[j] = rn;
        }
    }
    __syncthreads();
    {   // kr = K' r : the numerators of the next LV's c (same partial order as the one-kernel path's c = K' r / tt)
        const int k = tid & 15, gg = tid >> 4;
        scratch[gg * 16 + k] = k < QP ? kcol_dot(Kl, ldk, k, rnl, gg, FT / 16, p) : 0.0;
    }
    __syncthreads();
    if (tid < 16) {
        double t = 0.0;
#pragma unroll
        for (int gg = 0; gg < FT / 16; ++gg) t += scratch[gg * 16 + tid];
        g.s.kr[tid] = tid < q ? t : 0.0;
    }
    JCH_SSTAMP(4);
}

template <int QP>
__global__ __launch_bounds__(FT) void k_lv_solve(lvs_args g)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    solve_body<QP>(g, lds, threadIdx.x);
}

// MERGED (round 4, second half; OPT-IN, JCH_LV_MERGED=1 — measured SLOWER than the two launches): one launch per LV.  Every block
// runs the p-parallel half; the block that arrives LAST at the fit's counter — no block ever waits for another — goes on as the
// single-workgroup half.  Release / acquire: every thread fences its stores at device scope, the block's thread 0 bumps the counter
// (acq_rel), and the last block fences again before it reads what the other blocks (other XCDs, other L2s) wrote.
// Results are bit-identical to the two launches (tests/test_gpu_parity.py::test_merged_small_state_kernel_...).  Measured (cfg2,
// JCH_LV_DEBUG stamps): small state + gaps 0.51 -> 0.64 ms per fit.  The boundary it removes is worth 3.7 k cycles (end of block 0's
// spread half to the solve kernel's first instruction); against that the spread half takes 11.1 k cycles instead of 7.6 k in
// 512-thread blocks that reserve the solve half's LDS, and the solve half's first phase (partials in + barrier) 13.2 k instead of
// 9.2 k behind the device-scope acquire — a kernel boundary is the CHEAPER release / acquire on this part.
template <int QP, bool P2P>
__global__ __launch_bounds__(FT) void k_lv_merged(lvs_args g)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int last;
    const int tid = threadIdx.x;
    if (!spread_body<P2P, true>(g, tid)) return;
    __threadfence();
    __syncthreads();
    if (tid == 0) last = __hip_atomic_fetch_add(g.ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == g.ctr_target;
    __syncthreads();
    if (!last) return;
    __threadfence();
    solve_body<QP>(g, lds, tid);
}

static int qp_of(int q) { return q <= 1 ? 1 : (q <= 2 ? 2 : (q <= 4 ? 4 : (q <= 8 ? 8 : 16))); }

int jch_lv_split_blocks(int p) { return (p + 15) / 16; }
int jch_lv_split_gld(int nlv) { return (SP_GP + 16 * nlv + 7) & ~7; }
// doubles of jch_small::gpart: the block partials + 8 for the merged kernel's arrival counter (jch_small::lvctr)
size_t jch_lv_split_doubles(int p, int nlv) { return (size_t)jch_lv_split_blocks(p) * jch_lv_split_gld(nlv) + 8; }
// carve the split path's buffers out of `gbuf` (jch_lv_split_doubles(p, nlv) doubles); the arrival counter starts every fit at zero
// (enqueued on the fit's stream before its first LV)
int32_t jch_lv_split_begin_fit(jch_ctx *ctx, jch_small &s, double *gbuf, int p, int nlv)
{
    s.gpart = gbuf;
    s.lvctr = reinterpret_cast<unsigned *>(gbuf + jch_lv_split_doubles(p, nlv) - 8);
    // (only the opt-in merged kernel reads the counter: the default path saves the launch)
    const char *e_mg = getenv("JCH_LV_MERGED");
    if (e_mg && atoi(e_mg) == 1) JCH_HIP(ctx, hipMemsetAsync(s.lvctr, 0, 64, ctx->stream));
    return JCH_OK;
}

size_t jch_lv_solve_lds_bytes(int p, int q, int ldr, int nlv)
{
    const int QP = qp_of(q), ldk = QP | 1, lda = QP + 2;
    return sizeof(double) * ((size_t)p * ldk + ldr + 2 * FT + 48 + 2 * (size_t)((nlv + 1) & ~1) + (size_t)nlv * QP + 5 * (size_t)QP * lda + 2 * (QP + 2) + 8);
}

// LV a: the sweep's partial rows (or the all-reduced slices) -> K_new, P_a / W_a / R_a / C_a / TT_a, then (unless it was the last
// LV) the next w, r and kr.
bool jch_lv_split_p2p_ok(const jch_ctx *ctx, int p)
{
    return jch_lv_split_blocks(p) <= P2P_MAXBLK && (size_t)jch_lv_split_blocks(p) * 24 <= ctx->p2p.cap;
}

int32_t jch_launch_lv_split(jch_ctx *ctx, const jch_small &s, int p, int q, int ldr, int a, int nlv, const double *part, int nb,
                            int ldpart, int itt, int ist, int mode, bool solve, bool fuse_p2p)
{
    if (!s.kr || !s.gpart || !part || nb < 1) return jch_fail(ctx, JCH_EINVAL, "internal: split small-state path without its buffers");
    lvs_args g;
    g.s = s; g.p = p; g.q = q; g.ldr = ldr; g.a = a; g.nlv = nlv; g.part = part; g.nb = nb; g.ldpart = ldpart; g.itt = itt; g.ist = ist;
    g.mode = mode; g.nblk = jch_lv_split_blocks(p); g.gld = jch_lv_split_gld(nlv);
    g.px = p2p_dev{};
    g.ctr = s.lvctr;
    g.ctr_target = (unsigned)(a + 1) * (unsigned)g.nblk;     // LV a is the fit's (a + 1)-th launch of the merged kernel
    // default: two launches per LV (k_lv_spread, k_lv_solve); JCH_LV_MERGED=1: one (k_lv_merged) wherever an LV has a solve half
    const char *e_mg = getenv("JCH_LV_MERGED");
    const bool merged = solve && s.lvctr && e_mg && atoi(e_mg) == 1;
    if (fuse_p2p) {
        if (!jch_lv_split_p2p_ok(ctx, p)) return jch_fail(ctx, JCH_EINVAL, "internal: per-block inbox exchange outside its envelope");
        jch_p2p_next(ctx, &g.px);
    }
    if (solve) {
        static jch_per_device_once attr_once;
        if (!attr_once.done(ctx->device)) {
#define JCH_ATTR(QP) JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_lv_solve<QP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                     JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_lv_merged<QP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
                     JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_lv_merged<QP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))
            JCH_ATTR(1); JCH_ATTR(2); JCH_ATTR(4); JCH_ATTR(8); JCH_ATTR(16);
#undef JCH_ATTR
            attr_once.mark(ctx->device);
        }
    }
    const size_t lds = jch_lv_solve_lds_bytes(p, q, ldr, nlv);
    if (merged) {
#define JCH_MERGED(QP) do { if (fuse_p2p) hipLaunchKernelGGL((k_lv_merged<QP, true>), dim3(g.nblk), dim3(FT), lds, ctx->stream, g); \
                            else hipLaunchKernelGGL((k_lv_merged<QP, false>), dim3(g.nblk), dim3(FT), lds, ctx->stream, g); } while (0)
        switch (qp_of(q)) {
        case 1: JCH_MERGED(1); break;
        case 2: JCH_MERGED(2); break;
        case 4: JCH_MERGED(4); break;
        case 8: JCH_MERGED(8); break;
        default: JCH_MERGED(16); break;
        }
#undef JCH_MERGED
        JCH_HIP(ctx, hipGetLastError());
        return JCH_OK;
    }
    if (fuse_p2p) hipLaunchKernelGGL(k_lv_spread<true>, dim3(g.nblk), dim3(SP_NT), 0, ctx->stream, g);
    else hipLaunchKernelGGL(k_lv_spread<false>, dim3(g.nblk), dim3(SP_NT), 0, ctx->stream, g);
    if (solve) {
        switch (qp_of(q)) {
        case 1: hipLaunchKernelGGL((k_lv_solve<1>), dim3(1), dim3(FT), lds, ctx->stream, g); break;
        case 2: hipLaunchKernelGGL((k_lv_solve<2>), dim3(1), dim3(FT), lds, ctx->stream, g); break;
        case 4: hipLaunchKernelGGL((k_lv_solve<4>), dim3(1), dim3(FT), lds, ctx->stream, g); break;
        case 8: hipLaunchKernelGGL((k_lv_solve<8>), dim3(1), dim3(FT), lds, ctx->stream, g); break;
        default: hipLaunchKernelGGL((k_lv_solve<16>), dim3(1), dim3(FT), lds, ctx->stream, g); break;
        }
    }
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}
