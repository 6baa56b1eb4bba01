// K3/K5 fast path — same math as smallstate.hip (phase A: src/plskern.jl:165-174 / src/plsnipals.jl:82-93;
// phase B: src/plskern.jl:150-161 / src/plsnipals.jl:72-77), restructured for latency: this kernel sits on the
// critical path between two sweeps, once per latent variable, so every dependent memory round trip counts.
//   * every global load of the kernel is issued up front (K, zt slices, r, w, Z, and register prefetches of the
//     P rows / R column this thread will need); K (p x q) then lives in LDS.
//   * dominant eigenvector of the Gram matrix G = K'K by repeated squaring in one wave
//     (A <- A^2 / tr(A)^2, ~10-14 wave-synchronous rounds of q FMAs; monitor 1 - tr(A^2)/tr(A)^2 -> 0), with the
//     cyclic Jacobi solver as the fallback when the singular-value gap is below ~3e-5 (no convergence in 24
//     squarings).  Eigenvector error is O(eps / gap) for both, the same as LAPACK's.
//   * the r-recursion r = w - sum_i (w . P_i) R_i is evaluated as r = (K v - R (Z v)) / ||K v|| with
//     Z = P'K (a x q) kept in the small state and updated incrementally per LV:
//     Z_i <- Z_i - (P_i . zp) c' (i < a), Z_a = P_a' K_new — a dot products instead of a x q.
//   * the sweep's second-stage partial slices zt[s][.] are summed here (saves a launch on one GPU).
//   * no integer division in any loop (a runtime div/mod costs ~40 instructions on this ISA).
// Used when q <= 16 and everything fits in LDS; otherwise jch_launch_lv_update falls back to k_lv_update.
#include <stdlib.h>

#include "jch_internal.h"
#include "lv_device.h"

// diagnostic stamps (JCH_LV_DEBUG only): thread 0 records the shader clock at phase boundaries
#define JCH_STAMP(k) do { if (g.s.dbg && tid == 0) g.s.dbg[512 + 16 * (g.do_a ? a + 1 : 0) + (k)] = (double)__builtin_readcyclecounter(); } while (0)
#define JCH_STAMP_W1(k) do { if (g.s.dbg && tid == 64) g.s.dbg[512 + 16 * (g.do_a ? a + 1 : 0) + (k)] = (double)__builtin_readcyclecounter(); } while (0)


// P2P: the cross-GPU all-reduce of the sweep output is done HERE through the inbox transport (p2p.hip) instead of by a
// kernel of its own: local slice sums pushed into every rank's inbox, epoch flags, bounded wait, rank-ordered sum.
template <int QP, bool P2P>
__global__ __launch_bounds__(FT) void k_lv_update_fast(lvf_args g)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int p = g.p, q = g.q, ldr = g.ldr, a = g.a, tid = threadIdx.x;   // qpad == 16 on this path
    const int lane = tid & 63, wv = tid >> 6;
    constexpr int ldk = QP | 1, lda = QP + 2;
    double *Kl = lds;                              // [p][ldk]   columns q..QP-1 are zero
    double *ztl = Kl + (size_t)p * ldk;            // [ldr + 1 + 16]
    double *rl = ztl + (ldr + 18);                 // [ldr]
    double *wl = rl + ldr;                         // [ldr]
    double *scratch = wl + ldr;                    // [2 FT]
    double *cl = scratch + 2 * FT;                 // [16]  (zero beyond q)
    double *vl = cl + 16;                          // [16]
    double *ul = vl + 16;                          // [nlv]   u = Z v ; phase A: s_i = P_i . zp
    double *Zl = ul + ((g.nlv + 1) & ~1);          // [nlv][QP]
    double *G0 = Zl + (size_t)g.nlv * QP, *A0 = G0 + QP * lda, *A1 = A0 + QP * lda, *V0 = A1 + QP * lda, *V1 = V0 + QP * lda;
    double *csl = V1 + QP * lda;                   // [2 (QP + 2)]
    double *K = g.s.K;
    const bool needK = g.do_b || g.algo == 0;
    const int a_old = a;                                  // LVs whose P/R/Z rows are in global memory at entry
    const int anext = g.do_a ? a + 1 : a;                 // LVs finished once phase A is done
    const bool rec = g.algo == 0 && anext > 0;            // r-recursion state in use (src/plskern.jl:156-161)
    const bool recA = rec && g.do_a && a_old > 0;         // phase A has old Z rows to update
    // DEFER (round 2): the dominant-eigenvector step below occupies ONE wave for ~10 k cycles while the other seven idle, so
    // everything of phase A that the eigenvector does not depend on moves into that window and runs on waves 1..7 — the
    // global write-back of K_new and of the P / W / R columns, the dots s_i = P_i . zp with their Z_i updates, and the new
    // row Z_a = zp' K_new / tt.  (Needs both phases in one call and q > 1; otherwise the original order below.)
    // Measured (JCH_LV_DEBUG stamps, cfg2, LV 12): 34.3 k -> 30.0 k cycles from the end of the staging to the end of the kernel;
    // by LV 24 the deferred work (4 finished LVs per worker wave) outlasts the eigenvector and the gain is gone.
    const bool defer = g.algo == 0 && g.do_a && g.do_b && q > 1;
    // Workers: waves 1, 2, 3, 5, 6, 7.  Wave 4 shares its SIMD with wave 0 (the eigenvector wave) and finished the same
    // share 4 k cycles after the others, so it gets none.
    constexpr int DW = FT / 64 - 2;                       // waves doing deferred work
    const int wi = wv == 0 || wv == 4 ? -1 : (wv < 4 ? wv - 1 : wv - 2);   // worker index, -1: none
    JCH_STAMP(0);
    // ---- issue EVERY global load of this kernel up front: one memory latency instead of one per phase
    double rreg[32];   // R[i][tid], i < min(a_old, 32)                         (phase B tail)
    double preg[4][8]; // P[wv + 8 v][lane + 64 c]                              (phase A: s_i = P_i . zp)
    if (rec && g.do_b) {
#pragma unroll
        for (int i = 0; i < 32; ++i) rreg[i] = (i < a_old && tid < p) ? g.s.R[(size_t)i * p + tid] : 0.0;
    }
    if (recA) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = defer ? (wi < 0 ? -1 : wi + DW * v) : wv + (FT / 64) * v;   // (defer: waves 0 and 4 own no vector)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int j = lane + 64 * c;
                preg[v][c] = (i >= 0 && i < a_old && j < p) ? g.s.P[(size_t)i * p + j] : 0.0;
            }
        }
    }
    // (round 3) ... including the first trip of every small staging loop below: the sweep's slices, r, w, the raw-mode shift and
    // the Z rows used to be loaded where they are consumed, each a round trip of its own BEHIND the wait for K — five dependent
    // trips to L2, 15 k of the kernel's 48 k cycles.  Their first FT entries now go out here, with everything else.
    const double *zsrc = g.bf_src ? g.bf_src : g.s.zt;
    const int zld = g.bf_src ? g.bf_ld : g.ldz;
    const int mz = g.bf_src ? g.bf_ldr + 2 : ldr + 1 + (g.algo == 1 ? 16 : (g.raw_mu ? 1 : 0));
    const bool pre = !P2P && g.do_a;
    double zpre[JCH_ZT_SLICES], rpre = 0.0, wpre = 0.0, mpre = 0.0, spre = 1.0, zlpre = 0.0;
#pragma unroll
    for (int sl = 0; sl < JCH_ZT_SLICES; ++sl) zpre[sl] = 0.0;
    if (pre && tid < mz) {
        if (g.nslice == 1) zpre[0] = zsrc[tid];
        else {
#pragma unroll
            for (int sl = 0; sl < JCH_ZT_SLICES; ++sl) zpre[sl] = zsrc[(size_t)sl * zld + tid];
        }
    }
    if (pre && tid < ldr) { rpre = g.s.r[tid]; wpre = g.s.w[tid]; }
    if (pre && !g.bf_src && g.raw_mu && tid < p) { mpre = g.s.mshift[tid]; if (g.s.rs) spre = g.s.scl[tid]; }
    if (rec && tid < a_old * QP) zlpre = g.s.Z[tid];
    if (needK) {   // K is [p][16] contiguous: flat coalesced loads, 16 in flight per thread, addresses clamped
        const int tot = p * 16;
        for (int base = 0; base < tot; base += FT * 16) {
            double kr[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) kr[i] = K[min(base + tid + FT * i, tot - 1)];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = base + tid + FT * i;
                if (e < tot && (e & 15) < QP) Kl[(e >> 4) * ldk + (e & 15)] = kr[i];
            }
        }
    }
    for (int e = tid; e < 5 * QP * lda; e += FT) G0[e] = 0.0;   // G0, A0, A1, V0, V1: zero padding
    if (tid < 32) cl[tid] = 0.0;                                 // cl, vl
    for (int e = tid; e < g.nlv; e += FT) ul[e] = 0.0;           // (prefetched R rows beyond a_old multiply zeros)
    if (rec) {
        if (tid < a_old * QP) Zl[tid] = zlpre;
        for (int e = tid + FT; e < a_old * QP; e += FT) Zl[e] = g.s.Z[e];
    }
    // sweep output to reduce: f64 path: s.zt [nslice][ldz], first mz entries; bf16 storage mode (g.bf_src): the raw
    // [zp_raw (bf_ldr), tt, st] slices, turned into [zp, tt] below (zp_j = (zp_raw_j - m_j st) / s_j, bf16.hip header)
    if (P2P && g.do_a) {
        const int par = (int)(g.px.epoch & 1ull);
        char *mine = g.px.peer[g.px.rank];
        volatile int *bail_s = reinterpret_cast<volatile int *>(scratch);   // (scratch is idle during the staging phase; all of the
        // 160 KB the kernel may ask for is dynamic LDS, so no static __shared__ here)
        if (tid == 0) *bail_s = __hip_atomic_load(p2p_status(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
        __syncthreads();
        if (*bail_s) return;
        __syncthreads();
        const long long ts0 = p2p_stat_begin(g.px, tid);
        for (int c = tid; c < mz; c += FT) {
            double s = 0.0;
            if (g.nslice == 1) s = zsrc[c];
            else {
                double z[JCH_ZT_SLICES];
#pragma unroll
                for (int sl = 0; sl < JCH_ZT_SLICES; ++sl) z[sl] = zsrc[(size_t)sl * zld + c];
#pragma unroll
                for (int sl = 0; sl < JCH_ZT_SLICES; ++sl) s += z[sl];
            }
            for (int r = 0; r < g.px.nranks; ++r) p2p_slot(g.px.peer[r], par, g.px.rank, g.px.nranks, g.px.cap)[c] = s;
        }
        __threadfence_system();
        __syncthreads();
        p2p_publish_and_wait(g.px, tid);
        __syncthreads();
        if (tid == 0) *bail_s = __hip_atomic_load(p2p_status(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
        __syncthreads();
        if (*bail_s) return;
        for (int c = tid; c < mz; c += FT) {
            double s = 0.0;
            for (int r = 0; r < g.px.nranks; ++r) s += p2p_load_slot(p2p_slot(mine, par, r, g.px.nranks, g.px.cap) + c);
            ztl[c] = s;
        }
        p2p_stat_end(g.px, tid, ts0);
        for (int j = tid; j < ldr; j += FT) { rl[j] = g.s.r[j]; wl[j] = g.s.w[j]; }
    } else if (g.do_a) {
        if (tid < mz) {   // (the first FT entries: loaded up front; same order of the slice sum)
            double s = zpre[0];
            if (g.nslice != 1) {
                s = 0.0;
#pragma unroll
                for (int sl = 0; sl < JCH_ZT_SLICES; ++sl) s += zpre[sl];
            }
            ztl[tid] = s;
        }
        for (int c = tid + FT; c < mz; c += FT) {
            double s;
            if (g.nslice == 1) {
                s = zsrc[c];
            } else {   // all JCH_ZT_SLICES slices are written by k_reduce_part (unused ones hold zeros)
                double z[JCH_ZT_SLICES];
#pragma unroll
                for (int sl = 0; sl < JCH_ZT_SLICES; ++sl) z[sl] = zsrc[(size_t)sl * zld + c];
                s = 0.0;
#pragma unroll
                for (int sl = 0; sl < JCH_ZT_SLICES; ++sl) s += z[sl];
            }
            ztl[c] = s;
        }
        if (tid < ldr) { rl[tid] = rpre; wl[tid] = wpre; }
        for (int j = tid + FT; j < ldr; j += FT) { rl[j] = g.s.r[j]; wl[j] = g.s.w[j]; }
    }
    if (g.do_a && g.bf_src) {   // (each thread rewrites only the entries it reads; slot ldr is outside every j < ldr)
        __syncthreads();
        const double tt_ = ztl[g.bf_ldr], st_ = ztl[g.bf_ldr + 1];
        __syncthreads();
        for (int j = tid; j < ldr; j += FT) ztl[j] = j < p ? (ztl[j] - g.s.mom[j] * st_) / g.s.scl[j] : 0.0;
        if (tid == 0) ztl[ldr] = tt_;
    } else if (g.do_a && g.raw_mu) {   // f64 raw mode (uncentred row copy): zp = zp_raw - mu * st, st = sum_i d_i t_i at [ldr + 1]
        __syncthreads();
        const double st_ = ztl[ldr + 1];
        if (pre) {   // (first FT entries from the registers loaded up front)
            if (tid < p) ztl[tid] = g.s.rs ? (ztl[tid] - mpre * st_) / spre : ztl[tid] - mpre * st_;
            if (g.s.rs) { for (int j = tid + FT; j < p; j += FT) ztl[j] = (ztl[j] - g.s.mshift[j] * st_) / g.s.scl[j]; }
            else for (int j = tid + FT; j < p; j += FT) ztl[j] -= g.s.mshift[j] * st_;
        } else if (g.s.rs) { for (int j = tid; j < p; j += FT) ztl[j] = (ztl[j] - g.s.mshift[j] * st_) / g.s.scl[j]; }
        else for (int j = tid; j < p; j += FT) ztl[j] -= g.s.mshift[j] * st_;
    }
    __syncthreads();
    JCH_STAMP(1);
    if (g.do_a && g.tt_from_r) {   // kernel algorithm #2: zt = G r, so tt = t'Dt = r'G r = r . zp
        double s = 0.0;
        for (int j = tid; j < p; j += FT) s += rl[j] * ztl[j];
        s = jch_block_sum<FT>(s, scratch);
        if (tid == 0) ztl[ldr] = s;
        __syncthreads();
    }

    // ------------------------------------------------------------------ phase A
    if (g.do_a) {
        const double tt = ztl[ldr];
        if (g.algo == 0) {
            {   // c = K' r / tt : partials over 32 row groups
                const int k = tid & 15, gr = tid >> 4;
                scratch[gr * 16 + k] = k < QP ? kcol_dot(Kl, ldk, k, rl, gr, FT / 16, p) : 0.0;
            }
            if (recA && !defer) {   // s_i = P_i . zp for the finished LVs (wave per vector)
                for (int v = 0; wv + (FT / 64) * v < a_old; ++v) {
                    const int i = wv + (FT / 64) * v;
                    double s = 0.0;
                    if (v < 4) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const int j = lane + 64 * c;
                            double pij = 0.0;
#pragma unroll
                            for (int vv = 0; vv < 4; ++vv)
                                if (vv == v) pij = preg[vv][c];
                            s += pij * ztl[min(j, p - 1)];   // pij is 0 beyond p
                        }
                    }
                    for (int j = lane + (v < 4 ? 512 : 0); j < p; j += 64) s += g.s.P[(size_t)i * p + j] * ztl[j];
                    s = jch_wave_sum(s);
                    if (lane == 0) ul[i] = s;
                }
            }
            __syncthreads();
            JCH_STAMP(2);
            if (tid < 16) {
                double t = 0.0;
#pragma unroll
                for (int gg = 0; gg < FT / 16; ++gg) t += scratch[gg * 16 + tid];
                t /= tt;
                if (tid >= q) t = 0.0;
                cl[tid] = t;
                if (tid < q) g.s.C[(size_t)a * q + tid] = t;
            }
            __syncthreads();
            JCH_STAMP(3);
            for (int j = tid; j < p; j += FT) {   // K <- K - zp c'  (LDS + global copy); P_a, W_a, R_a
                const double zp = ztl[j];
                double kv[QP];
#pragma unroll
                for (int k = 0; k < QP; ++k) kv[k] = Kl[j * ldk + k] - zp * cl[k];
#pragma unroll
                for (int k = 0; k < QP; ++k) {
                    Kl[j * ldk + k] = kv[k];
                    if (!defer) K[(size_t)j * 16 + k] = kv[k];      // pad columns stay exactly zero (c_k = 0 there)
                }
                if (!defer) {
                    g.s.P[(size_t)a * p + j] = zp / tt;
                    g.s.W[(size_t)a * p + j] = wl[j];
                    g.s.R[(size_t)a * p + j] = rl[j];
                }
            }
            if (recA && !defer)     // Z_i <- Z_i - (P_i . zp) c'
                for (int i = wv; i < a_old; i += FT / 64)
                    if (lane < QP) Zl[i * QP + lane] -= ul[i] * cl[lane];
            __syncthreads();
            JCH_STAMP(4);
            if (g.do_b && !defer) {   // new row Z_a = P_a' K_new = (zp' K_new) / tt
                const int k = tid & 15, gr = tid >> 4;
                scratch[FT + gr * 16 + k] = k < QP ? kcol_dot(Kl, ldk, k, ztl, gr, FT / 16, p) : 0.0;
            }
        } else {
            for (int j = tid; j < ldr; j += FT) {
                const double z = j < p ? ztl[j] / tt : 0.0;
                g.s.zpc[j] = z;
                if (j < p) {
                    g.s.P[(size_t)a * p + j] = z;
                    g.s.W[(size_t)a * p + j] = wl[j];
                }
            }
            if (tid < 16) {
                const double c = tid < q ? ztl[ldr + 1 + tid] / tt : 0.0;
                g.s.zpc[ldr + tid] = c;
                if (tid < q) g.s.C[(size_t)a * q + tid] = c;
            }
            if (g.s.variant == 3) {   // OPT-IN one-pass NIPALS (JCH_NIPALS_ONE_PASS): K_{a+1} = K_a - zp_raw c_raw' / tt
                for (int e = tid; e < p * 16; e += FT) {
                    const int j = e >> 4, k = e & 15;
                    if (k < q) K[e] -= ztl[j] * (ztl[ldr + 1 + k] / tt);
                }
            }
        }
        if (tid == 0) g.s.TT[a] = tt;
    }
    if (!g.do_b) { JCH_STAMP(15); return; }

    // ------------------------------------------------------------------ phase B
    if constexpr (QP == 16) {
        // Gram G = K'K on the matrix cores: v_mfma_f64_16x16x4 with A = B^T = a 4-row slab of K (the SAME register
        // is both operands: lane l holds K[j0 + (l>>4)][l&15]); each wave takes every 8th slab, partials combined in LDS.
        typedef double v4f64 __attribute__((ext_vector_type(4)));
        // (eight slabs' operands fetched together, two accumulators: one read + one DEPENDENT product per trip was 16 round trips
        // of LDS latency + matrix-pipe latency per wave)
        v4f64 acc = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        for (int j0 = 4 * wv; j0 < p; j0 += 8 * 4 * (FT / 64)) {
            double xs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = j0 + u * 4 * (FT / 64) + (lane >> 4);
                const double x = Kl[min(row, p - 1) * ldk + (lane & 15)];
                xs[u] = row < p ? x : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[u], xs[u], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[u + 1], xs[u + 1], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc[reg] += acc1[reg];
        double *gsc = csl + 2 * (QP + 2) + 8;   // [FT/64][256]
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) gsc[wv * 256 + reg * 64 + lane] = acc[reg];
        __syncthreads();
        JCH_STAMP(5);
        if (tid < 256) {
            double t = 0.0;
#pragma unroll
            for (int w8 = 0; w8 < FT / 64; ++w8) t += gsc[w8 * 256 + tid];
            const int reg = tid >> 6, l = tid & 63;
            const int m = (l >> 4) + 4 * reg, nn = l & 15;    // D[m][n]: n = lane&15, m = (lane>>4) + 4 reg
            G0[m * lda + nn] = t;
        }
        __syncthreads();
        JCH_STAMP(6);
    } else if (q > 1) {
        // Gram G = K'K: upper entries e -> (k1, k2), rows split over FT/64 groups
        const int nent = q * (q + 1) / 2;
        const int el = tid & 63, gr = tid >> 6;
        for (int e0 = 0; e0 < nent; e0 += 64) {
            int e = e0 + el, k1 = 0;
            double s = 0.0;
            const bool act = e < nent;
            if (act) {
                while (e >= q - k1) { e -= q - k1; ++k1; }
                const int k2 = k1 + e;
                for (int j = gr; j < p; j += 8 * (FT / 64)) {
                    double x1[8], x2[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int jj = min(j + u * (FT / 64), p - 1);
                        x1[u] = Kl[jj * ldk + k1];
                        x2[u] = Kl[jj * ldk + k2];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += (j + u * (FT / 64) < p) ? x1[u] * x2[u] : 0.0;
                }
            }
            scratch[gr * 64 + el] = s;
            __syncthreads();
            JCH_STAMP(5);
            if (act && gr == 0) {
                double t = 0.0;
#pragma unroll
                for (int gg = 0; gg < FT / 64; ++gg) t += scratch[gg * 64 + el];
                const int k2 = k1 + e;
                G0[k1 * lda + k2] = t;
                G0[k2 * lda + k1] = t;
            }
            __syncthreads();
            JCH_STAMP(6);
        }
    } else {
        if (tid == 0) vl[0] = 1.0;
        __syncthreads();
    }
    if (rec && g.do_a && !defer && tid < QP) {   // finish Z_a (its partials were written before the Gram barriers)
        double t = 0.0;
#pragma unroll
        for (int gg = 0; gg < FT / 16; ++gg) t += scratch[FT + gg * 16 + tid];
        Zl[a * QP + tid] = t / ztl[ldr];
    }
    if (defer && wi >= 0) {
        // ---- work of phase A that the eigenvector does not depend on, on waves 1..DW while wave 0 solves
        const double tt = ztl[ldr];
        // write-back of K_new, flat and coalesced (one wave-instruction = 4 rows x 16 columns = 512 contiguous bytes; a
        // thread-per-row loop scatters every store over 64 lines and kept the CU's address unit busy for ~8 k cycles)
        for (int e = 64 * wi + lane; e < p * 16; e += 64 * DW) K[e] = (e & 15) < QP ? Kl[(e >> 4) * ldk + (e & 15)] : 0.0;
        for (int j = 64 * wi + lane; j < p; j += 64 * DW) {        // the new P / W / R columns
            g.s.P[(size_t)a * p + j] = ztl[j] / tt;
            g.s.W[(size_t)a * p + j] = wl[j];
            g.s.R[(size_t)a * p + j] = rl[j];
        }
        if (recA) {   // s_i = P_i . zp and Z_i <- Z_i - s_i c'  (wave per finished LV: no cross-wave step)
            for (int v = 0; wi + DW * v < a_old; ++v) {
                const int i = wi + DW * v;
                double sd = 0.0;
                if (v < 4) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const int j = lane + 64 * c;
                        double pij = 0.0;
#pragma unroll
                        for (int vv = 0; vv < 4; ++vv)
                            if (vv == v) pij = preg[vv][c];
                        sd += pij * ztl[min(j, p - 1)];   // pij is 0 beyond p
                    }
                }
                for (int j = lane + (v < 4 ? 512 : 0); j < p; j += 64) sd += g.s.P[(size_t)i * p + j] * ztl[j];
                sd = jch_wave_sum(sd);
                if (lane < QP) Zl[i * QP + lane] -= sd * cl[lane];
            }
        }
        if (rec) {   // new row Z_a = P_a' K_new = (zp' K_new) / tt : 16 columns x (4 row groups per wave) x DW waves; the DW
                     // partial rows meet in `scratch` and are added up after the barrier below by the thread that owns row a
            const int k = lane & 15;
            double zr = k < QP ? kcol_dot(Kl, ldk, k, ztl, wi * 4 + (lane >> 4), 4 * DW, p) : 0.0;
            zr += __shfl_xor(zr, 16, 64);
            zr += __shfl_xor(zr, 32, 64);
            if (lane < 16) scratch[FT + wi * 16 + lane] = zr;
        }
        JCH_STAMP_W1(7);
    }
    if (q > 1 && wv == 0) {
        bool solved;
        if constexpr (QP == 16) solved = dominant_by_squaring_mfma16(q, lda, G0, vl, g.s.dbg ? g.s.dbg + anext : nullptr,
                                                                  g.s.dbg ? g.s.dbg + 512 + 16 * (g.do_a ? a + 1 : 0) + 9 : nullptr);
        else solved = dominant_by_squaring<QP>(q, lda, G0, A0, A1, vl, g.s.dbg ? g.s.dbg + anext : nullptr);
        if (!solved) {
            for (int e = lane; e < QP * lda; e += 64) A0[e] = G0[e];
            wavesync();
            jacobi_wave(q, lda, A0, A1, V0, V1, csl, vl, g.s.dbg ? g.s.dbg + anext : nullptr);
        }
    }
    __syncthreads();
    JCH_STAMP(8);
    if (rec) {
        if (defer) {
            // the new row Z_a arrives as DW partial rows (deferred work above).  Sixteen lanes of wave 1 finish one entry each and
            // sum z_k v_k among themselves (was: ONE lane of wave 0 doing the 16 entries, 16 divisions and 16 stores in turn while
            // its wave waited — 5 k of the kernel's 45 k cycles)
            if (wv == 1) {
                double zv = 0.0;
                if (lane < QP) {
                    const double tt = ztl[ldr];
                    double z = 0.0;
#pragma unroll
                    for (int w = 0; w < DW; ++w) z += scratch[FT + w * 16 + lane];
                    z /= tt;
                    Zl[a * QP + lane] = z;
                    g.s.Z[a * QP + lane] = z;
                    zv = z * vl[lane];
                }
                zv += __shfl_xor(zv, 8, 64); zv += __shfl_xor(zv, 4, 64); zv += __shfl_xor(zv, 2, 64); zv += __shfl_xor(zv, 1, 64);
                if (lane == 0) ul[a] = zv;
            }
            if (tid < a) {   // u = Z v, finished rows
                double u = 0.0;
#pragma unroll
                for (int k = 0; k < QP; ++k) u += Zl[tid * QP + k] * vl[k];
                ul[tid] = u;
            }
        } else if (tid < anext) {   // u = Z v
            double u = 0.0;
#pragma unroll
            for (int k = 0; k < QP; ++k) u += Zl[tid * QP + k] * vl[k];
            ul[tid] = u;
        }
        for (int e = tid; e < (defer ? a : anext) * QP; e += FT) g.s.Z[e] = Zl[e];
    }
    JCH_STAMP(12);
    // w_raw = K v ; ||w_raw||
    double wr[JCH_SWEEP_MAXP / FT];
    double ssq = 0.0;
#pragma unroll
    for (int it = 0; it < JCH_SWEEP_MAXP / FT; ++it) {
        wr[it] = 0.0;
        if (it * FT >= p) continue;                    // (block-uniform: at p = 500 three of the four trips had nothing to do)
        const int j = min(tid + it * FT, p - 1);
        double wv_ = 0.0;
#pragma unroll
        for (int k = 0; k < QP; ++k) wv_ += Kl[j * ldk + k] * vl[k];
        if (tid + it * FT >= p) wv_ = 0.0;
        wr[it] = wv_;
        ssq += wv_ * wv_;
    }
    JCH_STAMP(13);
    const double inv = 1.0 / sqrt(jch_block_sum<FT>(ssq, scratch));   // (its barriers also publish ul)
    JCH_STAMP(14);
    // w = w_raw / ||.|| ;  r = (w_raw - R (Z v)) / ||.||   ==  w - sum_i (w . P_i) R_i
#pragma unroll
    for (int it = 0; it < JCH_SWEEP_MAXP / FT; ++it) {
        const int j = tid + it * FT;
        if (j < ldr) {
            double wn = 0.0, rn = 0.0;
            if (j < p) {
                wn = wr[it] * inv;
                rn = wr[it];
                if (rec) {
                    double r0 = 0.0, r1 = 0.0;
                    int i0 = 0;
                    if (it == 0) {
#pragma unroll
                        for (int i = 0; i < 32; i += 2) {
                            r0 += rreg[i] * ul[min(i, g.nlv - 1)];          // rreg is 0 beyond a_old
                            r1 += rreg[i + 1] * ul[min(i + 1, g.nlv - 1)];
                        }
                        i0 = a_old < 32 ? a_old : 32;
                    }
                    for (int i = i0; i < a_old; ++i) r0 += g.s.R[(size_t)i * p + j] * ul[i];
                    if (g.do_a) r1 += rl[j] * ul[a];     // R_a = r of the LV just finished (still in LDS)
                    rn -= r0 + r1;
                }
                rn *= inv;
            }
            g.s.w[j] = wn;
            g.s.r[j] = rn;
            if (g.s.rs) g.s.rs[j] = j < p ? rn / g.s.scl[j] : 0.0;
            if (g.s.kr && !g.do_a) rl[j] = rn;          // (split path, first call: rl is free — nothing was staged into it)
        }
    }
    if (g.s.kr && !g.do_a) {   // split small-state path (smallstate_split.hip): K' r, the numerators of LV 0's c
        __syncthreads();
        {
            const int k = tid & 15, gr = tid >> 4;
            scratch[gr * 16 + k] = k < QP ? kcol_dot(Kl, ldk, k, rl, gr, FT / 16, p) : 0.0;
        }
        __syncthreads();
        if (tid < 16) {
            double t = 0.0;
#pragma unroll
            for (int gg = 0; gg < FT / 16; ++gg) t += scratch[gg * 16 + tid];
            g.s.kr[tid] = tid < q ? t : 0.0;
        }
    }
    JCH_STAMP(15);
}

size_t jch_lv_fast_lds_bytes(int p, int q, int qpad, int ldr, int nlv)
{
    (void)qpad;
    const int QP = q <= 1 ? 1 : (q <= 2 ? 2 : (q <= 4 ? 4 : (q <= 8 ? 8 : 16)));
    const int ldk = QP | 1, lda = QP + 2;
    return sizeof(double) * ((size_t)p * ldk + (ldr + 18) + 2 * (size_t)ldr + 2 * FT + 32 + ((nlv + 1) & ~1) + (size_t)nlv * QP +
                             5 * (size_t)QP * lda + 2 * (QP + 2) + 8 + (QP == 16 ? (FT / 64) * 256 : 0));
}

int32_t jch_launch_lv_update_fast(jch_ctx *ctx, const jch_small &s, int p, int q, int qpad, int ldr, int a, int nlv, int algo,
                                  int do_a, int do_b, int nslice, int ldz, bool fuse_p2p, const double *bf_src, int bf_ld, int bf_ldr)
{
    if (qpad != 16) return jch_fail(ctx, JCH_EINVAL, "internal: fast small-state kernel needs q <= 16");
    lvf_args g;
    g.s = s; g.p = p; g.q = q; g.qpad = qpad; g.ldr = ldr; g.a = a; g.nlv = nlv; g.algo = algo;
    g.do_a = do_a; g.do_b = do_b; g.nslice = nslice; g.ldz = ldz; g.skip = 0; g.tt_from_r = s.variant == 1; g.maxit = 0; g.tol = 0.0;
    g.px = p2p_dev{};
    g.bf_src = bf_src; g.bf_ld = bf_ld; g.bf_ldr = bf_ldr; g.raw_mu = s.variant == 2;
    const bool fuse = fuse_p2p && do_a;
    if (fuse) jch_p2p_next(ctx, &g.px);
    const size_t lds = jch_lv_fast_lds_bytes(p, q, qpad, ldr, nlv);
    static jch_per_device_once attr_once;
    if (!attr_once.done(ctx->device)) {
#define JCH_ATTR(QP) do { \
        JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_lv_update_fast<QP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_lv_update_fast<QP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); } while (0)
        JCH_ATTR(1); JCH_ATTR(2); JCH_ATTR(4); JCH_ATTR(8); JCH_ATTR(16);
#undef JCH_ATTR
        attr_once.mark(ctx->device);
    }
#define JCH_LVF(QP) do { \
        if (fuse) hipLaunchKernelGGL((k_lv_update_fast<QP, true>), dim3(1), dim3(FT), lds, ctx->stream, g); \
        else hipLaunchKernelGGL((k_lv_update_fast<QP, false>), dim3(1), dim3(FT), lds, ctx->stream, g); } while (0)
    if (q <= 1) JCH_LVF(1);
    else if (q <= 2) JCH_LVF(2);
    else if (q <= 4) JCH_LVF(4);
    else if (q <= 8) JCH_LVF(8);
    else JCH_LVF(16);
#undef JCH_LVF
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}
