// Device helpers shared by the single-workgroup small-state kernels (smallstate_fast.hip, siblings.hip):
// wave-synchronous Jacobi and repeated-squaring eigen-solvers for the q x q Gram matrix, batched LDS dot products.
#pragma once
#include "jch_internal.h"
#include "p2p_dev.h"
#include "rowsum_dev.h"

#define FT 512   // threads of a single-workgroup small-state kernel

struct lvf_args {
    jch_small s;
    int p, q, qpad, ldr, a, nlv, algo, do_a, do_b, nslice, ldz, skip, tt_from_r;
    int maxit;    // plswold: inner-iteration cap (src/plswold.jl:89)
    double tol;   // plswold: convergence threshold on ||wx - w0||^2
    p2p_dev px;   // fused inbox all-reduce of the sweep output (only read by the P2P instantiations)
    const double *bf_src;   // bf16 storage mode: raw slices [nslice][bf_ld] of [zp_raw (bf_ldr), tt, st]; null otherwise
    int bf_ld, bf_ldr;
    int raw_mu;             // f64 raw mode: the sweep ran on uncentred rows; zp needs - mu * st (st at [ldr + 1])
};

__device__ __forceinline__ void wavesync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Rotation (c, s) annihilating apq.  The ANGLE only needs ~1e-8 accuracy (the off-diagonal then drops
// quadratically), but c^2 + s^2 must equal 1 to rounding or V loses orthogonality: t from the hardware
// reciprocal / rsqrt estimates (v_rcp_f64, v_rsq_f64: ~1e-8), c = rsqrt(1 + t^2) refined by one Newton step,
// s = t c.  ~15 dependent f64 ops instead of ~70 for correctly rounded sqrt + 2 divisions + rsqrt.
__device__ __forceinline__ void jacobi_rot(double app, double aqq, double apq, double &c, double &s)
{
    const double al = 0.5 * (aqq - app);
    const double x = al * al + apq * apq;
    const double h = x * __builtin_amdgcn_rsq(x);                       // sqrt(x), ~1e-8
    const double t = apq * __builtin_amdgcn_rcp(al + (al >= 0.0 ? h : -h));
    const double u = 1.0 + t * t;
    double y = __builtin_amdgcn_rsq(u);
    y = y * (1.5 - 0.5 * u * y * y);                                    // Newton: full double
    y = y * (1.5 - 0.5 * u * y * y);
    c = y;
    s = t * y;
}

// One-wave parallel cyclic Jacobi on the symmetric q x q matrix A0 (q <= 16).  Writes the dominant eigenvector
// (largest-|.| component positive) to vout[0..q).  Round-robin pairing: in round r, index x meets
// m-1 <-> r, and otherwise (2r - x) mod (m-1); every lane derives its partners arithmetically.
__device__ static void jacobi_wave(int q, int lda, double *A0, double *A1, double *V0, double *V1, double *csl,
                                   double *vout, double *dbg)
{
    const int lane = threadIdx.x & 63;
    const int m = (q + 1) & ~1, qq = q * q, mm1 = m - 1;
    int ie[4], je[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int e = lane + 64 * s;
        ie[s] = e / q;
        je[s] = e - ie[s] * q;
        if (e < qq) V0[ie[s] * lda + je[s]] = (ie[s] == je[s]) ? 1.0 : 0.0;
    }
    wavesync();
    // No integer division inside the rounds (a runtime modulo costs ~40 instructions on this ISA): every
    // round-dependent index is advanced incrementally.  pa/pb: the pair owned by this lane in step 1;
    // bi[s]/bj[s]: (2 round - i) mod (m-1), the generic partner of this lane's element row/column.
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool any_rot = false, any_big = false;
        int pa = lane < mm1 ? lane : 0, pb = lane == 0 ? 0 : (lane < mm1 ? mm1 - lane : 0);   // round 0
        int bi[4], bj[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bi[s] = (ie[s] == 0 || ie[s] >= mm1) ? 0 : mm1 - ie[s];
            bj[s] = (je[s] == 0 || je[s] >= mm1) ? 0 : mm1 - je[s];
        }
        for (int round = 0; round < mm1; ++round) {
            bool rot = false, big = false;
            if (lane < m / 2) {
                int a = lane == 0 ? mm1 : pa, b = lane == 0 ? round : pb;
                if (a > b) { const int t = a; a = b; b = t; }
                double c = 1.0, s = 0.0;
                if (b < q) {
                    const double app = A0[a * lda + a], aqq = A0[b * lda + b], apq = A0[a * lda + b];
                    const double lim = fabs(app * aqq), b2 = apq * apq;
                    if (b2 > 1e-300 && b2 > 1e-34 * lim) {
                        jacobi_rot(app, aqq, apq, c, s);
                        rot = true;
                        big = b2 > 1e-16 * lim;
                    }
                    csl[2 * a] = c; csl[2 * a + 1] = s;
                    csl[2 * b] = c; csl[2 * b + 1] = -s;
                } else if (a < q) {
                    csl[2 * a] = 1.0; csl[2 * a + 1] = 0.0;
                }
            }
            any_rot |= __ballot(rot) != 0ull;
            any_big |= __ballot(big) != 0ull;
            wavesync();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (lane + 64 * s < qq) {
                    const int i = ie[s], j = je[s];
                    int ip = i == mm1 ? round : (i == round ? mm1 : bi[s]);
                    int jp = j == mm1 ? round : (j == round ? mm1 : bj[s]);
                    if (ip >= q) ip = i;   // bye (odd q): identity rotation recorded for i
                    if (jp >= q) jp = j;
                    const double ci = csl[2 * i], si = csl[2 * i + 1], cj = csl[2 * j], sj = csl[2 * j + 1];
                    const double rij = ci * A0[i * lda + j] - si * A0[ip * lda + j];
                    const double rijp = ci * A0[i * lda + jp] - si * A0[ip * lda + jp];
                    A1[i * lda + j] = cj * rij - sj * rijp;
                    V1[i * lda + j] = cj * V0[i * lda + j] - sj * V0[i * lda + jp];
                }
            }
            wavesync();
            double *t = A0; A0 = A1; A1 = t;
            t = V0; V0 = V1; V1 = t;
            // advance to round + 1:  (round + lane) mod (m-1), (round - lane) mod (m-1), (2 round - i) mod (m-1)
            pa = pa + 1 >= mm1 ? pa + 1 - mm1 : pa + 1;
            pb = pb + 1 >= mm1 ? pb + 1 - mm1 : pb + 1;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bi[s] = bi[s] + 2 >= mm1 ? bi[s] + 2 - mm1 : bi[s] + 2;
                bj[s] = bj[s] + 2 >= mm1 ? bj[s] + 2 - mm1 : bj[s] + 2;
            }
        }
        // quadratic convergence: once every rotation of a sweep is below 1e-8 (relative), the off-diagonal left
        // behind is below 1e-16: done, no confirmation sweep needed.
        if (lane == 0 && dbg) *dbg = sweep + 1;
        if (!any_rot || !any_big) break;
    }
    if (lane == 0) {
        int best = 0;
        for (int k = 1; k < q; ++k)
            if (A0[k * lda + k] > A0[best * lda + best]) best = k;
        double bigv = 0.0;
        for (int k = 0; k < q; ++k)
            if (fabs(V0[k * lda + best]) > fabs(bigv)) bigv = V0[k * lda + best];
        const double sg = bigv < 0.0 ? -1.0 : 1.0;
        for (int k = 0; k < q; ++k) vout[k] = sg * V0[k * lda + best];
    }
}

// Dominant eigenvector of the symmetric positive semi-definite q x q matrix G (LDS, preserved) by repeated
// squaring, one wave.  Matrices are zero-padded to QP x QP (lda even), so every loop has a compile-time trip
// count and its LDS loads issue back to back (a predicated `k < q` loop compiles to one branch + one exposed LDS
// latency per k and made this routine 4x slower).  Returns false (wave-uniform) if the monitor did not
// converge: the caller falls back to Jacobi.  B0/B1: zero-initialised work buffers.
template <int QP>
__device__ static bool dominant_by_squaring(int q, int lda, const double *G, double *B0, double *B1, double *vout,
                                            double *dbg)
{
    const int lane = threadIdx.x & 63;
    const int nent = q * (q + 1) / 2;
    constexpr int NS = (QP * (QP + 1) / 2 + 63) / 64;
    int k1s[NS], k2s[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        int e = lane + 64 * s, k1 = 0;
        if (e < nent) {
            while (e >= q - k1) { e -= q - k1; ++k1; }
            k1s[s] = k1; k2s[s] = k1 + e;
            const double v = G[k1 * lda + k1 + e];
            B0[k1 * lda + k1 + e] = v;
            B0[(k1 + e) * lda + k1] = v;
        } else { k1s[s] = -1; k2s[s] = 0; }
    }
    wavesync();
    double *A = B0, *Bn = B1;
    int extra = -1, it = 0;
    bool ok = false;
    for (; it < 24; ++it) {
        double tp[4] = {0.0, 0.0, 0.0, 0.0};   // 4 partial sums: short dependent chains (one wave alone on its SIMD)
#pragma unroll
        for (int k = 0; k < QP; ++k) tp[k & 3] += A[k * lda + k];
        const double t = (tp[0] + tp[1]) + (tp[2] + tp[3]);
        // A = (previous A)^2 / tr(previous A)^2, so t = sum(lambda^2)/(sum lambda)^2 -> 1 as A -> rank one
        if (it > 0 && extra < 0 && (1.0 - t) < 1e-12) extra = 1;   // then ONE more squaring: rho ~5e-13 -> ~1e-25
        if (extra == 0) { ok = true; break; }
        if (extra > 0) --extra;
        double isc = __builtin_amdgcn_rcp(t);      // any common scale factor will do: estimate + one Newton step
        isc = isc * (2.0 - t * isc);
        const double isc2 = isc * isc;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k1 = k1s[s] < 0 ? 0 : k1s[s], k2 = k2s[s];
            const double *ra = A + k1 * lda, *rb = A + k2 * lda;
            double ap[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < QP; ++k) ap[k & 3] += ra[k] * rb[k];
            const double acc = ((ap[0] + ap[1]) + (ap[2] + ap[3])) * isc2;
            if (k1s[s] >= 0) {
                Bn[k1 * lda + k2] = acc;
                Bn[k2 * lda + k1] = acc;
            }
        }
        wavesync();
        double *tmp = A; A = Bn; Bn = tmp;
    }
    if (lane == 0 && dbg) *dbg = ok ? 100 + it : -1;
    if (!ok) return false;
    if (lane == 0) {
        double dg[QP];
#pragma unroll
        for (int k = 0; k < QP; ++k) dg[k] = A[k * lda + k];
        int best = 0;
        double bd = dg[0];
#pragma unroll
        for (int k = 1; k < QP; ++k)
            if (dg[k] > bd) { bd = dg[k]; best = k; }
        double col[QP];
#pragma unroll
        for (int k = 0; k < QP; ++k) col[k] = A[k * lda + best];
        double ss = 0.0, bigv = 0.0;
#pragma unroll
        for (int k = 0; k < QP; ++k) {
            ss += col[k] * col[k];
            if (fabs(col[k]) > fabs(bigv)) bigv = col[k];
        }
        const double sc = (bigv < 0.0 ? -1.0 : 1.0) / sqrt(ss);
#pragma unroll
        for (int k = 0; k < QP; ++k) vout[k] = sc * col[k];
    }
    return true;
}

// uniform broadcast of one lane's double (lane index may be a runtime value as long as it is wave-uniform)
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// Same job as dominant_by_squaring for the 16 x 16 case, entirely in registers on the matrix cores.
// v_mfma_f64_16x16x4 takes A[m = lane&15][k = lane>>4] and B[k = lane>>4][n = lane&15] and returns
// D[m = (lane>>4) + 4 reg][n = lane&15].  For a SYMMETRIC matrix held as x[j] = A[4 j + (lane>>4)][lane&15]
// (j = 0..3) the register x[j] is at once the A-operand and the B-operand of k-block j, and the four result registers
// of A*A are again exactly that layout: one squaring = 4 dependent MFMAs, no LDS, no shuffles.  The trace lives on
// the 16 lanes 16 (m&3) + m (register m>>2) and is collected with v_readlane.  Each trip normalises by the trace and
// squares THREE times: x <- (x / tr x)^8, so tr x = sum mu^8 for the normalised eigenvalues mu of the previous matrix.
// 1 - tr x < 1e-2 means mu_1^8 > 0.99, i.e. every other eigenvalue of the previous matrix was below 1.3e-3 of the first,
// and x (its 8th power) carries them at < (1.3e-3)^8 = 6e-24: rank one far below rounding.  Returns false (wave-uniform)
// after 9 trips = 27 squarings (singular-value gap below ~1e-5): the caller falls back to Jacobi.  One wave; G in LDS
// (ld lda), zero-padded to 16.
__device__ static bool dominant_by_squaring_mfma16(int q, int lda, const double *G, double *vout, double *dbg, double *stamps = nullptr)
{
    typedef double v4f64 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, kap = lane >> 4, l15 = lane & 15;
    double x[4];      // D layout of the 16 x 16 iterate: x[j] = A[4 j + kap][l15]
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = G[(4 * j + kap) * lda + l15];
    // the lane's diagonal entry, if it holds one: A[m][m] sits in lane (kap, l15 = m) with m = 4 j + kap (round 3: the trace by a
    // 16-lane DPP sum of these + four row totals instead of 16 readlanes per round; the column extraction below likewise works on
    // the four lanes that hold the column instead of broadcasting its 16 entries)
    const int jd = l15 - kap;
    const bool hasd = jd >= 0 && (jd & 3) == 0;
    const int jsel = jd >> 2;
    auto diag_of = [&]() { double d = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) d = (hasd && jsel == j) ? x[j] : d;
        return d; };
    auto rowsum16 = [](double v) { v += jch_dpp<0x128>(v); v += jch_dpp<0x124>(v); v += jch_dpp<0x122>(v); v += jch_dpp<0x121>(v); return v; };
    bool ok = false;
    int it = 0;
    if (stamps && lane == 0) stamps[0] = (double)__builtin_readcyclecounter();
    for (; it < 10; ++it) {
        const double rs = rowsum16(diag_of());
        const double t = (readlane_f64(rs, 0) + readlane_f64(rs, 16)) + (readlane_f64(rs, 32) + readlane_f64(rs, 48));
        if (it > 0 && (1.0 - t) < 1e-2) { ok = true; break; }
        if (it == 9) break;
        double isc = __builtin_amdgcn_rcp(t);
        isc = isc * (2.0 - t * isc);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] *= isc;
#pragma unroll
        for (int sq = 0; sq < 3; ++sq) {
            v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[j], x[j], acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = acc[j];
        }
    }
    if (lane == 0 && dbg) *dbg = ok ? 300 + 3 * it : -1;
    if (stamps && lane == 0) stamps[1] = (double)__builtin_readcyclecounter();
    if (!ok) return false;
    // dominant eigenvector = the column of the largest diagonal entry (first one among equals), normalised, largest-|.| > 0
    double dg[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) dg[m] = readlane_f64(x[m >> 2], 16 * (m & 3) + m);
    int best = 0;
    double bd = dg[0];
#pragma unroll
    for (int m = 1; m < 16; ++m)
        if (dg[m] > bd) { bd = dg[m]; best = m; }
    best = __builtin_amdgcn_readfirstlane(best);
    // column `best`: rows 4 j + kap are x[j] of lane (kap, best).  Sum of squares and the entry of largest magnitude (the first
    // one in row order among equals, as before) over those four lanes
    double pss = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) pss += x[j] * x[j];
    const double ss = (readlane_f64(pss, best) + readlane_f64(pss, 16 + best)) + (readlane_f64(pss, 32 + best) + readlane_f64(pss, 48 + best));
    double bigv = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {                 // row m = 4 j + kk, in order
            const double c = readlane_f64(x[j], 16 * kk + best);
            if (fabs(c) > fabs(bigv)) bigv = c;
        }
    double y = __builtin_amdgcn_rsq(ss);            // 1 / sqrt(ss): hardware estimate + two Newton steps (full double)
    y = y * (1.5 - 0.5 * ss * y * y);
    y = y * (1.5 - 0.5 * ss * y * y);
    const double sc = bigv < 0.0 ? -y : y;
    if (l15 == best) {
#pragma unroll
        for (int j = 0; j < 4; ++j) vout[4 * j + kap] = 4 * j + kap < q ? sc * x[j] : 0.0;
    }
    if (stamps && lane == 0) stamps[2] = (double)__builtin_readcyclecounter();
    return true;
}

// dot of column k of the LDS-resident K (ld ldk) with an LDS vector over rows j = j0, j0+stride, ... :
// 8 rows per batch with clamped addresses so the loads are unconditional and issue together.
__device__ __forceinline__ double kcol_dot(const double *Kl, int ldk, int k, const double *x, int j0, int stride, int p)
{
    double s = 0.0;
    for (int j = j0; j < p; j += 8 * stride) {
        double a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int jj = min(j + u * stride, p - 1);
            a[u] = Kl[jj * ldk + k];
            b[u] = x[jj];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (j + u * stride < p) ? a[u] * b[u] : 0.0;
    }
    return s;
}
