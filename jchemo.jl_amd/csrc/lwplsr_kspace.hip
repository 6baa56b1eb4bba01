// K8g — batched local weighted plskern in NEIGHBOUR SPACE (round 3): the local fits of predict(::Lwplsr)
// (src/locwlv.jl:9-48: one weighted plskern on the k neighbour rows + 1-row predictions per query, src/plskern.jl:106-178,
// 226-238) with ONE pass over the gathered rows instead of one per latent variable.
//
// Every p-vector of the local fit lies in the row space of the centred (scaled) neighbour block Xc (k x p, k = 200 at cfg5,
// p = 500): K = Xc'A with A = D Yc (k x q, deflated as A <- A - (D t) c'), w = Xc'om, r = Xc'rho, and the fit only ever
// needs inner products of rows — the k x k Gram matrix G = Xc Xc' and g = Xc xq_c:
//     M  = A'H               (q x q Gram of K = Xc'A;  H = G A maintained as H <- H - (G D t) c')
//     v  = dominant eigenvector of M (q == 1: 1),  nrm = sqrt(v'M v) = ||K v||
//     om = A v / nrm,  s = H v / nrm = Xc w                                        (src/plskern.jl:150-155)
//     beta_j = t_j'D s / tt_j,  t = s - sum_j beta_j t_j  (= Xc r: the r-recursion of :156-161 applied to the scores)
//     tau = g'om - sum_j beta_j tau_j  (= xq_c . r),  tt = t'D t,  c = A't / tt    (:162-166)
//     prediction_a = prediction_{a-1} + tau c .* ysd                               (:226-238 on the one query row)
// The rows are taken about a PIVOT — the query row itself, which sits in the middle of its neighbours — so the raw Gram matrix
// G0 = Z Z', Z = (X[s, :] - 1 xq') / sd, carries no cancellation, and the centring is applied implicitly:
//     u = G0 d,  mm = d'u,   G x = G0 x - u (1'x) - 1 (u'x - mm 1'x),   g = -u + mm 1        (xq - pivot = 0)
// (numpy prototype against the oracle at cfg5-like data: predictions equal to 4e-13 at 15 LVs, q = 1 and 3, scal on / off.)
//
// Layout: one 512-thread workgroup per query (eight waves, two per SIMD, 256 registers each).  G0 is built on the matrix cores
// (v_mfma_f64_16x16x4; the SAME LDS operand array serves as A and as B: G0 is a SYRK) from 16-column stages of the gathered rows
// that are staged through LDS once (two buffers) and shared by the eight waves; its 16 x 16 tiles (upper triangle: 91 for 13 row
// blocks) never leave the accumulator registers — wave w owns the tiles whose column block is congruent to w mod 8, 11 or 12 of
// them (see "Tile ownership" below) — and every later product G0 x is taken straight from them (one transposing DPP row sum per
// tile, the transposed use of a tile through a cross-row butterfly, 13 fixed-order partials per entry -> deterministic).  The
// latent-variable phase needs four workgroup barriers per LV: every 208-long sum (norm, g'om, beta, tt, c) is taken per wave.
// HBM traffic per query: the k gathered rows, once (0.8 MB at cfg5; the p-space kernel streamed them 15 + 3 times:
// profiles/r02_pmc_lwplsr_cfg5.txt).  Phase times of one query at cfg5 (JCH_LOCW_DBG=2, us): setup 8, Gram 115 (the matrix pipe
// alone: 23 tiles per SIMD x 125 k-steps x 64 cycles = 77), u and H 8, 15 LVs 84 (per LV: sums + beta 0.9, scores 0.5, tt / c
// 0.6, product 2.9, combine + deflate 0.5).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <utility>

#include "jch_internal.h"
#include "lv_device.h"
#include "rowsum_dev.h"
#include "lwplsr_dev.h"

#define KS_NT 512        // threads per workgroup: 8 waves, TWO per SIMD.  (Not for the matrix pipe — ONE wave saturates it: a
                         // v_mfma_f64_16x16x4 holds the SIMD for 64 cycles whoever issues it, 78 TFLOP/s chip-wide with one wave or
                         // two, tools/mfma_f64_rate2.hip — but the second wave's products run while the first one waits for its
                         // stage loads, LDS reads and barriers, and the vector-bound latent-variable phase interleaves two waves)
#define KS_NW 8
#define KS_KB 13         // 16-row blocks of the Gram matrix: k <= 208 (smaller k is zero-padded)
#define KS_KP (16 * KS_KB)
#define KS_ND 7          // tile classes delta = J - I (mod 13) = 0 .. 6: every unordered pair of row blocks exactly once
#define KS_QS (4 * KS_KP + 8)   // doubles per column quad of an LDS stage: four columns of KS_KP rows each, column-major — an operand
                         // read (lane (kap, l15): row 16 I + l15 of column kap) is then two 128-B runs 1664 B = 128 (mod 256) apart, i.e. all 64
                         // banks once (with [row][4 columns] the 16 rows of a read were 32 B apart and met in pairs: SQ_LDS_BANK_CONFLICT was
                         // 46 k cycles per query) — + 8 so that the four quads a stage store touches sit 64 B apart
#define KS_TPW 12        // accumulator slots per wave: its 11 or 12 tiles in the order of their row blocks
#define KS_AD 4          // operand pairs in flight ahead of their product
#define KS_CS 16         // columns per LDS stage
#define KS_NR 4          // load rounds per stage: 8 waves x 8 rows per round
#define KS_MAXNLV 48

// (The products are issued through __builtin_amdgcn_mfma_f64_16x16x4f64, not inline asm: an asm statement hides the instruction from the compiler's hazard recognizer, and back-to-back
// dependent f64 products on ONE accumulator — the four k-steps of a guarded tile — then lose part of the sum: measured, rows
// 12 .. 15 of those tiles.  Pinning the tiles to AccVGPRs through an asm constraint bought no time anyway.)
typedef double v2f64k __attribute__((ext_vector_type(2)));

// Tile ownership.  The 91 tiles of the upper triangle are the pairs (I, J = I + delta), I = 0 .. 12, delta = 0 .. 6 (J >= 13 is row
// block J - 13: every unordered pair of row blocks exactly once).  Wave w owns the COLUMN blocks J = w, w + 8, w + 16: among the seven J of a
// row block I at most one is congruent to w mod 8, so wave w has the tile (I, I + (w - I) mod 8) of every row block I except
// I = e1 = (w + 1) mod 8 and I = e1 + 8 (where (w - I) mod 8 = 7): 11 tiles (e1 <= 4) or 12, 23 / 23 / 23 / 22 per SIMD (waves
// w and w + 4).  Slot r of a wave is its r-th tile in the order of I; the LDS addresses of a slot's two operands are per-wave
// values kept in registers, so ONE instruction stream serves the eight waves and nothing but the operand reads sits between
// the products.  That matters on this chip: v_mfma_f64_16x16x4 occupies the SIMD for 64 cycles (78 TFLOP/s chip-wide, the
// vector f64 rate: tools/mfma_f64_rate2.hip) and vector instructions do NOT run beside it — a 64-bit select per product costs
// 17 cycles, a v_add_f64 12, an LDS read 2 .. 7 — whatever sits between the products is paid in full.
__device__ __forceinline__ int ks_e1(int w) { return (w + 1) & (KS_NW - 1); }
__device__ __forceinline__ int ks_ntiles(int w) { return ks_e1(w) + 8 < KS_KB ? KS_TPW - 1 : KS_TPW; }
__device__ __forceinline__ int ks_slot_I(int w, int r) { const int e1 = ks_e1(w); int I = r; if (I >= e1) ++I; if (I >= e1 + 8) ++I; return I; }
__device__ __forceinline__ int ks_slot_J(int w, int I) { const int J = I + ((w - I) & (KS_NW - 1)); return J >= KS_KB ? J - KS_KB : J; }   // its column block, as a row block
__device__ __forceinline__ int ks_rank(int w, int I) { const int e1 = ks_e1(w); return I - (I > e1 ? 1 : 0) - (I > e1 + 8 ? 1 : 0); }   // slot of row block I in wave w

typedef double v4f64k_ __attribute__((ext_vector_type(4)));
// NKS k-steps (4 columns each) of one LDS stage: G0 tile (I, J) += Z[rows of I][cols] Z[rows of J][cols]'.  pa[r] / pb[r]: the LDS
// address of slot r's row blocks I / J in this stage buffer + the lane part (kap + 4 l15: operand lane (m = l15 -> row, kap ->
// column)); the same array serves as A and as B (SYRK).  The 11 slots every wave has run as one software-pipelined stream with
// KS_AD operand pairs in flight; the twelfth (waves 4 .. 6) follows under a wave-uniform branch.
template <int NKS>
__device__ __forceinline__ void ks_gram(v4f64k_ (&acc)[KS_TPW], const double *const (&pa)[KS_TPW], const double *const (&pb)[KS_TPW], bool has12, int koff)
{
    constexpr int NM = KS_TPW - 1, NP = NKS * NM, KST = KS_QS;
    double aq[KS_AD], bq[KS_AD];
#pragma unroll
    for (int n = 0; n < KS_AD && n < NP; ++n) { aq[n] = pa[n % NM][koff + (n / NM) * KST]; bq[n] = pb[n % NM][koff + (n / NM) * KST]; }
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        const double a = aq[n % KS_AD], b = bq[n % KS_AD];
        if (n + KS_AD < NP) {
            aq[n % KS_AD] = pa[(n + KS_AD) % NM][koff + ((n + KS_AD) / NM) * KST];
            bq[n % KS_AD] = pb[(n + KS_AD) % NM][koff + ((n + KS_AD) / NM) * KST];
        }
        acc[n % NM] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[n % NM], 0, 0, 0);
    }
    if (has12) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            acc[NM] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[NM][koff + ks * KST], pb[NM][koff + ks * KST], acc[NM], 0, 0, 0);
    }
}

__device__ __forceinline__ double ks_rowsum16(double v)   // sum over the 16 lanes of a DPP row, result in every lane
{
    v += jch_dpp<0x128>(v);   // row_ror:8
    v += jch_dpp<0x124>(v);   // row_ror:4
    v += jch_dpp<0x122>(v);   // row_ror:2
    v += jch_dpp<0x121>(v);   // row_ror:1
    return v;
}

// sum over the 64 lanes without LDS traffic: DPP inside the 16-lane rows, then the four row totals through v_readlane
__device__ __forceinline__ double ks_wave_sum(double v)
{
    v = ks_rowsum16(v);
    return (jch_readlane(v, 0) + jch_readlane(v, 16)) + (jch_readlane(v, 32) + jch_readlane(v, 48));
}
// Sums of FOUR per-lane values over the 16 lanes of a DPP row in 5 DPP steps instead of 16 (the transposing reduction of
// rowsum_dev.h one level deeper): on return the lane with (bit 3, bit 2) = (b, c) of its row index holds the total of r[2 c + b].
__device__ __forceinline__ double ks_rowsum16x4(const double (&r)[4], int l15)
{
    const bool b3 = (l15 & 8) != 0, b2 = (l15 & 4) != 0;
    double u = b3 ? r[1] : r[0], v = b3 ? r[0] : r[1];
    u += jch_dpp<0x128>(v);                          // row_ror:8 == lane ^ 8: u = partial of r[b3]
    double u2 = b3 ? r[3] : r[2], v2 = b3 ? r[2] : r[3];
    u2 += jch_dpp<0x128>(v2);                        // partial of r[2 + b3]
    double t = b2 ? u2 : u;
    const double z = b2 ? u : u2;
    t += jch_dpp<0x141>(z);                          // row_half_mirror (flips bits 0..2): partial of r[2 b2 + b3]
    t += jch_dpp<0xB1>(t);                           // quad_perm [1,0,3,2]
    t += jch_dpp<0x4E>(t);                           // quad_perm [2,3,0,1]
    return t;
}
// Sums over the four lane ROWS (lanes l, l ^ 16, l ^ 32, l ^ 48) of four values at once (v_permlane32_swap / v_permlane16_swap):
// on return row 0 holds the sums of c0, row 1 of c2, row 2 of c1, row 3 of c3.
__device__ __forceinline__ double ks_colsum4(double c0, double c1, double c2, double c3)
{
    jch_fold32(c0, c1);
    jch_fold32(c2, c3);
    jch_fold16(c0, c2);
    return c0;
}

// The wave's share of y = G0 x, straight from the accumulator registers (D layout: lane (kap, l15), register reg of the slot of
// tile (I, J) holds G0[16 I + kap + 4 reg][16 J + l15]).  Rows of block I from G0[I][J] x_J: ONE transposing 16-lane reduction for
// the four registers (pR[slot]); rows of block J from the tile's columns against x_I (it is also G0[J][I]'): four registers, then
// the four lane rows, four tiles per permlane reduction (pC[slot]).
__device__ __forceinline__ void ks_matvec(const v4f64k_ (&acc)[KS_TPW], const double *x, double *pR, double *pC, int wv, int kap, int l15)
{
    const int nt = ks_ntiles(wv);
    const int rrow = kap + 4 * (2 * ((l15 >> 2) & 1) + (l15 >> 3));   // the row whose total ks_rowsum16x4 leaves in this lane
    double cp[KS_TPW];
#pragma unroll
    for (int r = 0; r < KS_TPW; ++r) {
        const bool valid = r < nt;                                  // (wave-uniform; an unused slot holds zeros)
        const int I = valid ? ks_slot_I(wv, r) : 0, J = valid ? ks_slot_J(wv, I) : 0;
        const double xJ = x[J * 16 + l15];
        double q4[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) q4[reg] = acc[r][reg] * xJ;
        const double t = ks_rowsum16x4(q4, l15);
        if ((l15 & 3) == 0) pR[r * 16 + rrow] = t;
        double c = 0.0;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) c += acc[r][reg] * x[I * 16 + kap + 4 * reg];
        cp[r] = J == I ? 0.0 : c;                                   // (a diagonal tile is its own transpose)
    }
#pragma unroll
    for (int gq = 0; gq < KS_TPW / 4; ++gq) {                       // four tiles per reduction; lane rows hold c0, c2, c1, c3
        const double c = ks_colsum4(cp[4 * gq], cp[4 * gq + 1], cp[4 * gq + 2], cp[4 * gq + 3]);
        pC[(4 * gq + (((kap & 1) << 1) | (kap >> 1))) * 16 + l15] = c;
    }
}

// block sums of NV values (one partial per thread and value): in-wave sums, then the 8 wave partials through LDS in a fixed
// order.  red: >= 8 * NV doubles.  Result valid in every thread.
template <int NV>
__device__ __forceinline__ void ks_block_sums(double (&v)[NV], double *red)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = ks_wave_sum(v[i]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wv * NV + i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i)
        v[i] = ((red[i] + red[NV + i]) + (red[2 * NV + i] + red[3 * NV + i])) + ((red[4 * NV + i] + red[5 * NV + i]) + (red[6 * NV + i] + red[7 * NV + i]));
}

struct ks_lds {   // offsets in doubles
    int stage, T, ypR, ypC, dl, A, H, gv, uv, xv, sv, red, ys, vl, eig, beta, tauh, tth, idx, total;
};
__host__ __device__ inline ks_lds ks_layout(int Q, int nlv)
{
    ks_lds L;
    int o = 0;
    // T history and the product partials alias the stage buffers (free after the Gram phase)
    L.stage = o; L.T = o; L.ypR = o + nlv * KS_KP; L.ypC = L.ypR + KS_NW * KS_TPW * 16;
    int need = 2 * (KS_CS / 4) * KS_QS;                    // two stage buffers [4 column quads][4 columns][KS_KP rows]
    if (nlv * KS_KP + 2 * KS_NW * KS_TPW * 16 > need) need = nlv * KS_KP + 2 * KS_NW * KS_TPW * 16;
    o += need;
    L.dl = o; o += KS_KP;
    L.xv = o; o += KS_KP;
    L.A = o; o += KS_KP * Q;
    L.H = o; o += KS_KP * Q;
    L.gv = o; o += KS_KP; L.uv = o; o += KS_KP; L.sv = o; o += KS_KP;
    L.red = o; o += 8 * (Q * (Q + 1) / 2 + Q + 2) + 16;
    L.ys = o; o += 4 * Q;
    L.vl = o; o += 16;
    L.eig = o; o += 5 * Q * (Q + 2) + 2 * (Q + 2) + 8;
    L.beta = o; o += KS_MAXNLV; L.tauh = o; o += KS_MAXNLV; L.tth = o; o += KS_MAXNLV;
    L.idx = o; o += KS_KP / 2 + 1;
    L.total = o;
    return L;
}

template <int Q>
__global__ __launch_bounds__(KS_NT, 1) void k_locw_kspace(locw_args g)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kap = lane >> 4, l15 = lane & 15;
    const int p = g.p, ldr = g.ldr, k = g.k, q = g.q;
    constexpr int KP = KS_KP;
    const ks_lds L = ks_layout(Q, g.nlv_hi);
    double *stage = lds + L.stage, *Th = lds + L.T, *ypR = lds + L.ypR, *ypC = lds + L.ypC;
    double *dl = lds + L.dl, *Am = lds + L.A, *Hm = lds + L.H, *gv = lds + L.gv, *uv = lds + L.uv, *xv = lds + L.xv;
    double *red = lds + L.red, *ys = lds + L.ys, *vl = lds + L.vl;
    double *beta = lds + L.beta, *tauh = lds + L.tauh, *tth = lds + L.tth;
    int *idx = reinterpret_cast<int *>(lds + L.idx);
    constexpr int lda = Q + 2;
    double *G0m = lds + L.eig, *E0 = G0m + Q * lda, *E1 = E0 + Q * lda, *V0 = E1 + Q * lda, *V1 = V0 + Q * lda, *csl = V1 + Q * lda;
    double *ymean = ys, *ysd = ys + Q, *prun = ys + 2 * Q;
    const int le = g.nlv_hi - g.nlv_lo + 1;
    const int nlvloc = min(min(k, p), g.nlv_hi);
    const int nstage = (ldr + KS_CS - 1) / KS_CS;
    double *sgl = g.scratch + (size_t)blockIdx.x * g.slab;   // [ldr] local column stds (scal only)
    v4f64k_ acc[KS_TPW];
    const int c8 = lane & 7, r8 = lane >> 3;   // stage-load role: column pair 2 c8 of row 8 wv + r8 (+ 64 per round)

#define KS_STAMP(i) do { if ((g.dbg & 2) && tid == 0 && blockIdx.x == 0 && qi == 0) sgl[i] = (double)wall_clock64(); } while (0)
    for (int qi = blockIdx.x; qi < g.m; qi += gridDim.x) {
        __syncthreads();
        KS_STAMP(0);
        if (tid == 0 && g.flags) g.flags[qi] = 0;   // (every query's flag is written by its workgroup: no memset in front of the launch; thread 0 also raises it)
        // ---- weights (mweight), neighbour ids, Y rows, Y means / stds; A = D Yc
        double s0 = 0.0;
        for (int e = tid; e < k; e += KS_NT) { idx[e] = (g.dbg & 1) ? e : g.ind[(size_t)qi * k + e]; s0 += g.w[(size_t)qi * k + e]; }
        {
            double t1[1] = {s0};
            ks_block_sums<1>(t1, red);
            s0 = t1[0];
        }
        const double sw = s0;
        double yrow[Q];
        double dme = 0.0;
        {
            double sy[Q];
#pragma unroll
            for (int y = 0; y < Q; ++y) { sy[y] = 0.0; yrow[y] = 0.0; }
            double ymin = __builtin_inf(), ymax = -__builtin_inf();
            if (tid < KP) {
                const int e = tid;
                dme = e < k ? g.w[(size_t)qi * k + e] / sw : 0.0;
                dl[e] = dme;
                if (e < k) {
#pragma unroll
                    for (int y = 0; y < Q; ++y) {
                        yrow[y] = y < q ? g.Y[(size_t)idx[e] + (size_t)y * (size_t)g.ldy] : 0.0;
                        sy[y] += dme * yrow[y];
                    }
                    ymin = ymax = yrow[0];
                }
            }
            ks_block_sums<Q>(sy, red);
            if (tid < Q) ymean[tid] = sy[tid];
            // constant-y shortcut, univariate y only (src/locwlv.jl:25-28)
            for (int o = 32; o > 0; o >>= 1) { ymin = fmin(ymin, __shfl_xor(ymin, o, 64)); ymax = fmax(ymax, __shfl_xor(ymax, o, 64)); }
            __syncthreads();
            if (lane == 0) { red[wv] = ymin; red[8 + wv] = ymax; }
            __syncthreads();
            double gmin = red[0], gmax = red[8];
#pragma unroll
            for (int w8 = 1; w8 < KS_NW; ++w8) { gmin = fmin(gmin, red[w8]); gmax = fmax(gmax, red[8 + w8]); }
            __syncthreads();
            if (q == 1 && gmin == gmax) {
                for (int a = tid; a < le; a += KS_NT) g.pred[(size_t)qi * le + a] = gmin;
                continue;
            }
        }
        {
            double vv[Q];
#pragma unroll
            for (int y = 0; y < Q; ++y) { const double z = yrow[y] - ymean[y]; vv[y] = g.scal ? dme * z * z : 0.0; }
            ks_block_sums<Q>(vv, red);
            if (tid < Q) { ysd[tid] = (g.scal && tid < q) ? sqrt(vv[tid]) : 1.0; prun[tid] = ymean[tid]; }
            __syncthreads();
            if (tid < KP)
#pragma unroll
                for (int y = 0; y < Q; ++y) {
                    const double z = yrow[y] - ymean[y];
                    Am[y * KP + tid] = (y < q && tid < k) ? dme * (g.scal ? z / ysd[y] : z) : 0.0;
                }
            if (tid < q && g.nlv_lo == 0) g.pred[((size_t)qi * le) * q + tid] = ymean[tid];   // nlv = 0: the intercept alone
        }
        __syncthreads();

        // ---- gathered rows in 16-column stages: thread (wave, lane) owns column pair 2 c8 of rows 64 rr + 8 wv + r8
        const int colst = 2 * c8;
        v2f64k xr[KS_NR], pv, sq = {1.0, 1.0};
        const double *rowp[KS_NR];             // the thread's four rows (rows beyond k: the last one again — they meet zero weights only)
#pragma unroll
        for (int rr = 0; rr < KS_NR; ++rr) rowp[rr] = g.Xrm + (size_t)idx[min(64 * rr + 8 * wv + r8, k - 1)] * ldr;
        auto issue = [&](int cs) {
            const int col = min(KS_CS * cs + colst, ldr - 2);
#pragma unroll
            for (int rr = 0; rr < KS_NR; ++rr) xr[rr] = *reinterpret_cast<const v2f64k *>(rowp[rr] + col);
            pv.x = g.Xq[(size_t)qi + (size_t)min(col, p - 1) * (size_t)g.ldxq];
            pv.y = g.Xq[(size_t)qi + (size_t)min(col + 1, p - 1) * (size_t)g.ldxq];
            if (g.scal) {   // written by other waves of this workgroup a moment ago: agent-scope loads (not through this CU's L1)
                sq.x = __hip_atomic_load(sgl + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sq.y = __hip_atomic_load(sgl + col + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        if (g.scal) {
            // local column stds about the local means (uncorrected, weighted: src/utility.jl:314-323), one extra sweep over the rows:
            // var_j = sum_i d_i z_ij^2 - (sum_i d_i z_ij)^2 with z about the pivot (|mean - pivot| is of the order of the spread)
            for (int cs = 0; cs < nstage; ++cs) {
                issue(cs);
                v2f64k s1 = {0.0, 0.0}, s2 = {0.0, 0.0};
#pragma unroll
                for (int rr = 0; rr < KS_NR; ++rr) {
                    const int e = 64 * rr + 8 * wv + r8;
                    const double d = e < k ? dl[e] : 0.0;
                    const double zx = xr[rr].x - pv.x, zy = xr[rr].y - pv.y;
                    s1.x += d * zx; s1.y += d * zy; s2.x += d * zx * zx; s2.y += d * zy * zy;
                }
                // 64 partials per column (8 waves x 8 row groups) through the (still unused) stage area
                double *cp = stage + ((wv * 8 + r8) * KS_CS + colst) * 2;
                cp[0] = s1.x; cp[1] = s2.x; cp[2] = s1.y; cp[3] = s2.y;
                __syncthreads();
                if (tid < KS_CS) {
                    double a1 = 0.0, a2 = 0.0;
                    for (int c = 0; c < 8 * KS_NW; ++c) { a1 += stage[(c * KS_CS + tid) * 2]; a2 += stage[(c * KS_CS + tid) * 2 + 1]; }
                    const int j = KS_CS * cs + tid;
                    if (j < ldr) __hip_atomic_store(sgl + j, j < p ? sqrt(fmax(a2 - a1 * a1, 0.0)) : 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int s = 0; s < KS_TPW; ++s) acc[s] = v4f64k_{0.0, 0.0, 0.0, 0.0};
        // registers -> LDS stage [column quad][row][4], rows about the pivot (and scaled).  Columns >= p (last stage only) are
        // zeroed; rows >= k are NOT (copies of row k - 1): every vector they meet is zero there, and so is every output
        auto put = [&](int cs, double *buf, auto last_) {
            constexpr bool last = decltype(last_)::value;
            const int col = KS_CS * cs + colst;
            const bool c0 = col < p, c1 = col + 1 < p;
            // scal: ONE reciprocal per column and stage, then products — a division per element is ~25 vector instructions, and
            // every one of them is time the matrix pipe stands still (the quotient differs from x / s by at most one ulp)
            v2f64k rq = {1.0, 1.0};
            if (g.scal) { rq.x = 1.0 / sq.x; rq.y = 1.0 / sq.y; }
#pragma unroll
            for (int rr = 0; rr < KS_NR; ++rr) {
                const int e = 64 * rr + 8 * wv + r8;
                if (e < KP) {                                     // (the fourth round covers rows 192 .. 255)
                    v2f64k z;
                    z.x = xr[rr].x - pv.x;
                    z.y = xr[rr].y - pv.y;
                    if (last) { z.x = c0 ? z.x : 0.0; z.y = c1 ? z.y : 0.0; }
                    if (g.scal) { z.x *= rq.x; z.y *= rq.y; }     // (uniform branch)
                    double *dst = buf + (c8 >> 1) * KS_QS + 2 * (c8 & 1) * KP + e;   // [quad][column][row]
                    dst[0] = z.x;
                    dst[KP] = z.y;
                }
            }
        };
        KS_STAMP(1);
        {
            const bool has12 = ks_ntiles(wv) == KS_TPW;
            const double *pa[KS_TPW], *pb[KS_TPW];                 // operand addresses of the slots in the current stage buffer
#pragma unroll
            for (int r = 0; r < KS_TPW; ++r) {
                const int I = r < ks_ntiles(wv) ? ks_slot_I(wv, r) : 0, J = r < ks_ntiles(wv) ? ks_slot_J(wv, I) : 0;
                pa[r] = stage + kap * KP + l15 + I * 16;
                pb[r] = stage + kap * KP + l15 + J * 16;
            }
            constexpr int STG = (KS_CS / 4) * KS_QS;               // doubles per stage buffer
            issue(0);
            if (nstage > 1) put(0, stage, std::false_type{}); else put(0, stage, std::true_type{});
            if (nstage > 1) issue(1);
            __syncthreads();
            // two stages per trip: the buffer parity is a compile-time LDS offset then, and the operand addresses never change
            auto stage_step = [&](int cs, auto par_) {
                constexpr int PAR = decltype(par_)::value;
                ks_gram<KS_CS / 4>(acc, pa, pb, has12, PAR * STG);  // G0 += Z_stage Z_stage': 4 k-steps of 4 columns
                double *nb = stage + (1 - PAR) * STG;
                if (cs + 2 < nstage) { put(cs + 1, nb, std::false_type{}); issue(cs + 2); }   // (its loads were issued a stage ago)
                else put(cs + 1, nb, std::true_type{});
                __syncthreads();
            };
            for (int cs = 0; cs + 1 < nstage; cs += 2) {
                stage_step(cs, std::integral_constant<int, 0>{});
                if (cs + 2 < nstage) stage_step(cs + 1, std::integral_constant<int, 1>{});
            }
            {                                                       // the last stage: only the k-steps that hold columns < p
                const int nks = (ldr - KS_CS * (nstage - 1) + 3) >> 2, lo = ((nstage - 1) & 1) * STG;
#pragma unroll 1
                for (int ks = 0; ks < nks; ++ks) ks_gram<1>(acc, pa, pb, has12, lo + ks * KS_QS);
            }
            __syncthreads();
        }

        // (The thread index is laundered through an empty asm here: every address of the latent-variable phase derives from it,
        // and the compiler otherwise computes them all ONCE before the query loop — some 80 registers held through the Gram
        // phase, which then spilled its stage loads.)
        int tid_lv = threadIdx.x;
        asm volatile("" : "+v"(tid_lv));
        double mm = 0.0, tme = 0.0, dte = 0.0, cv[Q];
#pragma unroll
        for (int y = 0; y < Q; ++y) cv[y] = 0.0;
        KS_STAMP(2);
        {
        const int tid = tid_lv, lane = tid & 63, kap = lane >> 4, l15 = lane & 15;
        for (int job = 0;; ++job) {
            if (job == q + 1) KS_STAMP(3);
            if (job == q + 2) KS_STAMP(4);
            const int a = job - (q + 1);
            const double *x = xv;
            double sx = 0.0, ux = 0.0;                 // 1'x and u'x of this job's product
            if (job == 0) x = dl;
            else if (job <= q) {
                if (tid < KP) {
                    const double v = Am[(job - 1) * KP + tid];
                    xv[tid] = v;
                }
                __syncthreads();                      // x published
            }
            else {
                // ---------------- LV a (src/plskern.jl:149-175 in neighbour space, see the header)
                if constexpr (Q > 1) {
                    constexpr int NE = Q * (Q + 1) / 2;
                    for (int e = tid; e < 5 * Q * lda; e += KS_NT) G0m[e] = 0.0;
                    __syncthreads();
                    // M = A'H, symmetrised: one wave per entry (y1, y2 >= y1) of the triangle, columns of A and H are contiguous in
                    // LDS — no per-thread arrays next to the accumulators, a fixed summation order
                    for (int e = wv; e < NE; e += KS_NW) {
                        int y1 = 0, r = e;
                        while (r >= Q - y1) { r -= Q - y1; ++y1; }
                        const int y2 = y1 + r;
                        double b = 0.0;
                        for (int i = lane; i < KP; i += 64) b += Am[y1 * KP + i] * Hm[y2 * KP + i] + Am[y2 * KP + i] * Hm[y1 * KP + i];
                        b = 0.5 * ks_wave_sum(b);
                        if (lane == 0) { G0m[y1 * lda + y2] = b; G0m[y2 * lda + y1] = b; }
                    }
                    __syncthreads();
                    if (wv == 0) {
                        if (!dominant_by_squaring<Q>(q, lda, G0m, E0, E1, vl, nullptr)) {
                            for (int e = lane; e < Q * lda; e += 64) E0[e] = G0m[e];
                            wavesync();
                            jacobi_wave(q, lda, E0, E1, V0, V1, csl, vl, nullptr);
                        }
                    }
                    __syncthreads();
                }
                if (a == 3) KS_STAMP(6);
                // ---- per WAVE, no workgroup sums (each costs two barriers; the 208-long sums are 4 LDS trips for a wave): |K v|^2 =
                // (A v)'(H v), g'(A v), and beta_j = t_j'D s / tt_j for the wave's share of the finished LVs
                // (the wave's first beta — j = wave — rides along in the same trip through LDS; j = wave + 8, ... only from LV 9 on)
                double p1 = 0.0, p2 = 0.0, pb = 0.0;
                const int j0 = wv < a ? wv : 0;
                for (int i = lane; i < KP; i += 64) {
                    double av = 0.0, hv = 0.0;
                    if constexpr (Q == 1) { av = Am[i]; hv = Hm[i]; }
                    else {
#pragma unroll
                        for (int y = 0; y < Q; ++y) { av += Am[y * KP + i] * vl[y]; hv += Hm[y * KP + i] * vl[y]; }
                    }
                    p1 += av * hv; p2 += gv[i] * av;
                    pb += Th[j0 * KP + i] * dl[i] * hv;
                }
                const double nrm = sqrt(ks_wave_sum(p1));
                const double gom = ks_wave_sum(p2) / nrm;           // g'om, om = A v / nrm
                pb = ks_wave_sum(pb);
                if (wv < a && lane == 0) beta[wv] = pb / (nrm * tth[wv]);
                for (int j = wv + KS_NW; j < a; j += KS_NW) {
                    double bsum = 0.0;
                    for (int i = lane; i < KP; i += 64) {
                        double hv = 0.0;
                        if constexpr (Q == 1) hv = Hm[i];
                        else {
#pragma unroll
                            for (int y = 0; y < Q; ++y) hv += Hm[y * KP + i] * vl[y];
                        }
                        bsum += Th[j * KP + i] * dl[i] * hv;
                    }
                    bsum = ks_wave_sum(bsum);
                    if (lane == 0) beta[j] = bsum / (nrm * tth[j]);
                }
                __syncthreads();                                    // beta published
                if (a == 3) KS_STAMP(7);
                // scores t = s - sum_j beta_j t_j, s = H v / nrm = Xc w;  x = D t for the deflation product
                tme = 0.0; dte = 0.0;
                if (tid < KP) {
                    double hv = 0.0;
                    if constexpr (Q == 1) hv = Hm[tid];
                    else {
#pragma unroll
                        for (int y = 0; y < Q; ++y) hv += Hm[y * KP + tid] * vl[y];
                    }
                    tme = hv / nrm;
                    for (int j = 0; j < a; ++j) tme -= beta[j] * Th[j * KP + tid];
                    Th[a * KP + tid] = tme;
                    dte = dl[tid] * tme;
                    xv[tid] = dte;
                }
                double tau = gom;
                for (int j = 0; j < a; ++j) tau -= beta[j] * tauh[j];
                __syncthreads();                                    // t_a and x published
                if (a == 3) KS_STAMP(8);
                // tt = t'D t and c = A't / tt, again per wave
                // (and 1'x, u'x of the product below for x = D t: the same trip through LDS)
                double pt = 0.0, pc[Q], px = 0.0, pu = 0.0;
#pragma unroll
                for (int y = 0; y < Q; ++y) pc[y] = 0.0;
                for (int i = lane; i < KP; i += 64) {
                    const double ti = Th[a * KP + i], xi = xv[i];
                    pt += xi * ti;
                    px += xi; pu += uv[i] * xi;
#pragma unroll
                    for (int y = 0; y < Q; ++y) pc[y] += Am[y * KP + i] * ti;
                }
                const double tt = ks_wave_sum(pt);
                sx = ks_wave_sum(px); ux = ks_wave_sum(pu);
#pragma unroll
                for (int y = 0; y < Q; ++y) cv[y] = ks_wave_sum(pc[y]) / tt;
                if (tid == 0) { tth[a] = tt; tauh[a] = tau; }
                const int kk = a + 1;
                if (tid < Q) {
                    double c = cv[0];
#pragma unroll
                    for (int y = 1; y < Q; ++y) c = tid == y ? cv[y] : c;
                    const double pr = prun[tid] + tau * c * ysd[tid];
                    prun[tid] = pr;
                    if (tid < q && kk >= g.nlv_lo && kk <= g.nlv_hi) g.pred[((size_t)qi * le + (kk - g.nlv_lo)) * q + tid] = pr;
                }
                if (a == 3) KS_STAMP(9);
                if (kk >= nlvloc) break;              // (uniform) the last LV needs no deflation
            }
            // ---------------- y = G0 x from the accumulator registers, then the centring and the fixed-order combination
            if (job <= q) {
                double px = 0.0, pu = 0.0;
                for (int i = lane; i < KP; i += 64) { const double xi = x[i]; px += xi; pu += uv[i] * xi; }
                sx = ks_wave_sum(px); ux = ks_wave_sum(pu);
            }
            if (job == q + 4) KS_STAMP(10);
            {
                double *pR = ypR + wv * (KS_TPW * 16), *pC = ypC + wv * (KS_TPW * 16);
                ks_matvec(acc, x, pR, pC, wv, kap, l15);
            }
            __syncthreads();
            if (job == q + 4) KS_STAMP(11);
            if (tid < KP) {
                // entry 16 I + ml of G0 x: the seven tiles (I, I + delta) of its row block and the six tiles (I - delta, I) that reach
                // it transposed, each from its owner wave (column block mod 8) and its slot there — 13 terms in a fixed order
                const int I = tid >> 4, ml = tid & 15;
                double v = 0.0;
#pragma unroll
                for (int dlt = 0; dlt < KS_ND; ++dlt) { const int ow = (I + dlt) & (KS_NW - 1); v += ypR[(ow * KS_TPW + ks_rank(ow, I)) * 16 + ml]; }
#pragma unroll
                for (int dlt = 1; dlt < KS_ND; ++dlt) {
                    const int Ip = I - dlt + (I < dlt ? KS_KB : 0);
                    const int ow = (Ip + dlt) & (KS_NW - 1);
                    v += ypC[(ow * KS_TPW + ks_rank(ow, Ip)) * 16 + ml];
                }
                if (job > 0) v = v - uv[tid] * sx - (ux - mm * sx);                       // G x = G0 x - u (1'x) - 1 (u'x - mm 1'x)
                v = tid < k ? v : 0.0;
                if (job == 0) uv[tid] = v;
                else if (job <= q) Hm[(job - 1) * KP + tid] = v;
                else {
#pragma unroll
                    for (int y = 0; y < Q; ++y) {      // deflation: A <- A - (D t) c', H <- H - (G D t) c'
                        Am[y * KP + tid] -= dte * cv[y];
                        Hm[y * KP + tid] -= v * cv[y];
                    }
                }
            }
            if (job == 0) {
                if ((g.dbg & 4) && blockIdx.x == 0 && qi == 0 && tid < KP) sgl[16 + tid] = uv[tid];   // (debug: u = G0 d of the first query)
                double t1[1] = {tid < KP ? dl[tid] * uv[tid] : 0.0};   // (own entry: written by this thread above)
                ks_block_sums<1>(t1, red);
                mm = t1[0];
                if (tid < KP) gv[tid] = tid < k ? mm - uv[tid] : 0.0;
                if (g.flags) {
                    // PIVOT CHECK (round 4).  The Gram matrix is taken about the query row, which is assumed to lie among its
                    // neighbours; they are chosen in the nlvdis-dimensional score space, so in full p-space an outlying or offset
                    // query need not.  The centred quantities then come out of G0 by cancellation, with an error that grows like
                    // (|local mean - query| / spread)^2 eps along the offset direction m = mean - query.  Both are in hand:
                    // u_i = z_i . m, so |m|^2 = d'u = mm and the weighted variance of the rows along m is (d'(u.u) - mm^2) / mm.
                    // Above a ratio of 64 (the bound of the f64 fit's own raw mode, fit.hip) the query is flagged and the caller
                    // refits it with the per-query path, which centres explicitly.
                    double t2[1] = {tid < KP ? dl[tid] * uv[tid] * uv[tid] : 0.0};
                    ks_block_sums<1>(t2, red);
                    const double varm = t2[0] - mm * mm;             // = mm * (variance along m)
                    const bool bad = !(mm * mm <= 4096.0 * varm);    // ratio^2 = mm / var_m = mm^2 / varm > 64^2 (or not finite)
                    if (tid == 0 && bad && mm > 0.0) g.flags[qi] = 1;
                }
#pragma unroll 1
                for (int y = q; y < Q; ++y) if (tid < KP) Hm[y * KP + tid] = 0.0;
            }
            if (job == q + 4) KS_STAMP(12);
            __syncthreads();
        }
        }
        // requested nlv beyond what the local model has: predict clamps to the model's nlv (src/plskern.jl:228-229)
        __syncthreads();
        KS_STAMP(5);
        for (int e = tid; e < (g.nlv_hi - nlvloc) * q; e += KS_NT) {
            const int kk = nlvloc + 1 + e / q, y = e % q;
            if (kk >= g.nlv_lo) g.pred[((size_t)qi * le + (kk - g.nlv_lo)) * q + y] = prun[y];
        }
    }
}

// the shape fits the neighbour-space kernel at all
bool jch_locw_kspace_feasible(const locw_args &g)
{
    if (const char *e = getenv("JCH_LOCW_KSPACE")) { if (atoi(e) == 0) return false; }   // 0: never
    if (g.k > KS_KP || g.k < 2 || g.q > 8 || g.nlv_hi > KS_MAXNLV || g.nlv_hi < 1 || g.ldr > JCH_SWEEP_MAXP || g.ldr < 2) return false;
    const int Q = g.q <= 1 ? 1 : (g.q <= 2 ? 2 : (g.q <= 4 ? 4 : 8));
    return sizeof(double) * (size_t)ks_layout(Q, g.nlv_hi).total + 64 <= 159 * 1024;
}
// ... and is expected to be the faster of the two kernels there
bool jch_locw_kspace_supported(const locw_args &g)
{
    if (!jch_locw_kspace_feasible(g)) return false;
    if (const char *e = getenv("JCH_LOCW_KSPACE")) { if (atoi(e) == 2) return true; }    // 2: whenever the shape fits (tests)
    // the Gram pass costs 208^2 p / 2 matrix flops per query whatever k is (smaller k is zero-padded to 13 row blocks): it pays
    // against nlv sweeps of a k x p slab when k is most of those 208 rows and the row is wide
    return g.k >= 128 && g.p >= 128 && g.nlv_hi >= 3;
}

template <int Q>
static int32_t launch_ks(jch_ctx *ctx, locw_args &g)
{
    const size_t lds = sizeof(double) * (size_t)ks_layout(Q, g.nlv_hi).total + 64;
    static jch_per_device_once attr;
    if (!attr.done(ctx->device)) {
        JCH_HIP(ctx, hipFuncSetAttribute((const void *)k_locw_kspace<Q>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr.mark(ctx->device);
    }
    // as many workgroups as the queries need for the same number of rounds: 1000 queries on 256 CUs are 4 rounds with 250 workgroups as
    // with 256, every workgroup then runs the same number of queries, and the CUs left free take the copy kernels of the neighbour
    // lists, which travel to the host beside this launch (no measurable difference at cfg5)
    const int rounds = (g.m + ctx->cus - 1) / ctx->cus;
    const int nb = std::min(ctx->cus, (g.m + rounds - 1) / rounds);
    g.slab = ((size_t)std::max(g.ldr, 16 + KS_KP) + 31) & ~(size_t)31;
    JCH_TRY(jch_reserve(ctx, ctx->xstage, sizeof(double) * g.slab * nb));
    g.scratch = (double *)ctx->xstage.ptr;
    hipLaunchKernelGGL((k_locw_kspace<Q>), dim3(nb), dim3(KS_NT), lds, ctx->stream, g);
    JCH_HIP(ctx, hipGetLastError());
    if (g.dbg & 4) {
        double u[KS_KP];
        JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        JCH_HIP(ctx, hipMemcpy(u, g.scratch + 16, sizeof u, hipMemcpyDeviceToHost));
        fprintf(stderr, "[jch] u =");
        for (int i = 0; i < KS_KP; ++i) fprintf(stderr, " %.17g", u[i]);
        fprintf(stderr, "\n");
    }
    if (g.dbg & 2) {   // phase stamps of block 0's first query (wall_clock64: 100 MHz)
        double st[13] = {};
        JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        JCH_HIP(ctx, hipMemcpy(st, g.scratch, sizeof st, hipMemcpyDeviceToHost));
        fprintf(stderr, "[jch] k_locw_kspace phases (us): setup %.1f  gram %.1f  u+H %.1f  first LV %.1f  remaining LVs %.1f  (nlv %d)\n", (st[1] - st[0]) * 0.01,
                (st[2] - st[1]) * 0.01, (st[3] - st[2]) * 0.01, (st[4] - st[3]) * 0.01, (st[5] - st[4]) * 0.01, g.nlv_hi);
        fprintf(stderr, "[jch]   LV 4 (us): sums+beta %.2f  scores %.2f  tt,c %.2f  (to product %.2f)  product %.2f  combine+deflate %.2f\n", (st[7] - st[6]) * 0.01,
                (st[8] - st[7]) * 0.01, (st[9] - st[8]) * 0.01, (st[10] - st[9]) * 0.01, (st[11] - st[10]) * 0.01, (st[12] - st[11]) * 0.01);
    }
    return JCH_OK;
}

int32_t jch_launch_locw_kspace(jch_ctx *ctx, locw_args &g)
{
    if (g.q <= 1) return launch_ks<1>(ctx, g);
    if (g.q <= 2) return launch_ks<2>(ctx, g);
    if (g.q <= 4) return launch_ks<4>(ctx, g);
    return launch_ks<8>(ctx, g);
}
