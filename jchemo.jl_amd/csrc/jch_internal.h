// Internal declarations of libjchemo_hip.so (gfx950 only).  Public ABI: include/jchemo_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "jchemo_hip.h"

#define JCH_MAXQ 64       // largest q whose Jacobi workspace (q x q matrices) lives in LDS; beyond: global memory (smallstate.hip)
#define JCH_ZT_SLICES 8      // max second-stage partial slices of the sweep reduction
#define JCH_SWEEP_MAXP 2048  // widest row the register-resident fused sweep holds (16 column chunks of 128)

struct xcopy_key {
    const void *X; int64_t n, ldx; int p, host;
    bool operator==(const xcopy_key &o) const { return X == o.X && n == o.n && ldx == o.ldx && p == o.p && host == o.host; }
};
struct jch_buf {  // grow-only device buffer
    void *ptr = nullptr;
    size_t bytes = 0;
};

struct jch_uid {
    char internal[128];
};
struct jch_rccl {  // RCCL entry points, dlopen'ed on first use (single-GPU users never need RCCL)
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, jch_uid, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

#define JCH_P2P_MAXR 16
struct jch_p2p {   // P2P inbox transport (p2p.hip)
    bool ready = false, tested = false;
    int nranks = 0, rank = 0;
    void *local = nullptr;                     // own inbox (fine-grained device memory)
    void *peer[JCH_P2P_MAXR] = {};             // every rank's inbox as mapped here
    bool opened[JCH_P2P_MAXR] = {};
    unsigned long long *host_status = nullptr, *host_status_dev = nullptr;   // pinned, device-visible sticky error word
    unsigned long long epoch = 0;
    long long timeout_ticks = 0;
    size_t cap = 0;                            // doubles per (parity, rank) slot
    unsigned long long *stats = nullptr;       // device [2 phases][4]: ticks in the exchange, ticks polling flags, calls, -
};

struct jch_ctx {
    int device = 0;
    int cus = 256;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // communicator
    void *comm = nullptr;
    int rank = 0, nranks = 1;
    void *loop = nullptr;            // loopback group (tests; ctx.hip)
    jch_p2p p2p;
    std::vector<double> loop_sum;
    // workspace (grow-only)
    void *hstage = nullptr;          // pinned host staging for the small outputs (grow-only)
    size_t hstage_bytes = 0;
    // the raw row-major working copy a plskern-shaped fit left in `xr` (JCH_REUSE_XCOPY): which X it is a copy of
    xcopy_key xcopy{};
    bool xcopy_valid = false;
    long long xcopy_reused = 0;   // fits that took their kernel matrix from it
    unsigned sweep_seq = 0;   // launches of the plskern-shaped sweep so far (JCH_SWEEP_ALT: alternating walk direction)
    jch_buf gram, xr, yr, xstage, ystage, wstage, tbuf, dnorm, part, kpart, small, colpart, gemm_b, gemm_out, xq, tickets, qz, lw_work, lw_xrm, lvws, lw_flags, lw_screen;
    // profiling
    bool profiling = false;
    int prof_stride = 1;        // jch_ctx_set_profiling(ctx, N > 1): event pairs around every N-th launch of the sampled dominant kernels only
    unsigned prof_seq = 0;
    int64_t sweeps_timed = 0;   // JCH_COUNTER_SWEEPS_TIMED
    jch_profile prof{};
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    size_t ev_mark = 0;
    // collective timing (profiling only): event pairs around the all-reduces that are calls of their own, tagged with the
    // phase of the fit (0 prologue, 1 LV loop); the fused inbox exchange is timed inside its kernel (p2p.stats)
    std::vector<hipEvent_t> cev_pool;
    std::vector<int> cev_phase;      // one entry per PAIR
    size_t cev_used = 0;             // events handed out (2 per pair)
    int coll_phase = 0;
    bool coll_in_fit = false;        // between jch_coll_reset (start of a fit) and jch_coll_collect
    int coll_transport = 0;          // JCH_TRANSPORT_* of the last LV-loop all-reduce
    // second stream + event (created on first use): result copies that may run beside the last kernel of a call (lwplsr.hip)
    hipStream_t aux_stream = nullptr;
    hipEvent_t aux_event = nullptr;
    // tuning knobs (env JCH_SWEEP_BLOCKS_PER_CU etc.)
    int sweep_blocks_per_cu = 0;
    // diagnostics
    long long pivot_refits = 0;      // raw-mode fits repeated on the centred copy because the sampled pivot was poor
    long long knn_screened = 0, knn_screen_redone = 0;   // kNN-LWPLSR queries done by the screened search / redone by the exact selection behind it
    long long locw_refits = 0;       // kNN-LWPLSR queries refitted by the per-query path after the neighbour-space kernel's pivot check
};

// One-time initialisation per (call site, device): hipFuncSetAttribute and occupancy queries are per device, and a
// process may hold ctxs on several GPUs.  The flag is raised AFTER the initialisation (two threads may both run it: benign).
struct jch_per_device_once {
    unsigned long long mask = 0;
    bool done(int dev) const { return (mask >> (dev & 63)) & 1ull; }
    void mark(int dev) { mask |= 1ull << (dev & 63); }
};

// ---- error plumbing --------------------------------------------------------------------------------
int32_t jch_fail(jch_ctx *ctx, int32_t code, const char *fmt, ...);
#define JCH_HIP(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return jch_fail(ctx, e__ == hipErrorOutOfMemory ? JCH_ENOMEM : JCH_EHIP, "%s: %s (%s:%d)", #expr, \
                            hipGetErrorString(e__), __FILE__, __LINE__);                           \
    } while (0)
#define JCH_TRY(expr)                 \
    do {                              \
        int32_t s__ = (expr);         \
        if (s__ != JCH_OK) return s__; \
    } while (0)

int32_t jch_reserve(jch_ctx *ctx, jch_buf &b, size_t bytes);
int32_t jch_reserve_host(jch_ctx *ctx, size_t bytes);   // ctx->hstage (pinned)
int32_t jch_allreduce_f64(jch_ctx *ctx, double *dev_buf, size_t count);  // no-op when nranks == 1
// all-reduce of the per-LV sweep output zt [nslice][ldz] (first m entries of every slice); *nslice_out = slices the
// consumer has to add afterwards (1 when the transport summed them)
int32_t jch_allreduce_slices(jch_ctx *ctx, double *zt, int m, int nslice, int ldz, int *nslice_out);
// p2p.hip
int32_t jch_p2p_allreduce(jch_ctx *ctx, const double *src, size_t count, int nslice, int ldz, double *dst);
int32_t jch_p2p_check(jch_ctx *ctx);
void jch_p2p_destroy(jch_ctx *ctx);

// profiling helpers: record an event on the stream when profiling is on
struct jch_span {
    hipEvent_t a = nullptr, b = nullptr;
};
hipEvent_t jch_ev(jch_ctx *ctx);  // nullptr when profiling is off
bool jch_prof_sample(jch_ctx *ctx);   // profiling on and this launch of a sampled dominant kernel is one to bracket with events
void jch_coll_begin(jch_ctx *ctx);   // profiling: event in front of / behind an all-reduce call (ctx.hip)
void jch_coll_end(jch_ctx *ctx);
void jch_coll_reset(jch_ctx *ctx);   // start of a fit: forget the pairs, zero the inbox tick counters
void jch_coll_collect(jch_ctx *ctx, jch_profile &pr);   // after the fit's final sync: fill the collective fields

// sweep output as the per-block partial rows (the launcher skipped its k_reduce_part): part [nb][ldpart]
struct jch_part_view {
    const double *part = nullptr;
    int nb = 0, ldpart = 0;
};

// ---- kernel launchers (each enqueues on ctx->stream; no host sync) --------------------------------
// prologue.hip
int32_t jch_launch_weights(jch_ctx *ctx, const double *w_dev /*may be null*/, int64_t n, double *dnorm,
                           double *hdr /*[4] device: sum w, n_total*/, double *zero0 = nullptr, int nzero0 = 0,
                           double *zero1 = nullptr, int nzero1 = 0 /*two small regions zeroed by the same launch*/);
int32_t jch_launch_moments(jch_ctx *ctx, const double *Xc, int64_t ldx, const double *Yc, int64_t ldy,
                           const double *d, int64_t n, int p, int q, const double *means /*null: first moment*/,
                           double *out /*[p+q] device*/, bool do_sqrt = true);
int32_t jch_launch_center_xty(jch_ctx *ctx, double *Xc, int64_t ldx, double *Yc, int64_t ldy, const double *d,
                              int64_t n, int p, int q, const double *mom, const double *scl, bool writeback,
                              double *Xr, int ldr, double *Yr, int qpad, double *K /*[p][qpad] device*/, bool scal,
                              double *means_out = nullptr /*raw mode: X is copied minus the pivot mom[0..p), its weighted means land here*/,
                              double *mshift_out = nullptr /*means - pivot*/,
                              const double *spread2 = nullptr /*[p] sample variances from jch_launch_pivot*/,
                              double *qual = nullptr /*[1] max_j |means - pivot| / spread (atomicMax; zeroed by the caller)*/,
                              double *ones_out = nullptr /*raw mode: == mom's storage; [0, p + q) becomes 1.0 and means_out[p..p+q) = mom[p..p+q)*/);
int32_t jch_launch_pivot(jch_ctx *ctx, const double *Xc, int64_t ldx, int64_t n, int p, const double *hdr /*[1] = n_total*/,
                         double *pivot /*[p] device, same on all ranks*/, double *spread2 /*[p]*/);
int32_t jch_launch_export_colmajor(jch_ctx *ctx, const double *Xr, int ldr, const double *Yr, int qpad, int64_t n,
                                   int p, int q, double *Xc, int64_t ldx, double *Yc, int64_t ldy,
                                   const double *sqrt_rowscale = nullptr /*rows scaled by sqrt(d_i): plswold! row metric*/);
// sweep.hip
int32_t jch_launch_sweep(jch_ctx *ctx, const double *Xr, int64_t n, int p, int ldr, const double *d, const double *rvec,
                         const double *Yr, int qpad, int q_extra /*0: plskern; q: also c_raw (plsnipals)*/,
                         double *tcol, double *zt /*[nslice][ldz] device, reduced over blocks*/, int ldz, int max_slices,
                         int *nslice_out, const double *mu = nullptr /*raw mode: Xr is uncentred; t = x.r - mu.r, st at [ldr+1]*/,
                         jch_part_view *pv = nullptr /*non-null: the launcher MAY leave the block partials unreduced and describe them here (pv->part stays null when it reduced into zt as usual)*/);
int32_t jch_launch_reduce_rows(jch_ctx *ctx, const double *part, int nb, int ldpart, int m, double *out);
int32_t jch_launch_xty_rows(jch_ctx *ctx, const double *Xr, int ldr, const double *Yc, int64_t ldy, const double *d, int64_t n, int p, int q,
                            const double *mom, double *Yr, int qpad, double *K, double *means_out, double *mshift_out,
                            const double *spread2, double *qual, double *ones_out, bool *done);
int32_t jch_launch_raw_scales(jch_ctx *ctx, const double *Xr, int64_t n, int p, int ldr, const double *d, const double *mshift,
                              const double *Yr, int qpad, int q, double *tmp, double *scl, double *K);
int32_t jch_launch_reduce_part8(jch_ctx *ctx, const double *part, int nb, int ldpart, int m, double *zt, int ldz, int *nslice_out);
int32_t jch_launch_sweep_wide(jch_ctx *ctx, const double *Xr, int64_t n, int ldr, const double *d, const double *rvec,
                              const double *Yr, int qpad, bool nipals, double *tcol, double *zt);
int32_t jch_launch_ytdt(jch_ctx *ctx, const double *Yr, int64_t n, int qpad, const double *d, const double *tcol, double *out /*[qpad]*/);   // sweep_wide.hip
// plsnipals with postponed write-back (sweep.hip, deflate.hip): the working copy holds the rows of `npend` LVs ago; pending
// loadings pend_p[k][jch_nipals_lazy_pitch(ldr)] (pad columns zero), pending scores tpend + k * tstride, oldest first
#define JCH_NIPALS_DEFER_DEFAULT 6   // rows rewritten every 6th LV (cfg4, ms per LV: eager 8.5-8.9; m = 2: 6.6, 4: 5.76, 5-7: 5.3-5.55, 8-9: 5.45-5.5 — up to 4 pending corrections hide behind the loads, each further one costs ~0.17 ms per pass in LDS reads)
static inline int jch_nipals_lazy_pitch(int ldr) { return 128 * (ldr <= 128 ? 1 : ldr <= 256 ? 2 : ldr <= 512 ? 4 : ldr <= 1024 ? 8 : 16); }
int32_t jch_sweep_tickets(jch_ctx *ctx, int **out);   // sweep.hip: counters of the fused slice sums
int jch_nipals_lazy_capacity(int ldr, int q);   // 0: shape outside the lazy kernels' envelope
int32_t jch_launch_sweep_lazy(jch_ctx *ctx, const double *Xr, int64_t n, int ldr, const double *d, const double *wvec,
                              const double *Yr, int qpad, double *tcol, double *zt, int ldz, int max_slices, int *nslice_out,
                              const double *pend_p, int npend, int npend_max, const double *tpend, int64_t tstride);
// next K = X'DY from the rows with all `npend` corrections applied (the newest is this LV's; its Y step uses cvec); the rows
// are written back only when `flush`
int32_t jch_launch_kpass_lazy(jch_ctx *ctx, double *Xr, int64_t n, int p, int ldr, double *Yr, int qpad, int q, const double *d,
                              const double *pend_p, int npend, int npend_max, const double *tpend, int64_t tstride,
                              const double *cvec, bool flush, double *Knext);
int32_t jch_launch_deflate(jch_ctx *ctx, double *Xr, int64_t n, int p, int ldr, double *Yr, int qpad, int q,
                           const double *d, const double *tcol, const double *zpc /*[ldr + qpad]: zp then c*/,
                           double *Knext /*[p][qpad] or null*/);
// smallstate.hip
struct jch_small {  // device-resident replicated small state of one fit
    double *K;      // [p][qpad]
    double *w, *r;  // [ldr]
    double *P, *R, *W;  // [nlv][p]   (== Julia's p x nlv column-major)
    double *C;          // [nlv][q]
    double *TT;         // [nlv]
    double *Z;          // [nlv][q]   Z = P'K of the finished LVs (fast small-state path)
    double *zt;         // [JCH_ZT_SLICES][ldz]  reduced sweep output slices: zp, tt, (c_raw); ldz = ldr + 1 + qpad (+pad)
    double *zpc;        // [ldr + qpad]      plsnipals: zp/tt, c/tt
    double *mom, *scl;  // [p+q]
    double *mshift;     // [ldr] raw mode: means - pivot (the stored rows are x - pivot); null otherwise
    double *rs;         // [ldr] raw mode with scaling: r / xscales, the vector the sweep multiplies the unscaled rows with; null otherwise
    double *hdr;        // [4]
    int variant;        // 0: algorithm #1 (zt holds [zp, tt] from the sweep); 1: algorithm #2 (zt = G r, tt = r'zp computed here);
                        // 2: algorithm #1 in raw mode (uncentred row copy: zt holds [zp_raw, tt, st], zp = zp_raw - mom * st)
    double *dbg;        // [nlv + 1] diagnostics (JCH_LV_DEBUG): Jacobi sweeps per LV; may be null
    double *niter;      // [nlv] plswold: inner iterations per LV (src/plswold.jl:93); null otherwise
    double *kr;         // [16] split small-state path (smallstate_split.hip): K' r of the current LV; null otherwise
    double *gpart;      // [blocks][gld] split path: per-block partials of K_new'K_new, zp'K_new and P_i'K_new; null otherwise
    unsigned *lvctr;    // split path, merged kernel: arrival counter of the fit's blocks (zeroed when a fit starts); null otherwise
};

int32_t jch_launch_lv_update(jch_ctx *ctx, const jch_small &s, int p, int q, int qpad, int ldr, int a /*-1: init*/,
                             int nlv, int algo /*0 plskern, 1 plsnipals*/, int nslice, int ldz, bool fast, bool fuse_p2p = false,
                             const double *bf_src = nullptr, int bf_ld = 0, int bf_ldr = 0, double tol = 0.0, int maxit = 0);
size_t jch_lv_fast_lds_bytes(int p, int q, int qpad, int ldr, int nlv);
int32_t jch_launch_lv_update_fast(jch_ctx *ctx, const jch_small &s, int p, int q, int qpad, int ldr, int a, int nlv, int algo,
                                  int do_a, int do_b, int nslice, int ldz, bool fuse_p2p = false, const double *bf_src = nullptr,
                                  int bf_ld = 0, int bf_ldr = 0);
// smallstate_split.hip: the per-LV step of the plskern-shaped loop as a p-parallel kernel + a single-workgroup kernel
int jch_lv_split_blocks(int p);
int jch_lv_split_gld(int nlv);
size_t jch_lv_split_doubles(int p, int nlv);
int32_t jch_lv_split_begin_fit(jch_ctx *ctx, jch_small &s, double *gbuf, int p, int nlv);
size_t jch_lv_solve_lds_bytes(int p, int q, int ldr, int nlv);
int32_t jch_launch_lv_split(jch_ctx *ctx, const jch_small &s, int p, int q, int ldr, int a, int nlv, const double *part, int nb,
                            int ldpart, int itt, int ist /*< 0: none*/, int mode /*0 centred copy, 1 f64 raw mode, 2 bf16*/, bool solve,
                            bool fuse_p2p = false /*the inbox all-reduce of the LOCAL partial rows inside the p-parallel kernel, block by block*/);
bool jch_lv_split_p2p_ok(const jch_ctx *ctx, int p);
struct p2p_dev;
void jch_p2p_next(jch_ctx *ctx, p2p_dev *out);   // p2p.hip: device view of the inbox transport for the next epoch
int32_t jch_launch_nipals_R(jch_ctx *ctx, const jch_small &s, int p, int nlv);
// siblings.hip (plssimp / plsrosa / plswold small-state kernels)
bool jch_sibling_supported(int p, int q, int ldr, int nlv);
int32_t jch_launch_lv_update_simp(jch_ctx *ctx, const jch_small &s, int p, int q, int ldr, int a /*-1: init*/, int nlv, int nslice, int ldz);
int32_t jch_launch_wold_b(jch_ctx *ctx, const jch_small &s, int p, int q, int ldr, int a, int nlv, double tol, int maxit);
int32_t jch_launch_rosa_orthw(jch_ctx *ctx, double *W, int p, int nlv);
int32_t jch_launch_ydeflate_all(jch_ctx *ctx, double *Yc, int64_t ldy, const double *T, int64_t n, const double *C, int q, int nlv);
// gemm.hip
int32_t jch_launch_affine_gemm(jch_ctx *ctx, const double *Xc, int64_t m, int p, int64_t ldx, const double *Bs /*[p][kpad] scaled*/,
                               int k, int kpad, const double *bias /*[kpad]*/, double *out, int64_t ldo);
int32_t jch_launch_weighted_ss(jch_ctx *ctx, const double *Xc, int64_t n, int p, int64_t ldx, const double *d,
                               const double *shift, const double *iscale, double *out1);
// bf16.hip
int32_t jch_fit_plskern_bf16(jch_ctx *ctx, const jch_pls_desc &d, const void *Xv, int64_t ldx, const void *Yv, int64_t ldy,
                             const double *wdev, double *dn, double *Tdev, jch_small &s, int ldr_small, int qpad, int ldz,
                             bool fast, int *nlv_out);
// kern2.hip (opt-in kernel algorithm #2)
int32_t jch_launch_syrk(jch_ctx *ctx, const double *Xr, int64_t n, int p, int ldr, const double *d, double *G, int ldg);
int32_t jch_launch_gmatvec(jch_ctx *ctx, const double *G, int ldg, int p, int ldr, const double *r, double *out);
int32_t jch_launch_scores(jch_ctx *ctx, const double *Xr, int64_t n, int p, int ldr, const double *Rm, int nlv, double *T);
// util.hip
int32_t jch_launch_fill(jch_ctx *ctx, double *out, int64_t n, int64_t p, int64_t ld, int64_t row0, int64_t n_total,
                        uint64_t seed);

// ---- device helpers ---------------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ double jch_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Sum over a block of NT threads (NT multiple of 64, <= 1024); result valid in every thread.
// scratch: >= NT/64 doubles of LDS.  Deterministic (fixed tree).
template <int NT>
__device__ __forceinline__ double jch_block_sum(double v, double *scratch)
{
    v = jch_wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wv] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) s += scratch[i];
    return s;
}
#endif
