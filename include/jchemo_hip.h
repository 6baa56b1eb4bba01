/*
 * jchemo_hip.h — C ABI of libjchemo_hip.so: the MI355X (gfx950) PLS regression engine behind
 * Jchemo.jl's plskern / plsnipals hot path.
 *
 * The reference (/root/reference, Jchemo.jl v0.1.23) has NO FFI boundary: its "operator API" is the
 * Julia call shape  fun(X, Y[, weights]; nlv, scal=false) -> ::Plsr  plus the generics
 * transform / coef / predict / summary.  Each entry point below names the reference lines it
 * replaces; INTEGRATION.md shows the Julia `ccall` stubs (and the ctypes stubs used by the tests).
 *
 * Conventions
 *   - every function returns an int32 status: 0 = OK, <0 = error (codes below); the message is
 *     available from jch_last_error().  Nothing throws or aborts across the boundary.
 *   - matrices are COLUMN-MAJOR float64 exactly as Julia stores them (element (i,j) at
 *     base[i + j*ld]); `ld` is in elements.
 *   - the caller owns every buffer it passes; the library keeps no host pointer after return and
 *     owns only device workspace inside the ctx (re-used across calls).
 *   - a ctx is bound to ONE GPU and one HIP stream and is not thread-safe.  All calls are blocking
 *     (they return after the ctx stream has drained).
 *   - multi-GPU = one process (one ctx) per GPU, rows of X/Y/weights/T sharded by rows; the ctx is
 *     joined to an RCCL communicator with jch_ctx_comm_init and every rank makes the same call with
 *     its own shard.  All p x q-and-smaller results are replicated (bit-identical) on every rank.
 */
#ifndef JCHEMO_HIP_H
#define JCHEMO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JCH_VERSION 108 /* 0.1.8: jch_ctx_set_profiling(ctx, N > 1) samples the sweeps, JCH_COUNTER_SWEEPS_TIMED; 0.1.7: JCH_REUSE_XCOPY, JCH_COUNTER_XCOPY_REUSED; 0.1.6: + jch_score_sums_lv, jch_predict over an nlv range as running sums over the scores; 0.1.5: screened kNN (JCH_COUNTER_KNN_SCREENED / _REDONE); 0.1.4: no shape limits (generic lwplsr / small-state paths), JCH_NIPALS_ONE_PASS, JCH_COUNTER_LOCW_REFITS; 0.1.3: + jch_lwplsr_add_query_map (0.1.2: collective fields of jch_profile, jch_ctx_allreduce_probe, jch_lwplsr_prepare / _release) */

#if defined(JCH_BUILD)
#define JCH_API __attribute__((visibility("default")))
#else
#define JCH_API
#endif

/* status codes */
#define JCH_OK 0
#define JCH_EINVAL (-1) /* bad argument / unsupported shape */
#define JCH_EHIP (-2)   /* HIP runtime error (message has the HIP string) */
#define JCH_ERCCL (-3)  /* RCCL error or RCCL not loadable */
#define JCH_ENOMEM (-4) /* device or host allocation failed */
#define JCH_ENODEV (-5) /* no usable gfx950 device */

/* where the n-sized arrays (X, Y, weights, T, weights_norm) of a call live */
#define JCH_LOC_HOST 0   /* host pointers: the library stages them through the device */
#define JCH_LOC_DEVICE 1 /* device pointers on the ctx's GPU (e.g. an AMDGPU.jl ROCArray / torch tensor) */

/* storage type of X / Y handed to a fit */
#define JCH_F64 0
#define JCH_BF16 1 /* bf16 storage (BASELINE config 3); device-resident only.  jch_plskern_fit (p <= 2048): bf16-resident kernels, fp32 row
                      arithmetic, fp64 small state; every other fit: the inputs are widened exactly to Float64 on the device and the
                      Float64 path runs (same contract — the Float64 algorithm on the bf16-rounded inputs —, no bandwidth saving) */

typedef struct jch_ctx jch_ctx; /* opaque: device, stream, RCCL communicator, workspace pool */

JCH_API int32_t jch_version(void);

/* Create a context on HIP device `device_id`.  `stream` is a hipStream_t to launch on (e.g.
 * torch.cuda.current_stream().cuda_stream) or NULL to let the ctx create its own. */
JCH_API int32_t jch_ctx_create(jch_ctx **out, int32_t device_id, void *stream, uint32_t flags);
JCH_API int32_t jch_ctx_destroy(jch_ctx *ctx);
/* Message of the last failing call on ctx (ctx == NULL: last jch_ctx_create failure).  Valid until
 * the next call on the same ctx. */
JCH_API const char *jch_last_error(const jch_ctx *ctx);

/* ---- row-sharded multi-GPU (RCCL over xGMI) ------------------------------------
 * Rank 0 fills a 128-byte id with jch_comm_unique_id, the host side broadcasts it (the Python mirror
 * uses torch.distributed, the Julia wrapper MPI/Distributed), every rank calls jch_ctx_comm_init. */
JCH_API int32_t jch_comm_unique_id(void *uid128);
JCH_API int32_t jch_ctx_comm_init(jch_ctx *ctx, const void *uid128, int32_t rank, int32_t nranks);
JCH_API int32_t jch_ctx_comm_info(const jch_ctx *ctx, int32_t *rank, int32_t *nranks);

/* ---- P2P inbox transport for the latency-bound all-reduces (the per-LV [zp, tt] message is 4 KB) -----------------
 * Each rank owns an inbox in fine-grained device memory; the other ranks' processes map it through a HIP IPC handle
 * and store their contribution + an epoch flag straight into it over xGMI; one single-workgroup kernel per rank and
 * all-reduce, sums in rank order (bit-identical on every rank), bounded waits (a lost peer yields JCH_ERCCL, never a
 * hang).  Set-up, driven by the host side that already exchanged the RCCL id:
 *   1. every rank: jch_ctx_p2p_export(ctx, nranks, handle64)             -> 64-byte IPC handle of its inbox
 *   2. all-gather the handles (torch.distributed / MPI), every rank: jch_ctx_p2p_import(ctx, handles, rank, nranks, 0)
 *      — maps the peers and runs a collective self-test; returns JCH_ERCCL on the ranks where it failed
 *   3. agree (min over ranks of "import succeeded") and call jch_ctx_p2p_enable(ctx, 1) everywhere, or leave it off:
 *      RCCL (jch_ctx_comm_init) remains the transport for large messages and the fallback.
 * `flags` is reserved (0).  nranks <= 16. */
JCH_API int32_t jch_ctx_p2p_export(jch_ctx *ctx, int32_t nranks, void *handle64);
JCH_API int32_t jch_ctx_p2p_import(jch_ctx *ctx, const void *handles, int32_t rank, int32_t nranks, uint32_t flags);
JCH_API int32_t jch_ctx_p2p_enable(jch_ctx *ctx, int32_t on);

/* Loopback communicator — TEST HARNESS for the row-sharded path on a one-GPU box (RCCL refuses two ranks on one device):
 * the "ranks" are host threads of ONE process, each with its own ctx on the same GPU; all-reduces are staged through
 * host memory in rank order (bit-identical sums on every rank, like the RCCL path).  Every rank thread must make the
 * same sequence of library calls.  Not a production transport. */
JCH_API int32_t jch_loopback_group_create(int32_t nranks, void **group_out);
JCH_API int32_t jch_loopback_group_destroy(void *group);
JCH_API int32_t jch_ctx_comm_init_loopback(jch_ctx *ctx, void *group, int32_t rank);

typedef struct jch_pls_desc {
    int64_t n;       /* rows held by THIS rank (all rows when single-GPU) */
    int64_t p;       /* columns of X */
    int64_t q;       /* columns of Y (>= 1; q <= 16 runs the LDS-resident small-state kernels, larger q a generic one) */
    int32_t nlv;     /* requested LVs; clamped to min(n_total, p, nlv) like plskern.jl:116 */
    int32_t scal;    /* 0/1: scale columns by their weighted uncorrected std (plskern.jl:123-126) */
    int32_t dtype;   /* JCH_F64 | JCH_BF16 */
    int32_t loc;     /* JCH_LOC_HOST | JCH_LOC_DEVICE for X, Y, weights, T, weights_norm */
    int32_t inplace; /* 1 = `plskern!` / `plsnipals!` semantics: X and Y are overwritten with their
                        centred/scaled (plsnipals: and deflated) versions; 0 = `plskern` (inputs untouched,
                        the reference's copy at plskern.jl:108 never materialises) */
    int32_t reserved; /* option bits.  0 = the reference's algorithm (improved kernel #1, one sweep over X per LV);
                         bit 0 (1) = OPT-IN kernel algorithm #2 (X'DX once, no pass over X and no collective in the LV
                         loop; plskern, q <= 16, p <= 2048): same results up to rounding;
                         JCH_WOLD_REF_ZERO_WEIGHT_NAN (jch_plswold_fit only): see below;
                         JCH_NIPALS_ONE_PASS (jch_plsnipals_fit / jch_plswold_fit): see below;
                         JCH_REUSE_XCOPY (any fit): X is unchanged since the previous fit on this ctx, see below */
} jch_pls_desc;
/* jch_plswold_fit: give the rows with weight 0 NaN scores, as the reference does (src/plswold.jl:107 divides by sqrt(w) = 0).
 * Default (bit clear): finite scores t_i = x_i' r for those rows — what a cross-validation fold with zero weights on its
 * held-out rows needs (gridcvlv). */
#define JCH_WOLD_REF_ZERO_WEIGHT_NAN 2
/* jch_plsnipals_fit / jch_plswold_fit, OPT-IN (never the default: it is not what the reference computes, src/plsnipals.jl:71 recomputes
 * X'DY from the deflated matrices every LV): ONE pass over X per LV.  The next kernel matrix follows from the exact identity
 * K_{a+1} = (X - t p')'D(Y - t c') = K_a - zp_raw c_raw' / tt (t is D-orthogonal to its own residuals), c_raw = Y'Dt is taken
 * against the UNdeflated Y (equal for the same reason), and the second read of X per LV disappears; the deflated rows are still
 * written back every few LVs (the postponed write-back needs q <= 16, p <= 2048, inplace = 0, Float64: JCH_EINVAL otherwise).
 * Same results up to rounding (gate of the tests: 1e-9 against the default path on well-conditioned LVs). */
#define JCH_NIPALS_ONE_PASS 4
/* Any fit: the caller PROMISES that X (pointer, shape, leading dimension and contents) is what the previous fit on this ctx was
 * given — the folds of a cross-validation (`gridcvlv`, src/gridcv.jl:200-208: the same X with other rows held out), the combinations
 * of a parameter grid.  A Float64 plskern-shaped fit whose predecessor left its row-major working copy in the workspace then skips
 * the staging of X (host arrays) and the transposing copy, and takes X'D[Yc | 1] from one streaming read of that copy (q <= 11,
 * p <= 512; same results up to the summation order of X'DY).  In every other situation the bit is ignored.  Y and the weights may
 * change freely; if X changed, the results are those of the OLD X. */
#define JCH_REUSE_XCOPY 8

/*
 * jch_plskern_fit — replaces `plskern!` / `plskern` (src/plskern.jl:106-178): weight normalisation
 * (utility.jl:715-723), weighted column means / stds (utility.jl:195,264,314-323), centring/scaling
 * (utility.jl:76-81,482-487), XtY = X'DY (plskern.jl:131-132) and the per-LV loop (plskern.jl:149-175).
 *   X  n x p (ldx >= n), Y  n x q (ldy >= n), weights n or NULL (= ones)          [desc->loc]
 *   T  n x nlv (ld n), weights_norm n                                              [desc->loc]
 *   P, R, W  p x nlv (ld p); C  q x nlv (ld q); TT nlv; xmeans, xscales p; ymeans, yscales q   [HOST]
 *   nlv_out: the clamped number of LVs actually computed (columns filled in T/P/R/W/C/TT).
 * Any output pointer may be NULL to skip it.  Sign of each LV (w, r, t, P, C columns flip together,
 * SURVEY F3) is fixed by: the largest-|.| component of the dominant right singular vector is positive.
 */
JCH_API int32_t jch_plskern_fit(jch_ctx *ctx, const jch_pls_desc *desc, void *X, int64_t ldx, void *Y, int64_t ldy,
                        const double *weights, double *T, double *P, double *R, double *W, double *C,
                        double *TT, double *xmeans, double *xscales, double *ymeans, double *yscales,
                        double *weights_norm, int32_t *nlv_out);

/* jch_plsnipals_fit — replaces `plsnipals!` / `plsnipals` (src/plsnipals.jl:31-97); same signature.
 * With inplace = 1, X and Y return centred AND deflated (plsnipals.jl:86-87). */
JCH_API int32_t jch_plsnipals_fit(jch_ctx *ctx, const jch_pls_desc *desc, void *X, int64_t ldx, void *Y, int64_t ldy,
                          const double *weights, double *T, double *P, double *R, double *W, double *C,
                          double *TT, double *xmeans, double *xscales, double *ymeans, double *yscales,
                          double *weights_norm, int32_t *nlv_out);

/* jch_plskern_fit_scaled — jch_plskern_fit with CALLER-SUPPLIED column divisors instead of `scal`: X is centred by its
 * weighted means and divided by xscales_in (p, HOST), Y by yscales_in (q, HOST; NULL = ones); desc->scal is ignored and
 * the divisors are echoed in xscales / yscales.  This is what multiblock PLSR needs (src/mbplsr.jl:77-113: per-block
 * column scales times one scalar per block, then plskern with scal = false on the concatenated blocks).  Float64. */
JCH_API int32_t jch_plskern_fit_scaled(jch_ctx *ctx, const jch_pls_desc *desc, void *X, int64_t ldx, void *Y, int64_t ldy,
                        const double *weights, const double *xscales_in, const double *yscales_in, double *T, double *P,
                        double *R, double *W, double *C, double *TT, double *xmeans, double *xscales, double *ymeans,
                        double *yscales, double *weights_norm, int32_t *nlv_out);

/* jch_col_stats — weighted column means and (stds != NULL) uncorrected standard deviations: `colmean`, `colstd`
 * (src/utility.jl:193-195,312-323) on their own.  X n x p [loc], weights n or NULL [loc]; means, stds (p) HOST. */
JCH_API int32_t jch_col_stats(jch_ctx *ctx, int32_t loc, const double *X, int64_t n, int64_t p, int64_t ldx, const double *weights,
                      double *means, double *stds);

/* ---- sibling algorithms: same row kernels, different small state --------------------------------
 * jch_plssimp_fit — `plssimp!` / `plssimp` (src/plssimp.jl:22-88, SIMPLS with un-normed scores); W is returned equal to R
 *   (src/plssimp.jl:85-87).  One fused sweep over X per LV, like plskern.
 * jch_plsrosa_fit — `plsrosa!` / `plsrosa` (src/plsrosa.jl:26-96): X never deflated, Y deflated (inplace = 1 hands back
 *   the centred X and the deflated Y, :87), W re-orthonormalised (:77-79), R = W inv(P'W) (:94).  One fused sweep per LV:
 *   the score orthogonalisation of :75-76 is applied to the weight vector before the sweep (t = X r), which is the same
 *   vector in exact arithmetic.
 * jch_plswold_fit — `plswold!` / `plswold` (src/plswold.jl:30-111): NIPALS with the inner power iteration; `tol` and
 *   `maxit` as the reference's keywords (defaults sqrt(eps), 200); niter (nlv, HOST, as Float64 like :71) receives the
 *   number of inner passes per LV.  The reference seeds each LV's first convergence check with `rand(p)` (:78), which
 *   can never pass; that check is skipped here.  Rows with zero weight get finite scores by default (the reference divides
 *   by sqrt(w) = 0 at :107 and returns NaN for them; desc->reserved |= JCH_WOLD_REF_ZERO_WEIGHT_NAN reproduces that).
 *   inplace = 1 hands back X, Y deflated AND carrying the row metric sqrt(w) (:57-58).
 * plssimp / plswold run their LDS-resident small-state kernels when q <= 16, p <= 2048 and the p x q state fits in LDS,
 * and a generic kernel (state in global memory; any p, q and nlv — for q > 64 its q x q eigen-solver matrices live in
 * global memory too) otherwise; Float64 only. */
JCH_API int32_t jch_plssimp_fit(jch_ctx *ctx, const jch_pls_desc *desc, void *X, int64_t ldx, void *Y, int64_t ldy,
                        const double *weights, double *T, double *P, double *R, double *W, double *C,
                        double *TT, double *xmeans, double *xscales, double *ymeans, double *yscales,
                        double *weights_norm, int32_t *nlv_out);
JCH_API int32_t jch_plsrosa_fit(jch_ctx *ctx, const jch_pls_desc *desc, void *X, int64_t ldx, void *Y, int64_t ldy,
                        const double *weights, double *T, double *P, double *R, double *W, double *C,
                        double *TT, double *xmeans, double *xscales, double *ymeans, double *yscales,
                        double *weights_norm, int32_t *nlv_out);
JCH_API int32_t jch_plswold_fit(jch_ctx *ctx, const jch_pls_desc *desc, void *X, int64_t ldx, void *Y, int64_t ldy,
                        const double *weights, double tol, int32_t maxit, double *T, double *P, double *R, double *W,
                        double *C, double *TT, double *xmeans, double *xscales, double *ymeans, double *yscales,
                        double *weights_norm, double *niter, int32_t *nlv_out);

/*
 * jch_affine_gemm — out (m x k, ld ldo) = ((X - 1*shift') * diag(1/scale)) * B + 1*bias'
 * the single device primitive behind `transform` (src/plskern.jl:187-195: shift = xmeans, scale =
 * xscales, B = R[:,1:k], bias = 0) and `predict` (src/plskern.jl:226-238 via coef :207-217:
 * shift = 0, scale = 1, B = [B_k0 | B_k1 | ...], bias = [int_k0 | ...]; one pass over X for the whole
 * nlv range instead of one GEMM per value).
 *   X m x p (ldx) and out [loc]; shift, scale (p, may be NULL), B (p x k, ld p), bias (k, may be NULL) HOST.
 */
JCH_API int32_t jch_affine_gemm(jch_ctx *ctx, int32_t loc, const double *X, int64_t m, int64_t p, int64_t ldx,
                        const double *shift, const double *scale, const double *B, int64_t k,
                        const double *bias, double *out, int64_t ldo);

/* jch_transform — `transform(object::Plsr, X; nlv)` (src/plskern.jl:187-195): T (m x nlv, ld ldt) [loc] =
 * cscale(X, xmeans, xscales) * R[:, 1:nlv].  X m x p [loc]; xmeans, xscales (p), R (p x >= nlv, ld p) HOST. */
JCH_API int32_t jch_transform(jch_ctx *ctx, int32_t loc, const double *X, int64_t m, int64_t p, int64_t ldx,
                      const double *xmeans, const double *xscales, const double *R, int32_t nlv, double *T, int64_t ldt);

/* jch_predict — `predict(object::Plsr, X; nlv)` (src/plskern.jl:226-238 through coef, :207-217) for every nlv in
 * [nlv_lo, nlv_hi] (0 = intercept only) in ONE pass over X: pred (m x q*(nlv_hi-nlv_lo+1), ld ldo) [loc]; block b =
 * columns b*q .. b*q+q-1 = prediction with nlv_lo + b LVs.  Model pieces (xmeans, xscales p; ymeans, yscales q;
 * R p x nlv_fit ld p; C q x nlv_fit ld q) HOST, nlv_hi <= nlv_fit is the caller's responsibility (the reference
 * clamps at :228).  One or two levels: one GEMM with the levels' coefficient matrices side by side.  Three or more on a long input
 * (m >= 4096): the scores X_c R once (m x nlv_hi), then the blocks as running sums over the score columns,
 * pred_a = pred_{a-1} + t_a (c_a .* yscales)' — X_c B_a = T_a C_a' — written in one streaming pass (cfg2 size, nlv = 0..25: 1.15 ms). */
JCH_API int32_t jch_predict(jch_ctx *ctx, int32_t loc, const double *X, int64_t m, int64_t p, int64_t ldx, const double *xmeans,
                    const double *xscales, const double *ymeans, const double *yscales, const double *R, const double *C,
                    int64_t q, int32_t nlv_lo, int32_t nlv_hi, double *pred, int64_t ldo);

/* jch_weighted_ss — sum_i d_i * || (x_i - shift) / scale ||^2 : the `sstot` of `summary`
 * (src/plskern.jl:250-251).  X n x p and d (n) [loc]; shift/scale HOST; result HOST.  With a
 * communicator the result is the sum over all ranks' shards. */
JCH_API int32_t jch_weighted_ss(jch_ctx *ctx, int32_t loc, const double *X, int64_t n, int64_t p, int64_t ldx,
                        const double *d, const double *shift, const double *scale, double *sstot);

/*
 * jch_lwplsr_predict — the prediction path of kNN-LWPLSR, batched over the m queries: replaces getknn
 * (src/getknn.jl:29-57), the wdist weight loop (src/lwplsr.jl:152-159, src/wdist.jl:64-75) and locwlv
 * (src/locwlv.jl:9-48, `Threads.@threads` over queries: one weighted plskern + 1-row predict per query).
 *   Xtrain n x p, Ytrain n x q, Xq m x p                                   [loc], column-major
 *   Ztrain n x dd, Zq m x dd: the space the neighbours are searched in (global PLS scores, already whitened by
 *     the host side for metric = "mahal": scores * inv(chol(cov).U), src/getknn.jl:37-49)      [loc]
 *   k neighbours (clamped to n), h / tol: weight shape and floor; scal; nlv range nlv_lo..nlv_hi (contiguous)
 *   pred  m x le x q (le = nlv_hi - nlv_lo + 1), query-major: pred[(i*le + a)*q + y]              [HOST]
 *   ind_out (m x k, 0-based, row-major), dist_out, w_out (m x k): optional                       [HOST]
 * No shape limits (the reference has none: src/getknn.jl:29-57, src/locwlv.jl:9-48).  Inside the batched kernels' envelope —
 * k <= 768 for the kNN scan; p <= 2048, q <= 16, nlv <= 48 and 150 KB of LDS for the local fits — one launch serves all m
 * queries: for k <= 208, q <= 8 and wide rows the fit runs in NEIGHBOUR space (the k x k Gram matrix of the gathered rows, built
 * on the matrix cores in one pass and held in registers: lwplsr_kspace.hip; same T, C and predictions up to rounding), otherwise
 * one workgroup per query sweeps the k x p slab once per LV.  Outside it the generic paths of lwplsr_generic.hip run: an exact
 * per-query selection of the k smallest distances (radix select + sort in global memory, the same distance expression: the
 * same neighbours, order and distance bits), and one jch_plskern_fit + jch_predict per query on the gathered rows — the
 * reference's own schedule (src/locwlv.jl:18-39); slower, never absent.
 * The constant-y shortcut of src/locwlv.jl:25-28 applies to q == 1 only, as in the reference.
 * Neighbours at equal distance are ordered by their row index.
 */
JCH_API int32_t jch_lwplsr_predict(jch_ctx *ctx, int32_t loc, const double *Xtrain, int64_t n, int64_t p, int64_t ldx,
                                   const double *Ytrain, int64_t q, int64_t ldy, const double *Ztrain, int64_t ldzt,
                                   const double *Zq, int64_t ldzq, int64_t dd, const double *Xq, int64_t m, int64_t ldxq,
                                   int32_t k, double h, double tol, int32_t scal, int32_t nlv_lo, int32_t nlv_hi,
                                   double *pred, int32_t *ind_out, double *dist_out, double *w_out);

/* ---- persistent model state of kNN-LWPLSR: the reference's `Lwplsr` object (src/lwplsr.jl:1-12) is fitted once
 * (`lwplsr`, :114-131) and predicted from many times (`predict`, :134-166).  jch_lwplsr_prepare keeps the model-constant
 * device data — the row-major copy of Xtrain the neighbour gathers read, Ytrain, the (whitened) training scores Ztrain — in
 * a handle; jch_lwplsr_predict_prepared is jch_lwplsr_predict without those three arguments and without their per-call
 * copies; jch_lwplsr_release frees the handle.  The handle belongs to the ctx's device; the inputs of prepare are not
 * referenced after it returns.  Same paths and errors as jch_lwplsr_predict. */
typedef struct jch_lwplsr_model jch_lwplsr_model;
JCH_API int32_t jch_lwplsr_prepare(jch_ctx *ctx, int32_t loc, const double *Xtrain, int64_t n, int64_t p, int64_t ldx,
                                   const double *Ytrain, int64_t q, int64_t ldy, const double *Ztrain, int64_t ldzt, int64_t dd,
                                   jch_lwplsr_model **model_out);
JCH_API int32_t jch_lwplsr_predict_prepared(jch_ctx *ctx, const jch_lwplsr_model *model, int32_t loc, const double *Zq, int64_t ldzq,
                                            const double *Xq, int64_t m, int64_t ldxq, int32_t k, double h, double tol, int32_t scal,
                                            int32_t nlv_lo, int32_t nlv_hi, double *pred, int32_t *ind_out, double *dist_out,
                                            double *w_out);
JCH_API int32_t jch_lwplsr_release(jch_ctx *ctx, jch_lwplsr_model *model);
/* The map that takes a query block to the neighbour-search space — the reference's `transform(object.fm, X)`
 * (src/lwplsr.jl:139-151) followed by getknn's whitening (src/getknn.jl:37-49) — as a chain of affine stages
 * Z <- ((Z - shift) ./ scale) B + bias (the arithmetic of jch_affine_gemm; shift, scale, bias may be NULL; B is p_in x k_out,
 * column-major, all on the HOST), kept on the device with the model.  Stage 1 takes the model's p columns, every later stage
 * the previous one's k_out, the last one must deliver dd columns; at most 4 stages.  With a map in place
 * jch_lwplsr_predict_prepared accepts Zq = NULL and computes the query scores itself: same numbers as two jch_affine_gemm
 * calls, without their uploads and synchronisations. */
JCH_API int32_t jch_lwplsr_add_query_map(jch_ctx *ctx, jch_lwplsr_model *model, const double *shift, const double *scale,
                                         const double *B, int64_t p_in, int64_t k_out, const double *bias);

/* jch_weighted_cov — S = (A - 1 mu')' D (A - 1 mu') (d x d), D = diag(weights / sum); weights NULL = ones:
 * `Statistics.cov(Xtrain, corrected = false)` of getknn's Mahalanobis branch (src/getknn.jl:38; with nlvdis = 0 the
 * reference whitens the raw X, src/lwplsr.jl:21: d = p).  d <= 64: the X'DY kernels of the fit with Y = A; wider: the centred
 * row-major copy + the tiled MFMA SYRK of the opt-in algorithm #2.  A n x d [loc]; S (column-major d x d) and mu (d, may be
 * NULL) HOST. */
JCH_API int32_t jch_weighted_cov(jch_ctx *ctx, int32_t loc, const double *A, int64_t n, int64_t d, int64_t lda,
                                 const double *weights, double *S, double *mu);

/* jch_score_sums — sufficient statistics of prediction scores over the rows selected by `mask` (NULL = all rows), per
 * prediction column c = level * q + k:  sums[c*6 + 0..5] = { sum e, sum e^2, sum y e, sum y, sum y^2, row count },
 * e = y - pred.  msep / rmsep / ssr / bias / r2 / cor2 (src/scores.jl:25-32,54-62,155-158,190-196,268,426-429) follow on
 * the host; this is what gridscorelv / gridcvlv (src/gridscore.jl:167-221, src/gridcv.jl:187-228) evaluate per nlv.
 *   Pred m x ncol (ncol multiple of q), Y m x q, mask m [loc]; sums ncol x 6 HOST. */
JCH_API int32_t jch_score_sums(jch_ctx *ctx, int32_t loc, const double *Pred, int64_t m, int64_t ncol, int64_t ldp,
                               const double *Y, int64_t q, int64_t ldy, const double *mask, double *sums);

/* jch_score_sums_lv — the statistics of jch_score_sums for the predictions with nlv = nlv_lo..nlv_hi latent variables, straight
 * from the rows' scores: pred_a = ymeans + sum_{l <= a} t_l (c_l .* yscales)' (src/plskern.jl:207-217, 226-238 applied to
 * `transform(object, X)`), accumulated level by level in registers — the m x (levels q) prediction matrix that gridscorelv /
 * gridcvlv (src/gridscore.jl:196-216, src/gridcv.jl:206-224) would score never exists.  Levels beyond kfit repeat level kfit (the
 * reference clamps, src/plskern.jl:228).
 *   T m x kfit (scores of the rows, ld ldt), Y m x q, mask m (may be NULL) [loc]; C q x kfit (ld q), ymeans, yscales (q; NULL = 0 / 1)
 *   HOST; sums (nlv_hi - nlv_lo + 1) q x 6 HOST, laid out like jch_score_sums on the level-major prediction matrix. */
JCH_API int32_t jch_score_sums_lv(jch_ctx *ctx, int32_t loc, const double *T, int64_t m, int64_t kfit, int64_t ldt, const double *C,
                                  const double *ymeans, const double *yscales, const double *Y, int64_t q, int64_t ldy, const double *mask,
                                  int32_t nlv_lo, int32_t nlv_hi, double *sums);

/* ---- harness utilities (bench / tests) ---------------------------------------------------------- */
/* Fill device matrix out (n x p, column-major ld) with rows [row0,row0+n) of the n_total x p matrix
 * whose element (i,j) is splitmix64-uniform(seed, i + j*n_total) — the README's `rand(n,p)` stand-in
 * (README.md:79-94), identical to oracle/plsr_oracle.py:splitmix64_uniform. */
JCH_API int32_t jch_fill_uniform(jch_ctx *ctx, double *dev_out, int64_t n, int64_t p, int64_t ld, int64_t row0,
                         int64_t n_total, uint64_t seed);

typedef struct jch_profile {
    double fit_ms;        /* device time of the last fit, first kernel -> last kernel (HIP events)   */
    double prologue_ms;   /* weights + means (+ var) + centre/transpose/XtY                           */
    double sweep_ms;      /* sum over LVs of the dominant kernel (fused sweep; plsnipals: sweep+deflate) */
    double smallstate_ms; /* sum over LVs of partial reduce + lv_update (+ all-reduce)                */
    int32_t sweep_launches;
    int32_t nlv;
    double sweep_bytes;   /* algorithmic bytes of ONE dominant-kernel launch          */
    /* ---- cross-GPU all-reduces of the last fit (all zero on one GPU).  RCCL / loopback / stand-alone inbox kernel: HIP
     * events around each call on the ctx stream (so the time includes waiting for the slowest rank); inbox fused into
     * the small-state kernel: wall_clock64 stamps inside that kernel, first peer store -> rank-ordered sum done.  The
     * fused time is part of smallstate_ms, the others sit between the kernels smallstate_ms spans: either way
     * smallstate_ms - collective_ms is the small-state kernels + launch gaps alone. */
    double collective_ms;          /* sum over the LV loop's all-reduces ([zp, tt] per LV; plsnipals: + K)      */
    double prologue_collective_ms; /* sum over the prologue's all-reduces (weights, moments / pivot, XtY)        */
    double collective_wait_ms;     /* inbox only: the part of collective_ms spent polling the peers' flags      */
    int32_t collective_calls;      /* all-reduces inside the LV loop                                            */
    int32_t collective_transport;  /* JCH_TRANSPORT_* of the LV loop's all-reduce                               */
} jch_profile;
#define JCH_TRANSPORT_NONE 0
#define JCH_TRANSPORT_RCCL 1
#define JCH_TRANSPORT_INBOX 2       /* stand-alone single-workgroup inbox kernel (p2p.hip)                       */
#define JCH_TRANSPORT_INBOX_FUSED 3 /* inbox exchange inside the small-state kernel (no launch of its own)       */
#define JCH_TRANSPORT_LOOPBACK 4    /* test harness                                                              */
/* Enable (1) / disable (0) per-kernel HIP-event timing of subsequent fits (adds event records on the
 * ctx stream, no host syncs inside the fit).  enable = N > 1: the plskern-shaped sweeps (Float64 p <= 1024 and bf16) are SAMPLED —
 * an event pair around every N-th launch only, counted across fits (an event record costs the stream about 3 us: 50 of them are
 * 1 % of a cfg2 fit and 6 % of a 125 k-row share); jch_profile then reports sweep_ms = the sampled launches' mean x the launches
 * made, sweep_launches = the launches made; JCH_COUNTER_SWEEPS_TIMED counts the launches actually bracketed. */
JCH_API int32_t jch_ctx_set_profiling(jch_ctx *ctx, int32_t enable);
JCH_API int32_t jch_ctx_get_profile(const jch_ctx *ctx, jch_profile *out);

/* jch_ctx_allreduce_probe — diagnostic: all-reduce (sum) the `count` doubles of `vec` (HOST, in/out) `iters` times
 * back to back through ONE named transport and report the average time per all-reduce in microseconds (HIP events on
 * the ctx stream; iteration 0 is a warm-up and excluded when iters > 1).  Every iteration starts from the caller's
 * values, so on return vec = the sum over ranks: a vector of ones comes back as the number of ranks the transport
 * actually reached.  Collective: every rank calls it with the same count / iters / transport.
 *   transport: JCH_TRANSPORT_RCCL (needs jch_ctx_comm_init), JCH_TRANSPORT_INBOX (needs a self-tested inbox; it need
 *   not be enabled), JCH_TRANSPORT_LOOPBACK, or JCH_TRANSPORT_NONE = whatever a fit would use for this message size.
 * This is how bench.py compares RCCL and the inbox on the per-LV message of the fit it times. */
JCH_API int32_t jch_ctx_allreduce_probe(jch_ctx *ctx, int32_t transport, double *vec, int64_t count, int32_t iters,
                                        double *avg_us);

/* Diagnostic counters of a ctx (cumulative since jch_ctx_create).  which = JCH_COUNTER_PIVOT_REFITS: fits whose one-pass
 * ("raw") prologue was repeated on the centred working copy because the sampled pivot turned out to be further than 64
 * sample standard deviations from a column mean (results are those of the centred formulation). */
#define JCH_COUNTER_PIVOT_REFITS 0
/* which = JCH_COUNTER_LOCW_REFITS: queries of jch_lwplsr_predict* that the neighbour-space local-fit kernel flagged as lying more
 * than 64 local standard deviations (along the offset) from the mean of their neighbours — possible because the neighbours are
 * chosen in the score space, not in p-space — and that were refitted by the per-query path (explicit centring). */
#define JCH_COUNTER_LOCW_REFITS 1
/* which = JCH_COUNTER_KNN_SCREENED: queries of jch_lwplsr_predict* whose neighbours were found by the screened search (all (row, query)
 * pairs on the bf16 matrix cores from two-piece operands, an error-bounded bar per query, exact Float64 distances for the survivors;
 * score spaces of <= 62 dimensions, k <= 768, n < 2^26 and enough rows for the bar to be tight — otherwise, and with
 * JCH_KNN_SCREEN=0 in the environment, the exact scan);
 * which = JCH_COUNTER_KNN_SCREEN_REDONE: those of them the screen could not settle (non-finite scores, or more rows within its
 * error bound of the k-th distance than a candidate list holds: ties on a lattice, neighbour distances far below the scores' norms)
 * and the exact scan redid.  Neighbours, their order, distances and weights do not depend on which search ran.  A prepared model
 * (jch_lwplsr_prepare) that sees more than a quarter of a call's queries redone stops screening. */
#define JCH_COUNTER_KNN_SCREENED 2
#define JCH_COUNTER_KNN_SCREEN_REDONE 3
/* which = JCH_COUNTER_XCOPY_REUSED: fits that honoured JCH_REUSE_XCOPY, i.e. took X'D[Yc | 1] from the previous fit's row-major copy */
#define JCH_COUNTER_XCOPY_REUSED 4
/* which = JCH_COUNTER_SWEEPS_TIMED: launches of the plskern-shaped sweep that were bracketed by HIP events (jch_ctx_set_profiling) */
#define JCH_COUNTER_SWEEPS_TIMED 5
JCH_API int32_t jch_ctx_get_counter(const jch_ctx *ctx, int32_t which, int64_t *out);

#ifdef __cplusplus
}
#endif
#endif /* JCHEMO_HIP_H */
