// Fit orchestration: jch_plskern_fit / jch_plsnipals_fit (include/jchemo_hip.h).
// Everything between the first and the last kernel of a fit is enqueued on ctx->stream with no host sync
// (single GPU); a multi-GPU fit syncs once in the prologue to learn the global row count.
#include <stdlib.h>
#include <time.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "jch_internal.h"

__global__ __launch_bounds__(256) void k_fill_const(double *__restrict__ v, int m, double c)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < m) v[j] = c;
}

// plswold with JCH_WOLD_REF_ZERO_WEIGHT_NAN: the reference's `Tx .= (1 ./ sqrtw) .* Tx` (src/plswold.jl:107) turns the scores of a
// zero-weight row into 0 * Inf = NaN
__global__ __launch_bounds__(256) void k_nan_zero_weight_rows(double *__restrict__ T, int64_t n, int nlv, const double *__restrict__ d)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && d[i] == 0.0)
        for (int a = 0; a < nlv; ++a) T[(size_t)i + (size_t)a * (size_t)n] = __builtin_nan("");
}

// bf16 -> f64, column by column (the values are exact): the inputs of a fit that has no bf16-resident kernels of its own
__global__ __launch_bounds__(256) void k_widen_bf16(const unsigned short *__restrict__ src, int64_t ld, int64_t n, int64_t cols, double *__restrict__ dst)
{
    const int64_t tot = n * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < tot; e += (int64_t)gridDim.x * 256) {
        const int64_t j = e / n, i = e - j * n;
        dst[e] = (double)__uint_as_float(((unsigned)src[(size_t)i + (size_t)j * (size_t)ld]) << 16);
    }
}

namespace {

struct fit_io {
    const jch_pls_desc *d;
    void *X; int64_t ldx; void *Y; int64_t ldy; const double *weights;
    double *T, *P, *R, *W, *C, *TT, *xmeans, *xscales, *ymeans, *yscales, *weights_norm;
    int32_t *nlv_out;
    double tol = 0.0; int maxit = 0; double *niter = nullptr;   // plswold only
    const double *xscales_in = nullptr, *yscales_in = nullptr;   // jch_plskern_fit_scaled: caller-supplied column divisors (host)
};

// algorithm codes of fit_impl
enum { ALGO_KERN = 0, ALGO_NIPALS = 1, ALGO_SIMP = 2, ALGO_ROSA = 3, ALGO_WOLD = 4 };

int32_t validate(jch_ctx *ctx, const fit_io &io, const char *who)
{
    const jch_pls_desc *d = io.d;
    if (!d) return jch_fail(ctx, JCH_EINVAL, "%s: desc is NULL", who);
    if (d->n < 1 || d->p < 1 || d->q < 1) return jch_fail(ctx, JCH_EINVAL, "%s: empty input (n=%lld p=%lld q=%lld)", who,
                                                         (long long)d->n, (long long)d->p, (long long)d->q);
    if (d->q > (1 << 12)) return jch_fail(ctx, JCH_EINVAL, "%s: q=%lld too large", who, (long long)d->q);
    if (d->p > (1 << 20)) return jch_fail(ctx, JCH_EINVAL, "%s: p=%lld too large", who, (long long)d->p);
    if (d->nlv < 1) return jch_fail(ctx, JCH_EINVAL, "%s: nlv=%d must be >= 1", who, d->nlv);
    if (d->dtype != JCH_F64 && d->dtype != JCH_BF16) return jch_fail(ctx, JCH_EINVAL, "%s: unknown dtype %d", who, d->dtype);
    if (d->dtype == JCH_BF16 && (d->loc != JCH_LOC_DEVICE || d->inplace))
        return jch_fail(ctx, JCH_EINVAL, "%s: bf16 storage needs device-resident inputs and inplace = 0", who);
    if (d->loc != JCH_LOC_HOST && d->loc != JCH_LOC_DEVICE) return jch_fail(ctx, JCH_EINVAL, "%s: bad loc %d", who, d->loc);
    if (!io.X || !io.Y) return jch_fail(ctx, JCH_EINVAL, "%s: X or Y is NULL", who);
    if (io.ldx < d->n || io.ldy < d->n) return jch_fail(ctx, JCH_EINVAL, "%s: ldx/ldy smaller than n", who);
    return JCH_OK;
}

struct carve {
    char *base; size_t off;
    double *take(size_t count) { double *p = (double *)(base + off); off += ((count * sizeof(double)) + 255) & ~(size_t)255; return p; }
};

int32_t h2d_matrix(jch_ctx *ctx, double *dst, const double *src, int64_t n, int64_t cols, int64_t ld)
{
    if (ld == n) JCH_HIP(ctx, hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n * cols, hipMemcpyHostToDevice, ctx->stream));
    else JCH_HIP(ctx, hipMemcpy2DAsync(dst, sizeof(double) * n, src, sizeof(double) * ld, sizeof(double) * n, cols,
                                       hipMemcpyHostToDevice, ctx->stream));
    return JCH_OK;
}
int32_t d2h_matrix(jch_ctx *ctx, double *dst, const double *src, int64_t n, int64_t cols, int64_t ld)
{
    if (ld == n) JCH_HIP(ctx, hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n * cols, hipMemcpyDeviceToHost, ctx->stream));
    else JCH_HIP(ctx, hipMemcpy2DAsync(dst, sizeof(double) * ld, src, sizeof(double) * n, sizeof(double) * n, cols,
                                       hipMemcpyDeviceToHost, ctx->stream));
    return JCH_OK;
}

float ev_ms(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.f;
    if (a && b && hipEventElapsedTime(&ms, a, b) == hipSuccess) return ms;
    return 0.f;
}

// Host-resident outputs: the score columns go back WHILE the fit runs.  T[:, a] is final as soon as sweep a has finished, so a
// helper thread waits for that sweep's event and copies the column into the caller's array on a stream of its own — the
// 0.2 GB transfer of cfg2 and, more to the point, the first-touch page faults of a freshly allocated n x nlv host array
// (~18 ms: the caller's array is pageable) disappear behind the remaining sweeps.  The fit thread only records events; it
// never waits on the copies until the end (a pageable D2H issued from the fit thread itself would block it, and the GPU
// would idle between latent variables).
class t_column_copier {
public:
    t_column_copier(jch_ctx *ctx, double *host_T, const double *dev_T, int64_t n, int ncols)
        : ctx_(ctx), host_(host_T), dev_(dev_T), n_(n)
    {
        if (!host_T || ncols < 1 || getenv("JCH_HOST_T_OVERLAP_OFF")) return;
        if (hipStreamCreateWithFlags(&cs_, hipStreamNonBlocking) != hipSuccess) { cs_ = nullptr; return; }
        ev_.resize((size_t)ncols, nullptr);
        for (auto &e : ev_)
            if (hipEventCreateWithFlags(&e, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { e = nullptr; release(); return; }
        // nothing may throw across the extern "C" entry points: a failed thread start (std::system_error at the process'
        // thread limit) leaves the copier inactive and the fit takes the end-of-fit d2h path
        try {
            th_ = std::thread([this] { run(); });
            active_ = true;
        } catch (...) {
            release();
        }
    }
    ~t_column_copier() { finish(false); }
    bool active() const { return active_; }
    // column a is final once everything enqueued on the ctx stream so far has run
    void column_done(int a)
    {
        if (!active_ || a < 0 || a >= (int)ev_.size()) return;
        (void)hipEventRecord(ev_[(size_t)a], ctx_->stream);
        { std::lock_guard<std::mutex> lk(m_); ready_ = a + 1; }
        cv_.notify_one();
    }
    // wait for the copies of columns [0, ncols) (ok = false: give up, e.g. on an error path); returns false if a copy failed
    bool finish(bool ok, int ncols = 0)
    {
        if (!active_) return true;
        { std::lock_guard<std::mutex> lk(m_); stop_at_ = ok ? ncols : 0; stop_ = true; }
        cv_.notify_one();
        th_.join();
        active_ = false;
        release();
        return !failed_;
    }

private:
    void run()
    {
        (void)hipSetDevice(ctx_->device);
        int done = 0;
        for (;;) {
            int upto;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return ready_ > done || stop_; });
                upto = stop_ ? std::min(ready_, stop_at_) : ready_;
                if (stop_ && done >= upto) return;
            }
            for (; done < upto; ++done) {
                if (hipEventSynchronize(ev_[(size_t)done]) != hipSuccess ||
                    hipMemcpyAsync(host_ + (size_t)done * (size_t)n_, dev_ + (size_t)done * (size_t)n_, sizeof(double) * (size_t)n_, hipMemcpyDeviceToHost, cs_) != hipSuccess ||
                    hipStreamSynchronize(cs_) != hipSuccess)
                    failed_ = true;
            }
        }
    }
    void release()
    {
        for (auto &e : ev_) if (e) (void)hipEventDestroy(e);
        ev_.clear();
        if (cs_) (void)hipStreamDestroy(cs_);
        cs_ = nullptr;
    }
    jch_ctx *ctx_; double *host_; const double *dev_; int64_t n_;
    hipStream_t cs_ = nullptr;
    std::vector<hipEvent_t> ev_;
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    int ready_ = 0, stop_at_ = 0;
    bool stop_ = false, active_ = false, failed_ = false;
};

// raw-mode pivot check: |means - pivot| / spread above this sends the fit to the centred copy (error ~ ratio^2 * eps)
constexpr double JCH_PIVOT_MAX_RATIO = 64.0;

int32_t fit_impl(jch_ctx *ctx, const fit_io &io, int algo, bool allow_raw = true)
{
    static const char *const names[] = {"jch_plskern_fit", "jch_plsnipals_fit", "jch_plssimp_fit", "jch_plsrosa_fit", "jch_plswold_fit"};
    const char *who = names[algo];
    // plssimp and plsrosa run the plskern-shaped loop (one fused sweep per LV), plswold the plsnipals-shaped one
    const bool kern_like = algo == ALGO_KERN || algo == ALGO_SIMP || algo == ALGO_ROSA;
    if (!ctx) return JCH_EINVAL;
    JCH_TRY(validate(ctx, io, who));
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    if (io.d->dtype == JCH_BF16 && (algo != ALGO_KERN || io.d->p > JCH_SWEEP_MAXP)) {
        // bf16-stored inputs WITHOUT bf16-resident kernels (every fit but plskern; plskern with p > 2048; round 4): the contract of the
        // mode is "the Float64 algorithm on the bf16-rounded inputs" (bf16.hip), so the inputs are widened once — exactly — into
        // Float64 device buffers and the Float64 path runs on them.  No bandwidth saving there, but no refusal either.
        const jch_pls_desc &db = *io.d;
        JCH_HIP(ctx, hipSetDevice(ctx->device));
        ctx->xcopy_valid = false;
        JCH_TRY(jch_reserve(ctx, ctx->xstage, sizeof(double) * (size_t)db.n * db.p));
        JCH_TRY(jch_reserve(ctx, ctx->ystage, sizeof(double) * (size_t)db.n * db.q));
        const unsigned nbw = (unsigned)std::min<int64_t>(((int64_t)db.n * db.p + 255) / 256, (int64_t)ctx->cus * 16);
        hipLaunchKernelGGL(k_widen_bf16, dim3(nbw), dim3(256), 0, ctx->stream, (const unsigned short *)io.X, io.ldx, db.n, db.p, (double *)ctx->xstage.ptr);
        hipLaunchKernelGGL(k_widen_bf16, dim3(std::max(1u, std::min(nbw, (unsigned)(((int64_t)db.n * db.q + 255) / 256)))), dim3(256), 0, ctx->stream,
                           (const unsigned short *)io.Y, io.ldy, db.n, db.q, (double *)ctx->ystage.ptr);
        JCH_HIP(ctx, hipGetLastError());
        jch_pls_desc dw = db;
        dw.dtype = JCH_F64;
        fit_io io2 = io;
        io2.d = &dw; io2.X = ctx->xstage.ptr; io2.ldx = db.n; io2.Y = ctx->ystage.ptr; io2.ldy = db.n;
        return fit_impl(ctx, io2, algo, allow_raw);
    }
    const jch_pls_desc &d = *io.d;
    const int64_t n = d.n;
    const int p = (int)d.p, q = (int)d.q;
    // row pitch of the row-major working copy: a multiple of `ralign` doubles (pad columns are zero)
    const char *e_al = getenv("JCH_LDR_ALIGN");
    const int ralign = e_al ? std::max(2, atoi(e_al) & ~1) : 2;
    const int ldr = std::min(((p + ralign - 1) / ralign) * ralign, std::max(JCH_SWEEP_MAXP, (p + 1) & ~1)), qpad = ((q + 15) / 16) * 16;
    const bool host = d.loc == JCH_LOC_HOST;
    const bool inplace = d.inplace != 0;
    ctx->ev_used = 0;
    ctx->prof = jch_profile{};
    jch_coll_reset(ctx);
    ctx->sweep_seq = 0;   // (JCH_SWEEP_ALT: every fit starts its walk in the same direction — repeated fits stay bit-identical)
    // JCH_HOST_TIMING=1: host-side timeline of a fit on host arrays (stderr): where the wall time of the secondary metric goes
    static const bool host_timing = getenv("JCH_HOST_TIMING") != nullptr;
    auto now_ms = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double tl0 = now_ms();
    double tl_h2d = tl0, tl_enq = tl0, tl_sync = tl0;

    // JCH_REUSE_XCOPY: the caller promises that X is what the previous fit on this ctx was given (the folds of a cross-validation:
    // other weights, other Y rows held out — the same X).  If that fit left its raw row-major copy in the workspace (same pointer,
    // shape and leading dimension, f64, raw mode) this one skips the staging of X and the transposing pass and takes X'D[Yc | 1]
    // from the copy (prologue.hip k_xty_rows); otherwise the bit changes nothing.
    const xcopy_key xkey{io.X, (int64_t)n, (int64_t)io.ldx, p, host ? 1 : 0};
    const bool reuse_x = (d.reserved & JCH_REUSE_XCOPY) && allow_raw && ctx->xcopy_valid && ctx->xcopy == xkey && d.dtype == JCH_F64 &&
                         q + 1 <= 12 && ldr <= 512 && !getenv("JCH_NO_REUSE_XCOPY");
    if (!reuse_x) ctx->xcopy_valid = false;   // (whatever this fit does to the workspace, the old copy is not to be trusted afterwards)
    // ---- inputs on the device (column-major as handed over)
    double *Xc = (double *)io.X, *Yc = (double *)io.Y;
    const double *wdev = io.weights;
    int64_t ldxc = io.ldx, ldyc = io.ldy;
    if (host) {
        JCH_TRY(jch_reserve(ctx, ctx->xstage, sizeof(double) * (size_t)n * p));
        JCH_TRY(jch_reserve(ctx, ctx->ystage, sizeof(double) * (size_t)n * q));
        Xc = (double *)ctx->xstage.ptr; Yc = (double *)ctx->ystage.ptr; ldxc = n; ldyc = n;
        if (!reuse_x) JCH_TRY(h2d_matrix(ctx, Xc, (const double *)io.X, n, p, io.ldx));   // (reuse: the staging copy of the previous fit is this X)
        JCH_TRY(h2d_matrix(ctx, Yc, (const double *)io.Y, n, q, io.ldy));
        if (io.weights) {
            JCH_TRY(jch_reserve(ctx, ctx->wstage, sizeof(double) * (size_t)n));
            JCH_HIP(ctx, hipMemcpyAsync(ctx->wstage.ptr, io.weights, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
            wdev = (const double *)ctx->wstage.ptr;
        }
    }
    if (host && host_timing) { (void)hipStreamSynchronize(ctx->stream); tl_h2d = now_ms(); }
    // ---- working copies and small state
    const int nlv_cap = (int)std::min<int64_t>(d.nlv, p);  // upper bound before the global-n clamp
    if (d.dtype == JCH_F64) {
        JCH_TRY(jch_reserve(ctx, ctx->xr, sizeof(double) * (size_t)n * ldr));
        JCH_TRY(jch_reserve(ctx, ctx->yr, sizeof(double) * (size_t)n * qpad));
    }
    double *Xr = (double *)ctx->xr.ptr, *Yr = (double *)ctx->yr.ptr;
    double *dn = nullptr;
    if (!host && io.weights_norm) dn = io.weights_norm;
    else { JCH_TRY(jch_reserve(ctx, ctx->dnorm, sizeof(double) * (size_t)n)); dn = (double *)ctx->dnorm.ptr; }
    double *Tdev = nullptr;
    if (!host && io.T) Tdev = io.T;
    else { JCH_TRY(jch_reserve(ctx, ctx->tbuf, sizeof(double) * (size_t)n * nlv_cap)); Tdev = (double *)ctx->tbuf.ptr; }

    t_column_copier tcopy(ctx, (host && d.dtype == JCH_F64 && !(algo == ALGO_WOLD && (d.reserved & JCH_WOLD_REF_ZERO_WEIGHT_NAN))) ? io.T : nullptr, Tdev, n, nlv_cap);
    const size_t small_bytes = 256 * 21 + sizeof(double) * 16 * 2048 + sizeof(double) * (64 + (size_t)p + (size_t)p * qpad + 2 * (size_t)ldr + 3 * (size_t)nlv_cap * p +
                                                         (size_t)nlv_cap * q + 36 * nlv_cap + 1024 + (JCH_ZT_SLICES + 1) * ((size_t)ldr + 8 + qpad) + 2 * (size_t)(p + q) + 8 + 2 * (size_t)ldr + 128 +
                                                         (p <= JCH_SWEEP_MAXP ? jch_lv_split_doubles(p, nlv_cap) + 128 : 0));
    JCH_TRY(jch_reserve(ctx, ctx->small, small_bytes));
    carve cv{(char *)ctx->small.ptr, 0};
    jch_small s;
    s.K = cv.take((size_t)p * qpad); s.w = cv.take(ldr); s.r = cv.take(ldr);
    // everything the caller gets back lives in ONE contiguous range [P .. niter]: fetched with a single D2H copy
    s.P = cv.take((size_t)nlv_cap * p); s.R = cv.take((size_t)nlv_cap * p); s.W = cv.take((size_t)nlv_cap * p);
    s.C = cv.take((size_t)nlv_cap * q); s.TT = cv.take(nlv_cap);
    s.mom = cv.take(p + q); s.scl = cv.take(p + q);
    double *niter_dev = cv.take(nlv_cap);
    double *qual_dev = cv.take(8);   // [0]: pivot quality of the raw mode (fetched with the results)
    double *mshift_buf = cv.take((size_t)ldr + 2);   // raw mode: means - pivot (both zeroed by the weights launch)
    s.mshift = nullptr; s.rs = nullptr;
    const size_t out_bytes = (size_t)((char *)(qual_dev + 8) - (char *)s.P);
    double qual_host = 0.0;
    s.Z = cv.take((size_t)nlv_cap * 16);
    const int ldz = (ldr + 1 + qpad + 7) & ~7;
    s.zt = cv.take((size_t)JCH_ZT_SLICES * ldz); s.zpc = cv.take((size_t)ldr + qpad);
    s.hdr = cv.take(8);
    s.variant = 0;
    s.niter = algo == ALGO_WOLD ? niter_dev : nullptr;
    s.kr = nullptr; s.gpart = nullptr; s.lvctr = nullptr;
    // host copies of the small outputs: one pinned staging buffer, then plain memcpy into the caller's arrays
    auto fetch_small = [&](int k) -> int32_t {
        JCH_TRY(jch_reserve_host(ctx, out_bytes));
        char *h = (char *)ctx->hstage;
        JCH_HIP(ctx, hipMemcpyAsync(h, s.P, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        auto put = [&](double *dst, const double *dev, size_t count) {
            if (dst) memcpy(dst, h + ((const char *)dev - (const char *)s.P), sizeof(double) * count);
        };
        put(io.P, s.P, (size_t)k * p); put(io.R, s.R, (size_t)k * p);
        put(io.W, algo == ALGO_SIMP ? s.R : s.W, (size_t)k * p);   // SIMPLS has no W: R in its place (src/plssimp.jl:85-87)
        put(io.C, s.C, (size_t)k * q); put(io.TT, s.TT, k);
        put(io.xmeans, s.mom, p); put(io.ymeans, s.mom + p, q); put(io.xscales, s.scl, p); put(io.yscales, s.scl + p, q);
        if (algo == ALGO_WOLD) put(io.niter, niter_dev, k);
        memcpy(&qual_host, h + ((const char *)qual_dev - (const char *)s.P), sizeof(double));
        return JCH_OK;
    };
    s.dbg = getenv("JCH_LV_DEBUG") ? cv.take(512 + 16 * (nlv_cap + 2)) : nullptr;
    // plsnipals / plswold with a narrow Y: the rewrite of X after every LV is postponed and done every `defer_m`-th LV
    // (k_sweep_lazy / k_kpass_lazy; JCH_NIPALS_DEFER=1 restores the eager deflation)
    int defer_m = 1;
    double *pend_p = nullptr;
    if (!kern_like && d.dtype == JCH_F64) {
        const char *e_m = getenv("JCH_NIPALS_DEFER");
        defer_m = std::max(1, std::min(e_m ? atoi(e_m) : JCH_NIPALS_DEFER_DEFAULT, jch_nipals_lazy_capacity(ldr, q)));
        if (defer_m > 1) pend_p = cv.take((size_t)defer_m * jch_nipals_lazy_pitch(ldr));
    }

    // OPT-IN one-pass NIPALS (JCH_NIPALS_ONE_PASS): the next X'DY by a rank-one update in the small-state kernel, rows written back
    // every defer_m-th LV by a pass that computes nothing else
    const bool onepass = (d.reserved & JCH_NIPALS_ONE_PASS) != 0 && (algo == ALGO_NIPALS || algo == ALGO_WOLD);
    double *zero16 = nullptr;
    if ((d.reserved & JCH_NIPALS_ONE_PASS) && !(onepass && pend_p && !inplace))
        return jch_fail(ctx, JCH_EINVAL, "%s: JCH_NIPALS_ONE_PASS needs plsnipals / plswold with q <= 16, p <= 2048, inplace = 0, Float64", who);
    if (onepass) { zero16 = cv.take(16); JCH_HIP(ctx, hipMemsetAsync(zero16, 0, sizeof(double) * 16, ctx->stream)); s.variant = 3; }
    if (s.dbg) JCH_HIP(ctx, hipMemsetAsync(s.dbg, 0, sizeof(double) * (512 + 16 * (nlv_cap + 2)), ctx->stream));
    if (d.dtype == JCH_BF16) {   // bf16 storage mode (plskern only): its own prologue + sweep, same small-state kernels
        const bool fastb = q <= 16 && p <= JCH_SWEEP_MAXP && jch_lv_fast_lds_bytes(p, q, qpad, ldr, nlv_cap) <= 150 * 1024 && !getenv("JCH_SMALLSTATE_GENERIC");
        {   // split small-state path (see the f64 loop below); the fused inbox exchange keeps the one-kernel path
            const char *e_sp = getenv("JCH_LV_SPLIT");
            const bool fuse_b = ctx->p2p.ready && !ctx->loop && !getenv("JCH_P2P_UNFUSED");
            if (fastb && (!fuse_b || jch_lv_split_p2p_ok(ctx, p)) && !(e_sp && atoi(e_sp) == 0) && jch_lv_solve_lds_bytes(p, q, ldr, nlv_cap) <= 150 * 1024) {
                s.kr = cv.take(16); JCH_TRY(jch_lv_split_begin_fit(ctx, s, cv.take(jch_lv_split_doubles(p, nlv_cap)), p, nlv_cap));
            }
        }
        hipEvent_t evb = jch_ev(ctx);
        int nlvb = 0;
        JCH_TRY(jch_fit_plskern_bf16(ctx, d, io.X, io.ldx, io.Y, io.ldy, wdev, dn, Tdev, s, ldr, qpad, ldz, fastb, &nlvb));
        hipEvent_t eve = jch_ev(ctx);
        const size_t ev_last = ctx->ev_used - 1;
        JCH_TRY(fetch_small(nlvb));
        JCH_TRY(jch_p2p_check(ctx));
        if (io.nlv_out) *io.nlv_out = nlvb;
        if (ctx->profiling) {
            jch_profile &pr = ctx->prof;
            pr.fit_ms = ev_ms(evb, eve);
            pr.prologue_ms = ctx->ev_mark > 0 ? ev_ms(evb, ctx->ev_pool[ctx->ev_mark - 1]) : 0.0;
            double sw = 0.0; int cnt = 0;
            for (size_t i = ctx->ev_mark; i + 1 < ev_last; i += 2) { sw += ev_ms(ctx->ev_pool[i], ctx->ev_pool[i + 1]); ++cnt; }
            if (ctx->prof_stride > 1 && cnt > 0 && cnt < nlvb) { sw = sw * (double)nlvb / (double)cnt; cnt = nlvb; }   // sampled sweeps, see fit_impl's own block
            pr.sweep_ms = sw; pr.sweep_launches = cnt; pr.nlv = nlvb;
            pr.smallstate_ms = pr.fit_ms - pr.prologue_ms - sw;
            pr.sweep_bytes = (double)n * ((p + 7) & ~7) * 2.0 + 16.0 * (double)n;
            jch_coll_collect(ctx, pr);
        }
        return JCH_OK;
    }
    hipEvent_t ev_begin = jch_ev(ctx);
    // ---- K0 weights; global row count for the nlv clamp (src/plskern.jl:116-117)
    JCH_TRY(jch_launch_weights(ctx, wdev, n, dn, s.hdr, qual_dev, 8, mshift_buf, ldr + 2));
    int64_t n_total = n;
    // only a shard smaller than min(p, nlv) has to learn the global row count (one host sync); every rank reaches the
    // same clamp either way, because n_total >= the largest shard
    if (ctx->nranks > 1 && n < std::min<int64_t>(p, d.nlv)) {
        double hdr_h[2];
        JCH_HIP(ctx, hipMemcpyAsync(hdr_h, s.hdr, sizeof hdr_h, hipMemcpyDeviceToHost, ctx->stream));
        JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        n_total = (int64_t)(hdr_h[1] + 0.5);
    }
    const int nlv = (int)std::min<int64_t>(std::min<int64_t>(n_total, p), d.nlv);
    // small-state fast path (smallstate_fast.hip): everything in LDS, q <= 16
    const bool fast = q <= 16 && nlv <= 1024 && p <= JCH_SWEEP_MAXP && jch_lv_fast_lds_bytes(p, q, qpad, ldr, nlv) <= 150 * 1024 && !getenv("JCH_SMALLSTATE_GENERIC");
    // the fast small-state kernel sums the second-stage slices itself; with several GPUs the [slices][ldz] block is
    // all-reduced as one message (still latency-bound at 32 KB) instead of being collapsed by an extra launch
    // plssimp / plswold: their fast kernels (siblings.hip) need the p x q state in LDS; outside that envelope the generic
    // small-state kernel (K in global memory, q <= 64, any p) takes over
    const bool sib = algo == ALGO_SIMP || algo == ALGO_WOLD;
    const bool sib_fast = sib && jch_sibling_supported(p, q, ldr, nlv) && !getenv("JCH_SMALLSTATE_GENERIC");
    const bool all_fast = sib ? (sib_fast && (algo == ALGO_SIMP || fast)) : fast;   // every small-state kernel of this fit is a fast one
    const int max_slices = all_fast ? JCH_ZT_SLICES : 1;
    const bool fuse_inbox = all_fast && ctx->p2p.ready && !ctx->loop && !getenv("JCH_P2P_UNFUSED") &&
                            (size_t)(ldr + 1 + qpad) <= ctx->p2p.cap;   // the fused kernel writes one whole message into one inbox slot
    int nslice = 1;
    // ---- K1 means (+ two-pass std), K2 centre/scale + row-major copy + XtY
    const bool ext_scales = io.xscales_in != nullptr;
    // RAW MODE (plskern / plsrosa, no scaling, not in place): the separate means pass over X is dropped.  K2 copies X
    // UNCENTRED into the row-major working copy and computes X'D[Yc | 1] — the kernel matrix against the centred Y plus,
    // in a spare pad column, the weighted column sums, i.e. the means (one read of X instead of two, one all-reduce
    // instead of two).  The sweeps then use t_i = x_i.r - mu.r and zp = zp_raw - mu * sum_i d_i t_i (sweep.hip,
    // smallstate_fast.hip); T, P, C, TT, xmeans are the same quantities as in the centred formulation.
    const bool raw_mode = (((algo == ALGO_KERN || algo == ALGO_ROSA) && fast) || (algo == ALGO_SIMP && all_fast)) && !ext_scales && !inplace &&
                          q <= 15 && (d.reserved & ~JCH_REUSE_XCOPY) == 0 && p <= JCH_SWEEP_MAXP && allow_raw && !getenv("JCH_CENTRED_COPY") &&
                          !(d.scal && getenv("JCH_CENTRED_COPY_SCAL"));
    if (raw_mode) {
        // (mshift and the pivot-quality word were zeroed by the weights launch above; the divisor slots `s.mom` handed to K2
        // as `scl` are never read with SCAL = false; k_extract_means moves the Y means and resets s.scl to ones)
        s.mshift = mshift_buf;
        double *spread2 = cv.take(p);
        JCH_TRY(jch_launch_pivot(ctx, Xc, ldxc, n, p, s.hdr, s.scl, spread2));                      // scl[0..p): the pivot K2 subtracts
        JCH_TRY(jch_launch_moments(ctx, Yc, ldyc, nullptr, 0, dn, n, q, 0, nullptr, s.scl + p));  // Y means -> scl[p..p+q)
        bool from_copy = false;
        if (reuse_x) JCH_TRY(jch_launch_xty_rows(ctx, Xr, ldr, Yc, ldyc, dn, n, p, q, /*mom =*/s.scl, Yr, qpad, s.K, /*means_out =*/s.mom,
                                                 /*mshift_out =*/s.mshift, spread2, qual_dev, /*ones_out =*/s.scl, &from_copy));
        if (from_copy) ctx->xcopy_reused++;
        else {
        JCH_TRY(jch_launch_center_xty(ctx, Xc, ldxc, Yc, ldyc, dn, n, p, q, /*mom =*/s.scl, /*scl =*/s.mom, false, Xr, ldr, Yr, qpad, s.K, false,
                                      /*means_out =*/s.mom, /*mshift_out =*/s.mshift, spread2, qual_dev, /*ones_out =*/s.scl));
        ctx->xcopy = xkey; ctx->xcopy_valid = true;   // the raw copy x - pivot of THIS X now sits in the workspace
        }
        if (d.scal) {   // stds from ONE streaming pass over the row-major copy; the scaling itself is folded into r / s and zp / s
            s.rs = cv.take((size_t)ldr + 2);
            JCH_HIP(ctx, hipMemsetAsync(s.rs, 0, sizeof(double) * ((size_t)ldr + 2), ctx->stream));
            JCH_TRY(jch_launch_raw_scales(ctx, Xr, n, p, ldr, dn, s.mshift, Yr, qpad, q, s.zt, s.scl, s.K));
        }
        s.variant = 2;
    } else {
    ctx->xcopy_valid = false;   // (the working copy becomes the centred / scaled / deflated one)
    JCH_TRY(jch_launch_moments(ctx, Xc, ldxc, Yc, ldyc, dn, n, p, q, nullptr, s.mom));
    if (ext_scales) {   // divisors handed in by the caller (multiblock scaling): no second-moment pass
        JCH_TRY(jch_reserve_host(ctx, sizeof(double) * (size_t)(p + q)));
        double *h = (double *)ctx->hstage;
        for (int j = 0; j < p; ++j) h[j] = io.xscales_in[j];
        for (int k = 0; k < q; ++k) h[p + k] = io.yscales_in ? io.yscales_in[k] : 1.0;
        JCH_HIP(ctx, hipMemcpyAsync(s.scl, h, sizeof(double) * (size_t)(p + q), hipMemcpyHostToDevice, ctx->stream));
        JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (the staging buffer is reused for the outputs)
    } else if (d.scal) JCH_TRY(jch_launch_moments(ctx, Xc, ldxc, Yc, ldyc, dn, n, p, q, s.mom, s.scl));
    else hipLaunchKernelGGL(k_fill_const, dim3((p + q + 255) / 256), dim3(256), 0, ctx->stream, s.scl, p + q, 1.0);
    JCH_TRY(jch_launch_center_xty(ctx, Xc, ldxc, Yc, ldyc, dn, n, p, q, s.mom, s.scl, inplace && kern_like, Xr, ldr, Yr, qpad, s.K, d.scal != 0 || ext_scales));
    }
    hipEvent_t ev_prologue = jch_ev(ctx);
    ctx->coll_phase = 1;   // all-reduces from here on belong to the LV loop (profile: collective_ms)

    // ---- LV loop
    const size_t sweep_ev0 = ctx->ev_used;
    const bool variant2 = (d.reserved & 1) != 0;
    bool split = false;
    const bool wold_ref_nan = algo == ALGO_WOLD && (d.reserved & JCH_WOLD_REF_ZERO_WEIGHT_NAN) != 0;
    int x_reads = 0, x_writes = 0;   // plsnipals-shaped loops: whole passes over the working copy (profile: bytes actually moved)
    if (variant2) {   // OPT-IN kernel algorithm #2 (kern2.hip): Gram once, LV loop without X and without collectives
        if (algo != ALGO_KERN || !fast) return jch_fail(ctx, JCH_EINVAL, "%s: variant 2 needs plskern with q <= 16 and p <= %d", who, JCH_SWEEP_MAXP);
        JCH_TRY(jch_reserve(ctx, ctx->gram, sizeof(double) * (size_t)p * ldr));
        double *G = (double *)ctx->gram.ptr;
        JCH_TRY(jch_launch_syrk(ctx, Xr, n, p, ldr, dn, G, ldr));
        s.variant = 1;
        JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr, -1, nlv, 0, 1, ldz, true));
        for (int a = 0; a < nlv; ++a) {
            JCH_TRY(jch_launch_gmatvec(ctx, G, ldr, p, ldr, s.r, s.zt));
            JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr, a, nlv, 0, 1, ldz, true));
        }
        JCH_TRY(jch_launch_scores(ctx, Xr, n, p, ldr, s.R, nlv, Tdev));
    } else {
    if (algo == ALGO_SIMP && all_fast) JCH_TRY(jch_launch_lv_update_simp(ctx, s, p, q, ldr, -1, nlv, 1, ldz));
    else if (algo == ALGO_SIMP) JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr, -1, nlv, 2, 1, ldz, false));
    else if (algo == ALGO_WOLD && all_fast) JCH_TRY(jch_launch_wold_b(ctx, s, p, q, ldr, 0, nlv, io.tol, io.maxit));
    else if (algo == ALGO_WOLD) JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr, 0 | 0x20000000, nlv, 4, 1, ldz, false, false, nullptr, 0, 0, io.tol, io.maxit));
    else {
        // SPLIT small-state path (smallstate_split.hip; round 4): per LV a p-parallel kernel on (p + 15) / 16 CUs (slice sums, c,
        // K update, P / W / R columns, partial Gram / Z sums) + a single-workgroup kernel that starts at the eigenvector.  plskern /
        // plsrosa on the fast path; not with the inbox all-reduce fused into the one-kernel path.  JCH_LV_SPLIT=0: the one-kernel path.
        const char *e_sp = getenv("JCH_LV_SPLIT");
        split = (algo == ALGO_KERN || algo == ALGO_ROSA) && fast && !(e_sp && atoi(e_sp) == 0) &&
                jch_lv_solve_lds_bytes(p, q, ldr, nlv) <= 150 * 1024 && (!fuse_inbox || jch_lv_split_p2p_ok(ctx, p));
        if (split) { s.kr = cv.take(16); JCH_TRY(jch_lv_split_begin_fit(ctx, s, cv.take(jch_lv_split_doubles(p, nlv)), p, nlv)); }
        JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr, -1, nlv, kern_like ? 0 : 1, 1, ldz, fast));
    }
    int npend = 0, pend_a0 = 0;   // postponed deflations: LVs pend_a0 .. pend_a0 + npend - 1
    if (pend_p) JCH_HIP(ctx, hipMemsetAsync(pend_p, 0, sizeof(double) * (size_t)defer_m * jch_nipals_lazy_pitch(ldr), ctx->stream));
    for (int a = 0; a < nlv; ++a) {
        double *tcol = Tdev + (size_t)a * (size_t)n;
        if (kern_like) {
            jch_part_view pv;
            const bool split_fused = split && fuse_inbox;   // the exchange happens inside k_lv_spread, on the rank's OWN partial rows
            JCH_TRY(jch_launch_sweep(ctx, Xr, n, p, ldr, dn, s.rs ? s.rs : s.r, Yr, qpad, 0, tcol, s.zt, ldz, max_slices, &nslice, raw_mode ? s.mshift : nullptr,
                                     split && (ctx->nranks == 1 || split_fused) ? &pv : nullptr));
            tcopy.column_done(a);
            if (ctx->nranks > 1 && max_slices > 1) nslice = JCH_ZT_SLICES;   // rank-independent message size (a small shard may use 1 slice; the rest hold zeros)
            if (split) {
                // one GPU: k_lv_spread sums the sweep's block partials itself; several: the slices are all-reduced first
                if (!pv.part) {   // (a sweep kernel that reduced into zt: the slices are the partial rows)
                    if (!split_fused) JCH_TRY(jch_allreduce_slices(ctx, s.zt, ldr + 1 + (raw_mode ? 1 : 0), nslice, ldz, &nslice));
                    pv.part = s.zt; pv.nb = nslice; pv.ldpart = ldz;
                }
                if (split_fused) ctx->coll_transport = JCH_TRANSPORT_INBOX_FUSED;
                JCH_TRY(jch_launch_lv_split(ctx, s, p, q, ldr, a, nlv, pv.part, pv.nb, pv.ldpart, ldr, raw_mode ? ldr + 1 : -1, raw_mode ? 1 : 0, a + 1 < nlv, split_fused));
                continue;
            }
            // ONE collective per LV: [zp (p), tt].  With the inbox transport and the fast small-state kernel it happens
            // INSIDE that kernel (no launch of its own); otherwise here (RCCL / inbox kernel / loopback).
            if (fuse_inbox && algo != ALGO_SIMP) {
                ctx->coll_transport = JCH_TRANSPORT_INBOX_FUSED;
                JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr, a, nlv, 0, nslice, ldz, fast, true));
            } else {
                JCH_TRY(jch_allreduce_slices(ctx, s.zt, ldr + 1 + (raw_mode ? 1 : 0), nslice, ldz, &nslice));
                if (algo == ALGO_SIMP && all_fast) JCH_TRY(jch_launch_lv_update_simp(ctx, s, p, q, ldr, a, nlv, nslice, ldz));
                else if (algo == ALGO_SIMP) JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr, a, nlv, 2, nslice, ldz, false));
                else JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr, a, nlv, 0, nslice, ldz, fast));
            }
        } else {
            if (pend_p) JCH_TRY(jch_launch_sweep_lazy(ctx, Xr, n, ldr, dn, s.w, Yr, qpad, tcol, s.zt, ldz, max_slices, &nslice, pend_p, npend,
                                                      defer_m - 1, Tdev + (size_t)pend_a0 * (size_t)n, n));
            else if (qpad > 64) {   // more responses than the NIPALS sweep has lanes for: t, tt, zp from the plskern-shaped sweep, c_raw = Y'Dt on its own
                JCH_TRY(jch_launch_sweep(ctx, Xr, n, p, ldr, dn, s.w, Yr, qpad, 0, tcol, s.zt, ldz, max_slices, &nslice));
                JCH_TRY(jch_launch_ytdt(ctx, Yr, n, qpad, dn, tcol, s.zt + ldr + 1));
            } else JCH_TRY(jch_launch_sweep(ctx, Xr, n, p, ldr, dn, s.w, Yr, qpad, q, tcol, s.zt, ldz, max_slices, &nslice));
            tcopy.column_done(a);
            if (ctx->nranks > 1 && max_slices > 1) nslice = JCH_ZT_SLICES;
            ++x_reads;
            if (fuse_inbox) {   // [zp_raw, tt, c_raw] reduced inside the phase-A kernel
                ctx->coll_transport = JCH_TRANSPORT_INBOX_FUSED;
                JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr, a | 0x40000000, nlv, 1, nslice, ldz, fast, true));
            } else {
                JCH_TRY(jch_allreduce_slices(ctx, s.zt, ldr + 1 + qpad, nslice, ldz, &nslice));
                JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr, a | 0x40000000, nlv, 1, nslice, ldz, fast));
            }
            const bool last = a + 1 == nlv;
            if (!last || inplace) {
                // X -= t zp', Y -= t c' fused with the next K = X'DY (src/plsnipals.jl:86-87,71; src/plswold.jl:98-99)
                if (pend_p && onepass) {   // K_{a+1} is already in place (phase A); only the write-back of the rows, every defer_m-th LV
                    if (npend == 0) pend_a0 = a;
                    JCH_HIP(ctx, hipMemcpyAsync(pend_p + (size_t)npend * jch_nipals_lazy_pitch(ldr), s.zpc, sizeof(double) * (size_t)ldr,
                                                hipMemcpyDeviceToDevice, ctx->stream));
                    ++npend;
                    if (npend == defer_m && !last) {
                        JCH_TRY(jch_launch_kpass_lazy(ctx, Xr, n, p, ldr, Yr, qpad, q, dn, pend_p, npend, defer_m,
                                                      Tdev + (size_t)pend_a0 * (size_t)n, n, zero16, true, nullptr));
                        npend = 0; ++x_writes; ++x_reads;
                    }
                    --x_reads;   // (balanced by the unconditional ++x_reads below: this LV made no second pass)
                } else
                if (pend_p) {
                    if (npend == 0) pend_a0 = a;
                    JCH_HIP(ctx, hipMemcpyAsync(pend_p + (size_t)npend * jch_nipals_lazy_pitch(ldr), s.zpc, sizeof(double) * (size_t)ldr,
                                                hipMemcpyDeviceToDevice, ctx->stream));
                    ++npend;
                    const bool flush = npend == defer_m || last;
                    JCH_TRY(jch_launch_kpass_lazy(ctx, Xr, n, p, ldr, Yr, qpad, q, dn, pend_p, npend, defer_m,
                                                  Tdev + (size_t)pend_a0 * (size_t)n, n, s.zpc + ldr, flush, last ? nullptr : s.K));
                    if (flush) { npend = 0; ++x_writes; }
                } else {
                    JCH_TRY(jch_launch_deflate(ctx, Xr, n, p, ldr, Yr, qpad, q, dn, tcol, s.zpc, last ? nullptr : s.K));
                    ++x_writes;
                }
                ++x_reads;
            }
            if (!last) {
                if (algo == ALGO_WOLD && all_fast) JCH_TRY(jch_launch_wold_b(ctx, s, p, q, ldr, a + 1, nlv, io.tol, io.maxit));
                else if (algo == ALGO_WOLD) JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr, (a + 1) | 0x20000000, nlv, 4, 1, ldz, false, false, nullptr, 0, 0, io.tol, io.maxit));
                else JCH_TRY(jch_launch_lv_update(ctx, s, p, q, qpad, ldr, (a + 1) | 0x20000000, nlv, 1, 1, ldz, fast));
            }
        }
    }
    }
    if (wold_ref_nan) {   // (the overlapped column copies are off for this flag: the scores are edited after the loop)
        hipLaunchKernelGGL(k_nan_zero_weight_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, Tdev, n, nlv, dn);
        JCH_HIP(ctx, hipGetLastError());
    }
    if (algo == ALGO_ROSA) JCH_TRY(jch_launch_rosa_orthw(ctx, s.W, p, nlv));   // src/plsrosa.jl:77-79
    if (algo == ALGO_NIPALS || algo == ALGO_WOLD || algo == ALGO_ROSA) JCH_TRY(jch_launch_nipals_R(ctx, s, p, nlv));   // R = W inv(P'W)
    hipEvent_t ev_end = jch_ev(ctx);
    const size_t sweep_ev1 = ctx->ev_used;

    // ---- results
    if (!kern_like && inplace) {  // hand back centred + deflated X, Y in the caller's column-major arrays
        // (plswold! additionally leaves the row metric sqrt(w) on them, src/plswold.jl:57-58)
        JCH_TRY(jch_launch_export_colmajor(ctx, Xr, ldr, Yr, qpad, n, p, q, Xc, ldxc, Yc, ldyc, algo == ALGO_WOLD ? dn : nullptr));
    }
    if (algo == ALGO_ROSA && inplace) JCH_TRY(jch_launch_ydeflate_all(ctx, Yc, ldyc, Tdev, n, s.C, q, nlv));   // src/plsrosa.jl:87
    auto d2h = [&](double *dst, const double *src, size_t count) -> int32_t {
        if (dst) JCH_HIP(ctx, hipMemcpyAsync(dst, src, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
        return JCH_OK;
    };
    if (host) {
        // (the score columns have been travelling since their sweeps finished; algorithm #2 writes T in one GEMM at the end)
        if (!tcopy.active() || variant2) JCH_TRY(d2h(io.T, Tdev, (size_t)n * nlv));
        JCH_TRY(d2h(io.weights_norm, dn, (size_t)n));
        if (inplace) {
            JCH_TRY(d2h_matrix(ctx, (double *)io.X, Xc, n, p, io.ldx));
            JCH_TRY(d2h_matrix(ctx, (double *)io.Y, Yc, n, q, io.ldy));
        }
    }
    tl_enq = now_ms();
    JCH_TRY(fetch_small(nlv));   // (ends with the stream sync of the whole fit)
    tl_sync = now_ms();
    if (tcopy.active() && !tcopy.finish(!variant2, nlv)) return jch_fail(ctx, JCH_EHIP, "%s: copying the scores to the host failed", who);
    JCH_TRY(jch_p2p_check(ctx));
    if (raw_mode && !(qual_host <= JCH_PIVOT_MAX_RATIO)) {
        // the sampled pivot was far from the means (sorted / trending / blank leading rows): the raw formulation would
        // lose ~ratio^2 * eps.  Inputs are untouched in raw mode, so simply fit again on the centred copy (every rank
        // takes the same decision: qual comes from all-reduced, bit-identical state).
        ctx->pivot_refits++;
        return fit_impl(ctx, io, algo, false);
    }
    if (io.nlv_out) *io.nlv_out = nlv;
    if (host && host_timing)
        fprintf(stderr, "[jch] host-arrays fit: H2D %.1f ms | enqueue %.1f ms | device drain + small outputs %.1f ms | score columns still in flight %.1f ms | total %.1f ms\n",
                tl_h2d - tl0, tl_enq - tl_h2d, tl_sync - tl_enq, now_ms() - tl_sync, now_ms() - tl0);
    if (s.dbg) {
        std::vector<double> h(nlv + 1);
        (void)hipMemcpy(h.data(), s.dbg, sizeof(double) * (nlv + 1), hipMemcpyDeviceToHost);
        fprintf(stderr, "[jch] jacobi sweeps per LV:");
        for (int i = 0; i < nlv; ++i) fprintf(stderr, " %d", (int)h[i]);
        fprintf(stderr, "\n");
        std::vector<double> st(16 * (nlv + 1));
        (void)hipMemcpy(st.data(), s.dbg + 512, sizeof(double) * st.size(), hipMemcpyDeviceToHost);
        for (int c : {0, 1, nlv / 2, nlv - 1}) {
            fprintf(stderr, "[jch] lv_update call %d stamps (cycles since kernel start):", c);
            for (int k = 1; k < 16; ++k) if (st[16 * c + k] > 0) fprintf(stderr, " %d:%.0f", k, st[16 * c + k] - st[16 * c]);   // (split path: 5-7 = k_lv_spread's block 0, before the solve kernel's origin)
            fprintf(stderr, "\n");
        }
    }

    if (ctx->profiling) {
        jch_profile &pr = ctx->prof;
        pr.fit_ms = ev_ms(ev_begin, ev_end);
        pr.prologue_ms = ev_ms(ev_begin, ev_prologue);
        pr.nlv = nlv;
        // events between sweep_ev0 and sweep_ev1 come in (begin,end) pairs recorded by the sweep/deflate launchers
        double sw = 0.0; int cnt = 0;
        for (size_t i = sweep_ev0; i + 1 < sweep_ev1; i += 2) { sw += ev_ms(ctx->ev_pool[i], ctx->ev_pool[i + 1]); ++cnt; }
        // sampled sweeps (jch_ctx_set_profiling(ctx, N > 1)): cnt of the nlv launches were bracketed — their mean stands for the rest
        const bool sampled = kern_like && !variant2 && ctx->prof_stride > 1;
        if (sampled) sw = cnt > 0 ? sw * (double)nlv / (double)cnt : 0.0;
        pr.sweep_ms = sw;
        pr.sweep_launches = variant2 ? 1 : (kern_like ? (sampled ? (cnt > 0 ? nlv : 0) : cnt) : nlv);
        pr.smallstate_ms = pr.fit_ms - pr.prologue_ms - sw;
        const double per_x = (double)n * ldr * 8.0;
        // plsnipals-shaped loops: average per LV of the passes actually made (eager: 2 reads + 1 write; postponed
        // write-back: 2 reads + one write every m-th LV)
        pr.sweep_bytes = kern_like ? per_x + 16.0 * (double)n : per_x * (double)(x_reads + x_writes) / (double)std::max(nlv, 1);
        jch_coll_collect(ctx, pr);
    }
    return JCH_OK;
}

}  // namespace

extern "C" int32_t jch_plskern_fit(jch_ctx *ctx, const jch_pls_desc *desc, void *X, int64_t ldx, void *Y, int64_t ldy,
                                   const double *weights, double *T, double *P, double *R, double *W, double *C, double *TT,
                                   double *xmeans, double *xscales, double *ymeans, double *yscales, double *weights_norm,
                                   int32_t *nlv_out)
{
    fit_io io{desc, X, ldx, Y, ldy, weights, T, P, R, W, C, TT, xmeans, xscales, ymeans, yscales, weights_norm, nlv_out};
    return fit_impl(ctx, io, 0);
}

extern "C" int32_t jch_plsnipals_fit(jch_ctx *ctx, const jch_pls_desc *desc, void *X, int64_t ldx, void *Y, int64_t ldy,
                                     const double *weights, double *T, double *P, double *R, double *W, double *C, double *TT,
                                     double *xmeans, double *xscales, double *ymeans, double *yscales, double *weights_norm,
                                     int32_t *nlv_out)
{
    fit_io io{desc, X, ldx, Y, ldy, weights, T, P, R, W, C, TT, xmeans, xscales, ymeans, yscales, weights_norm, nlv_out};
    return fit_impl(ctx, io, 1);
}

extern "C" int32_t jch_plssimp_fit(jch_ctx *ctx, const jch_pls_desc *desc, void *X, int64_t ldx, void *Y, int64_t ldy,
                                   const double *weights, double *T, double *P, double *R, double *W, double *C, double *TT,
                                   double *xmeans, double *xscales, double *ymeans, double *yscales, double *weights_norm,
                                   int32_t *nlv_out)
{
    fit_io io{desc, X, ldx, Y, ldy, weights, T, P, R, W, C, TT, xmeans, xscales, ymeans, yscales, weights_norm, nlv_out};
    return fit_impl(ctx, io, ALGO_SIMP);
}

extern "C" int32_t jch_plsrosa_fit(jch_ctx *ctx, const jch_pls_desc *desc, void *X, int64_t ldx, void *Y, int64_t ldy,
                                   const double *weights, double *T, double *P, double *R, double *W, double *C, double *TT,
                                   double *xmeans, double *xscales, double *ymeans, double *yscales, double *weights_norm,
                                   int32_t *nlv_out)
{
    fit_io io{desc, X, ldx, Y, ldy, weights, T, P, R, W, C, TT, xmeans, xscales, ymeans, yscales, weights_norm, nlv_out};
    return fit_impl(ctx, io, ALGO_ROSA);
}

extern "C" int32_t jch_plswold_fit(jch_ctx *ctx, const jch_pls_desc *desc, void *X, int64_t ldx, void *Y, int64_t ldy,
                                   const double *weights, double tol, int32_t maxit, double *T, double *P, double *R, double *W,
                                   double *C, double *TT, double *xmeans, double *xscales, double *ymeans, double *yscales,
                                   double *weights_norm, double *niter, int32_t *nlv_out)
{
    fit_io io{desc, X, ldx, Y, ldy, weights, T, P, R, W, C, TT, xmeans, xscales, ymeans, yscales, weights_norm, nlv_out};
    io.tol = tol; io.maxit = maxit; io.niter = niter;
    return fit_impl(ctx, io, ALGO_WOLD);
}

extern "C" int32_t jch_plskern_fit_scaled(jch_ctx *ctx, const jch_pls_desc *desc, void *X, int64_t ldx, void *Y, int64_t ldy,
                                          const double *weights, const double *xscales_in, const double *yscales_in, double *T,
                                          double *P, double *R, double *W, double *C, double *TT, double *xmeans, double *xscales,
                                          double *ymeans, double *yscales, double *weights_norm, int32_t *nlv_out)
{
    fit_io io{desc, X, ldx, Y, ldy, weights, T, P, R, W, C, TT, xmeans, xscales, ymeans, yscales, weights_norm, nlv_out};
    if (ctx && !xscales_in) return jch_fail(ctx, JCH_EINVAL, "jch_plskern_fit_scaled: xscales_in is NULL");
    if (ctx && desc && desc->dtype != JCH_F64) return jch_fail(ctx, JCH_EINVAL, "jch_plskern_fit_scaled: Float64 only");
    io.xscales_in = xscales_in; io.yscales_in = yscales_in;
    return fit_impl(ctx, io, ALGO_KERN);
}

// Weighted column means and (optionally) uncorrected standard deviations: `colmean` / `colstd` (src/utility.jl:193-195,
// 312-323) as a call of their own (K0 + K1 of the fit).
extern "C" int32_t jch_col_stats(jch_ctx *ctx, int32_t loc, const double *X, int64_t n, int64_t p, int64_t ldx, const double *weights,
                                 double *means, double *stds)
{
    if (!ctx) return JCH_EINVAL;
    if (!X || !means || n < 1 || p < 1 || ldx < n) return jch_fail(ctx, JCH_EINVAL, "jch_col_stats: bad arguments");
    if (loc != JCH_LOC_HOST && loc != JCH_LOC_DEVICE) return jch_fail(ctx, JCH_EINVAL, "jch_col_stats: bad loc");
    if (p > (1 << 20)) return jch_fail(ctx, JCH_EINVAL, "jch_col_stats: p too large");
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    const double *dX = X, *dw = weights;
    int64_t ldxd = ldx;
    if (loc == JCH_LOC_HOST) {
        if (ctx->xcopy.host) ctx->xcopy_valid = false;   // (the staging buffer a host-array fit would re-use is overwritten)
        JCH_TRY(jch_reserve(ctx, ctx->xstage, sizeof(double) * (size_t)n * p));
        JCH_TRY(h2d_matrix(ctx, (double *)ctx->xstage.ptr, X, n, p, ldx));
        dX = (const double *)ctx->xstage.ptr; ldxd = n;
        if (weights) {
            JCH_TRY(jch_reserve(ctx, ctx->wstage, sizeof(double) * (size_t)n));
            JCH_HIP(ctx, hipMemcpyAsync(ctx->wstage.ptr, weights, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
            dw = (const double *)ctx->wstage.ptr;
        }
    }
    JCH_TRY(jch_reserve(ctx, ctx->dnorm, sizeof(double) * (size_t)n));
    JCH_TRY(jch_reserve(ctx, ctx->small, sizeof(double) * (2 * (size_t)p + 64)));
    double *mom = (double *)ctx->small.ptr, *scl = mom + p, *hdr = scl + p;
    double *dn = (double *)ctx->dnorm.ptr;
    JCH_TRY(jch_launch_weights(ctx, dw, n, dn, hdr));
    JCH_TRY(jch_launch_moments(ctx, dX, ldxd, nullptr, 0, dn, n, (int)p, 0, nullptr, mom));
    if (stds) JCH_TRY(jch_launch_moments(ctx, dX, ldxd, nullptr, 0, dn, n, (int)p, 0, mom, scl));
    JCH_HIP(ctx, hipMemcpyAsync(means, mom, sizeof(double) * (size_t)p, hipMemcpyDeviceToHost, ctx->stream));
    if (stds) JCH_HIP(ctx, hipMemcpyAsync(stds, scl, sizeof(double) * (size_t)p, hipMemcpyDeviceToHost, ctx->stream));
    JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JCH_OK;
}
