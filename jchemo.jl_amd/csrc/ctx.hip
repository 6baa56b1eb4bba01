// Context, error plumbing, workspace, RCCL binding, profiling events.  Public ABI: include/jchemo_hip.h.
#include <dlfcn.h>

#include <algorithm>
#include <pthread.h>
#include <stdarg.h>
#include <stdlib.h>

#include "jch_internal.h"

static thread_local std::string g_create_err;

int32_t jch_fail(jch_ctx *ctx, int32_t code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_create_err = buf;
    return code;
}

extern "C" int32_t jch_version(void) { return JCH_VERSION; }

extern "C" const char *jch_last_error(const jch_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" int32_t jch_ctx_create(jch_ctx **out, int32_t device_id, void *stream, uint32_t flags)
{
    (void)flags;
    if (!out) return jch_fail(nullptr, JCH_EINVAL, "jch_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return jch_fail(nullptr, JCH_ENODEV, "jch_ctx_create: no HIP device (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev)
        return jch_fail(nullptr, JCH_EINVAL, "jch_ctx_create: device_id %d out of range [0,%d)", device_id, ndev);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return jch_fail(nullptr, JCH_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return jch_fail(nullptr, JCH_ENODEV, "jch_ctx_create: device %d is %s; this library is built for gfx950 only",
                        device_id, prop.gcnArchName);
    jch_ctx *ctx = new (std::nothrow) jch_ctx();
    if (!ctx) return jch_fail(nullptr, JCH_ENOMEM, "jch_ctx_create: host allocation failed");
    ctx->device = device_id;
    ctx->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // JCH_CUS=<count> (measurement knob, A/B runs): size every persistent grid as if the device had this many CUs
    if (const char *e_cus = getenv("JCH_CUS")) { const int v = atoi(e_cus); if (v >= 8 && v <= ctx->cus) ctx->cus = v; }
    e = hipSetDevice(device_id);
    if (e == hipSuccess) {
        if (stream) {
            ctx->stream = (hipStream_t)stream;
        } else {
            e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
            ctx->own_stream = true;
        }
    }
    if (e != hipSuccess) {
        delete ctx;
        return jch_fail(nullptr, JCH_EHIP, "jch_ctx_create: %s", hipGetErrorString(e));
    }
    if (const char *s = getenv("JCH_SWEEP_BLOCKS_PER_CU")) ctx->sweep_blocks_per_cu = atoi(s);
    *out = ctx;
    return JCH_OK;
}

static void free_buf(jch_buf &b)
{
    if (b.ptr) (void)hipFree(b.ptr);
    b.ptr = nullptr;
    b.bytes = 0;
}

static jch_rccl g_rccl;

extern "C" int32_t jch_ctx_destroy(jch_ctx *ctx)
{
    if (!ctx) return JCH_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(ctx->comm);
    jch_p2p_destroy(ctx);
    for (jch_buf *b : {&ctx->gram, &ctx->xr, &ctx->yr, &ctx->xstage, &ctx->ystage, &ctx->wstage, &ctx->tbuf, &ctx->dnorm, &ctx->part,
                       &ctx->kpart, &ctx->small, &ctx->colpart, &ctx->gemm_b, &ctx->gemm_out, &ctx->xq, &ctx->tickets, &ctx->qz, &ctx->lw_work, &ctx->lw_xrm, &ctx->lvws, &ctx->lw_flags, &ctx->lw_screen})
        free_buf(*b);
    if (ctx->hstage) (void)hipHostFree(ctx->hstage);
    for (hipEvent_t ev : ctx->ev_pool) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->cev_pool) (void)hipEventDestroy(ev);
    if (ctx->aux_event) (void)hipEventDestroy(ctx->aux_event);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return JCH_OK;
}

int32_t jch_reserve(jch_ctx *ctx, jch_buf &b, size_t bytes)
{
    if (bytes == 0) bytes = 256;
    if (b.bytes >= bytes) return JCH_OK;
    if (b.ptr) {
        JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(b.ptr);
        b.ptr = nullptr;
        b.bytes = 0;
    }
    bytes = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&b.ptr, bytes);
    if (e != hipSuccess) {
        b.ptr = nullptr;
        return jch_fail(ctx, JCH_ENOMEM, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    b.bytes = bytes;
    static const bool trace = getenv("JCH_TRACE_ALLOC") != nullptr;
    if (trace && bytes >= (1u << 20)) fprintf(stderr, "[jch] workspace %p  %zu bytes\n", b.ptr, bytes);
    return JCH_OK;
}

int32_t jch_reserve_host(jch_ctx *ctx, size_t bytes)
{
    if (ctx->hstage_bytes >= bytes) return JCH_OK;
    if (ctx->hstage) {
        JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipHostFree(ctx->hstage);
        ctx->hstage = nullptr;
        ctx->hstage_bytes = 0;
    }
    bytes = (bytes + 4095) & ~(size_t)4095;
    hipError_t e = hipHostMalloc(&ctx->hstage, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        ctx->hstage = nullptr;
        return jch_fail(ctx, JCH_ENOMEM, "hipHostMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    ctx->hstage_bytes = bytes;
    return JCH_OK;
}

// ---- RCCL (dlopen: no link-time dependency; inside a torch process this resolves to the RCCL torch loaded) ----
static int32_t rccl_load(jch_ctx *ctx)
{
    if (g_rccl.handle) return JCH_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return jch_fail(ctx, JCH_ERCCL, "cannot dlopen librccl.so.1: %s", dlerror());
    g_rccl.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void **, int, jch_uid, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy || !g_rccl.GetErrorString)
        return jch_fail(ctx, JCH_ERCCL, "librccl is missing an expected symbol");
    g_rccl.handle = h;
    return JCH_OK;
}

extern "C" int32_t jch_comm_unique_id(void *uid128)
{
    if (!uid128) return jch_fail(nullptr, JCH_EINVAL, "jch_comm_unique_id: NULL");
    JCH_TRY(rccl_load(nullptr));
    int r = g_rccl.GetUniqueId(uid128);
    if (r != 0) return jch_fail(nullptr, JCH_ERCCL, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
    return JCH_OK;
}

extern "C" int32_t jch_ctx_comm_init(jch_ctx *ctx, const void *uid128, int32_t rank, int32_t nranks)
{
    if (!ctx) return JCH_EINVAL;
    if (!uid128 || nranks < 1 || rank < 0 || rank >= nranks)
        return jch_fail(ctx, JCH_EINVAL, "jch_ctx_comm_init: bad rank %d / nranks %d", rank, nranks);
    if (ctx->comm) return jch_fail(ctx, JCH_EINVAL, "jch_ctx_comm_init: ctx already has a communicator");
    JCH_TRY(rccl_load(ctx));
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    jch_uid id;
    memcpy(&id, uid128, sizeof id);
    int r = g_rccl.CommInitRank(&ctx->comm, nranks, id, rank);
    if (r != 0) {
        ctx->comm = nullptr;
        return jch_fail(ctx, JCH_ERCCL, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, g_rccl.GetErrorString(r));
    }
    ctx->rank = rank;
    ctx->nranks = nranks;
    return JCH_OK;
}

extern "C" int32_t jch_ctx_comm_info(const jch_ctx *ctx, int32_t *rank, int32_t *nranks)
{
    if (!ctx) return JCH_EINVAL;
    if (rank) *rank = ctx->rank;
    if (nranks) *nranks = ctx->nranks;
    return JCH_OK;
}

// ---- loopback communicator (tests): the ranks are host THREADS of one process, each with its own ctx (and stream) on
// the same GPU.  It exists so that the row-sharded code path of the library (global weight sum, all-reduced moments,
// XtY, per-LV [zp, tt]) can be executed and checked on a one-GPU box, where RCCL refuses two ranks on one device.
// The all-reduce is staged through host memory in rank order, so every rank ends with bit-identical sums — the same
// property the RCCL path has.
struct jch_loop_group {
    int nranks = 0;
    pthread_barrier_t bar;
    std::vector<double> stage;
};

extern "C" int32_t jch_loopback_group_create(int32_t nranks, void **out)
{
    if (!out || nranks < 1 || nranks > 64) return jch_fail(nullptr, JCH_EINVAL, "jch_loopback_group_create: bad arguments");
    jch_loop_group *g = new (std::nothrow) jch_loop_group();
    if (!g) return jch_fail(nullptr, JCH_ENOMEM, "jch_loopback_group_create: host allocation failed");
    g->nranks = nranks;
    if (pthread_barrier_init(&g->bar, nullptr, (unsigned)nranks) != 0) {
        delete g;
        return jch_fail(nullptr, JCH_EINVAL, "pthread_barrier_init failed");
    }
    *out = g;
    return JCH_OK;
}

extern "C" int32_t jch_loopback_group_destroy(void *grp)
{
    jch_loop_group *g = (jch_loop_group *)grp;
    if (!g) return JCH_OK;
    pthread_barrier_destroy(&g->bar);
    delete g;
    return JCH_OK;
}

extern "C" int32_t jch_ctx_comm_init_loopback(jch_ctx *ctx, void *grp, int32_t rank)
{
    if (!ctx) return JCH_EINVAL;
    jch_loop_group *g = (jch_loop_group *)grp;
    if (!g || rank < 0 || rank >= g->nranks) return jch_fail(ctx, JCH_EINVAL, "jch_ctx_comm_init_loopback: bad group / rank");
    if (ctx->comm || ctx->loop) return jch_fail(ctx, JCH_EINVAL, "jch_ctx_comm_init_loopback: ctx already has a communicator");
    ctx->loop = g;
    ctx->rank = rank;
    ctx->nranks = g->nranks;
    return JCH_OK;
}

static int32_t loopback_allreduce(jch_ctx *ctx, double *dev_buf, size_t count)
{
    jch_loop_group *g = (jch_loop_group *)ctx->loop;
    const size_t need = count * (size_t)g->nranks;
    pthread_barrier_wait(&g->bar);                       // everybody is done with the previous round's stage
    if (ctx->rank == 0 && g->stage.size() < need) g->stage.resize(need);
    pthread_barrier_wait(&g->bar);
    hipError_t e = hipMemcpyAsync(g->stage.data() + (size_t)ctx->rank * count, dev_buf, sizeof(double) * count,
                                  hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    pthread_barrier_wait(&g->bar);                       // (always reached, also on error: no rank may be left waiting)
    if (e != hipSuccess) return jch_fail(ctx, JCH_EHIP, "loopback all-reduce: %s", hipGetErrorString(e));
    ctx->loop_sum.assign(count, 0.0);
    for (int r = 0; r < g->nranks; ++r) {
        const double *src = g->stage.data() + (size_t)r * count;
        for (size_t i = 0; i < count; ++i) ctx->loop_sum[i] += src[i];
    }
    JCH_HIP(ctx, hipMemcpyAsync(dev_buf, ctx->loop_sum.data(), sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
    JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JCH_OK;
}

// messages up to this many doubles go through the P2P inbox when it is enabled (latency-bound sizes); larger ones
// (X'DX of the opt-in algorithm #2) stay on RCCL when a communicator exists
static const size_t P2P_SMALL = 32768;

static int32_t rccl_allreduce(jch_ctx *ctx, double *dev_buf, size_t count)
{
    int r = g_rccl.AllReduce(dev_buf, dev_buf, count, /*ncclDouble*/ 8, /*ncclSum*/ 0, ctx->comm, ctx->stream);
    if (r != 0) return jch_fail(ctx, JCH_ERCCL, "ncclAllReduce(%zu f64): %s", count, g_rccl.GetErrorString(r));
    return JCH_OK;
}

int32_t jch_allreduce_f64(jch_ctx *ctx, double *dev_buf, size_t count)
{
    if (count == 0) return JCH_OK;
    if (ctx->loop) {
        jch_coll_begin(ctx);
        const int32_t st = loopback_allreduce(ctx, dev_buf, count);
        jch_coll_end(ctx);
        if (ctx->coll_phase) ctx->coll_transport = JCH_TRANSPORT_LOOPBACK;
        return st;
    }
    if (ctx->p2p.ready && (count <= P2P_SMALL || !ctx->comm)) {   // (timed inside the inbox kernel: p2p.stats)
        if (ctx->coll_phase) ctx->coll_transport = JCH_TRANSPORT_INBOX;
        return jch_p2p_allreduce(ctx, dev_buf, count, 1, 0, dev_buf);
    }
    if (!ctx->comm) {
        if (ctx->nranks > 1) return jch_fail(ctx, JCH_ERCCL, "rank %d of %d has no enabled transport (jch_ctx_p2p_enable not called?)", ctx->rank, ctx->nranks);
        return JCH_OK;  // single rank: the local sum is the global sum
    }
    jch_coll_begin(ctx);
    const int32_t st = rccl_allreduce(ctx, dev_buf, count);
    jch_coll_end(ctx);
    if (ctx->coll_phase) ctx->coll_transport = JCH_TRANSPORT_RCCL;
    return st;
}

// ---- collective timing (profiling only) ---------------------------------------------------------------------------
static hipEvent_t coll_event(jch_ctx *ctx)
{
    if (ctx->cev_used == ctx->cev_pool.size()) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return nullptr;
        ctx->cev_pool.push_back(ev);
    }
    hipEvent_t ev = ctx->cev_pool[ctx->cev_used++];
    (void)hipEventRecord(ev, ctx->stream);
    return ev;
}
void jch_coll_begin(jch_ctx *ctx)
{
    // pairs are recorded between the start of a fit and the collection of its profile only: the collectives of the other
    // entry points (score sums, column statistics, covariances) would otherwise pile up in the pool, tagged as LV-loop
    if (!ctx->profiling || !ctx->coll_in_fit || ctx->cev_used >= 8192) return;
    if (ctx->cev_used & 1) return;    // (unbalanced: a failed call left a begin behind — keep the pairing)
    if (coll_event(ctx)) ctx->cev_phase.push_back(ctx->coll_phase);
}
void jch_coll_end(jch_ctx *ctx)
{
    if (!ctx->profiling || !(ctx->cev_used & 1)) return;
    (void)coll_event(ctx);
}
void jch_coll_reset(jch_ctx *ctx)
{
    ctx->cev_used = 0;
    ctx->cev_phase.clear();
    ctx->coll_phase = 0;
    ctx->coll_in_fit = true;
    ctx->coll_transport = JCH_TRANSPORT_NONE;
    if (ctx->profiling && ctx->p2p.stats) (void)hipMemsetAsync(ctx->p2p.stats, 0, 64, ctx->stream);
}
void jch_coll_collect(jch_ctx *ctx, jch_profile &pr)
{
    pr.collective_ms = pr.prologue_collective_ms = pr.collective_wait_ms = 0.0;
    pr.collective_calls = 0;
    pr.collective_transport = ctx->coll_transport;
    ctx->coll_in_fit = false;          // the fit is over: later collectives on this ctx are not its LV loop
    ctx->coll_phase = 0;
    if (!ctx->profiling) return;
    for (size_t i = 0; 2 * i + 1 < ctx->cev_used && i < ctx->cev_phase.size(); ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->cev_pool[2 * i], ctx->cev_pool[2 * i + 1]) != hipSuccess) continue;
        if (ctx->cev_phase[i]) { pr.collective_ms += ms; pr.collective_calls++; }
        else pr.prologue_collective_ms += ms;
    }
    if (ctx->p2p.stats) {   // inbox kernels (stand-alone or fused): device tick counters, 100 MHz
        unsigned long long h[8] = {};
        if (hipMemcpy(h, ctx->p2p.stats, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
            const double nr = (double)std::max(1, ctx->p2p.nranks);
            pr.prologue_collective_ms += (double)h[0] * 1e-5;
            pr.collective_ms += (double)h[4] * 1e-5;
            pr.collective_wait_ms += (double)h[5] * 1e-5 / nr;   // every poller added its own wait: average over the nranks flags
            pr.collective_calls += (int32_t)h[6];
        }
    }
}

extern "C" int32_t jch_ctx_allreduce_probe(jch_ctx *ctx, int32_t transport, double *vec, int64_t count, int32_t iters, double *avg_us)
{
    if (!ctx) return JCH_EINVAL;
    if (!vec || count < 1 || count > (1 << 24) || iters < 1 || iters > 100000)
        return jch_fail(ctx, JCH_EINVAL, "jch_ctx_allreduce_probe: bad arguments");
    if (transport == JCH_TRANSPORT_RCCL && !ctx->comm) return jch_fail(ctx, JCH_EINVAL, "jch_ctx_allreduce_probe: no RCCL communicator on this ctx");
    if (transport == JCH_TRANSPORT_INBOX && !ctx->p2p.tested) return jch_fail(ctx, JCH_EINVAL, "jch_ctx_allreduce_probe: the inbox has not passed its self-test");
    if (transport == JCH_TRANSPORT_LOOPBACK && !ctx->loop) return jch_fail(ctx, JCH_EINVAL, "jch_ctx_allreduce_probe: no loopback group on this ctx");
    if (transport != JCH_TRANSPORT_NONE && transport != JCH_TRANSPORT_RCCL && transport != JCH_TRANSPORT_INBOX && transport != JCH_TRANSPORT_LOOPBACK)
        return jch_fail(ctx, JCH_EINVAL, "jch_ctx_allreduce_probe: unknown transport %d", transport);
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    JCH_TRY(jch_reserve(ctx, ctx->colpart, sizeof(double) * 2 * (size_t)count));
    double *src = (double *)ctx->colpart.ptr, *work = src + count;
    JCH_HIP(ctx, hipMemcpyAsync(src, vec, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    JCH_HIP(ctx, hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return jch_fail(ctx, JCH_EHIP, "hipEventCreate failed"); }
    const bool was_prof = ctx->profiling;
    ctx->profiling = false;                        // (no event pairs / tick counters from inside the probe)
    int32_t st = JCH_OK;
    for (int it = 0; it < iters && st == JCH_OK; ++it) {
        if (it == (iters > 1 ? 1 : 0)) (void)hipEventRecord(e0, ctx->stream);
        if (hipMemcpyAsync(work, src, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { st = jch_fail(ctx, JCH_EHIP, "probe copy failed"); break; }
        if (transport == JCH_TRANSPORT_RCCL) st = rccl_allreduce(ctx, work, (size_t)count);
        else if (transport == JCH_TRANSPORT_INBOX) st = jch_p2p_allreduce(ctx, work, (size_t)count, 1, 0, work);
        else st = jch_allreduce_f64(ctx, work, (size_t)count);
    }
    (void)hipEventRecord(e1, ctx->stream);
    ctx->profiling = was_prof;
    hipError_t he = hipMemcpyAsync(vec, work, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
    float ms = 0.f;
    if (he == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (st != JCH_OK) return st;
    if (he != hipSuccess) return jch_fail(ctx, JCH_EHIP, "jch_ctx_allreduce_probe: %s", hipGetErrorString(he));
    JCH_TRY(jch_p2p_check(ctx));
    if (avg_us) *avg_us = (double)ms * 1e3 / (double)(iters > 1 ? iters - 1 : 1);
    return JCH_OK;
}

// ---- profiling ------------------------------------------------------------------------------------------
hipEvent_t jch_ev(jch_ctx *ctx)
{
    if (!ctx->profiling) return nullptr;
    if (ctx->ev_used == ctx->ev_pool.size()) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return nullptr;
        ctx->ev_pool.push_back(ev);
    }
    hipEvent_t ev = ctx->ev_pool[ctx->ev_used++];
    if (hipEventRecord(ev, ctx->stream) != hipSuccess) return nullptr;
    return ev;
}

// An event record costs the stream ~3 us (measured: 50 of them = 0.16 ms per cfg2 fit, 6 % of a 125 k-row share).  With a stride N > 1
// only every N-th launch of the plskern-shaped sweeps (f64 v2 / bf16 v2 launchers) is bracketed; the counter runs across fits, so with
// N coprime to nlv every LV's sweep is sampled in turn.
bool jch_prof_sample(jch_ctx *ctx)
{
    if (!ctx->profiling) return false;
    const bool take = ctx->prof_stride <= 1 || (ctx->prof_seq++ % (unsigned)ctx->prof_stride) == 0u;
    if (take) ctx->sweeps_timed++;
    return take;
}

extern "C" int32_t jch_ctx_set_profiling(jch_ctx *ctx, int32_t enable)
{
    if (!ctx) return JCH_EINVAL;
    ctx->profiling = enable != 0;
    ctx->prof_stride = enable > 1 ? enable : 1;
    ctx->coll_in_fit = false;          // collective pairs are recorded from the next fit's start on
    return JCH_OK;
}

extern "C" int32_t jch_ctx_get_profile(const jch_ctx *ctx, jch_profile *out)
{
    if (!ctx || !out) return JCH_EINVAL;
    *out = ctx->prof;
    return JCH_OK;
}

extern "C" int32_t jch_ctx_get_counter(const jch_ctx *ctx, int32_t which, int64_t *out)
{
    if (!ctx || !out) return JCH_EINVAL;
    if (which == JCH_COUNTER_PIVOT_REFITS) { *out = ctx->pivot_refits; return JCH_OK; }
    if (which == JCH_COUNTER_LOCW_REFITS) { *out = ctx->locw_refits; return JCH_OK; }
    if (which == JCH_COUNTER_KNN_SCREENED) { *out = ctx->knn_screened; return JCH_OK; }
    if (which == JCH_COUNTER_KNN_SCREEN_REDONE) { *out = ctx->knn_screen_redone; return JCH_OK; }
    if (which == JCH_COUNTER_XCOPY_REUSED) { *out = ctx->xcopy_reused; return JCH_OK; }
    if (which == JCH_COUNTER_SWEEPS_TIMED) { *out = ctx->sweeps_timed; return JCH_OK; }
    return JCH_EINVAL;
}

int32_t jch_allreduce_slices(jch_ctx *ctx, double *zt, int m, int nslice, int ldz, int *nslice_out)
{
    *nslice_out = nslice;
    if (ctx->nranks <= 1 && !ctx->p2p.ready) return JCH_OK;   // (a one-rank inbox is allowed: it is how the transport's own cost is measured)
    if (ctx->p2p.ready && !ctx->loop) {   // the inbox kernel adds the slices itself: a 4 KB message instead of 33 KB
        *nslice_out = 1;
        if (ctx->coll_phase) ctx->coll_transport = JCH_TRANSPORT_INBOX;
        return jch_p2p_allreduce(ctx, zt, (size_t)m, nslice, ldz, zt);
    }
    return jch_allreduce_f64(ctx, zt, nslice > 1 ? (size_t)nslice * ldz : (size_t)m);
}
