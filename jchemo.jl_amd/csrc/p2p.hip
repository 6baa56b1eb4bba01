// P2P "inbox" all-reduce over xGMI (the per-LV message is 4 KB, so the collective is pure latency).
//
// Every rank owns an inbox in fine-grained device memory, shared with the other ranks' processes through HIP IPC
// handles.  One all-reduce = ONE single-workgroup kernel per rank:
//   1. sum the local partial slices and store the vector into slot [parity][my rank] of EVERY rank's inbox
//      (direct peer stores over xGMI), fence, then publish the epoch number in flag [parity][my rank] of every inbox;
//   2. wait (bounded) until all flags of the own inbox carry this epoch;
//   3. add the nranks slots in rank order -> the same bits on every rank (the property the RCCL path has).
// Two parities suffice: a rank can only be two epochs ahead of a peer after that peer has published the epoch in
// between, which it does at the START of its next call, i.e. after it finished reading the older parity.
// A wait that exceeds the timeout sets a sticky status word (device + pinned host copy) and every later kernel of
// this transport returns at once: a lost peer can never hang the GPU; the fit then reports JCH_ERCCL.
// The host side (bench.py / tests) exchanges the 64-byte IPC handles, runs the self-test on all ranks and only then
// enables the transport (jch_ctx_p2p_enable); RCCL stays the transport for large messages and the fallback.
#include <stdlib.h>

#include "jch_internal.h"
#include "p2p_dev.h"

#define P2P_NT 1024

struct p2p_args {
    p2p_dev t;
    const double *src;              // [nslice][ldz] partial slices (device)
    double *dst;                    // [count] result (may alias src)
    int count, nslice, ldz;
};

__global__ __launch_bounds__(P2P_NT) void k_p2p_allreduce(p2p_args g)
{
    __shared__ int bail;
    const int tid = threadIdx.x;
    const int par = (int)(g.t.epoch & 1ull);
    char *mine = g.t.peer[g.t.rank];
    if (tid == 0) bail = __hip_atomic_load(p2p_status(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
    __syncthreads();
    if (bail) return;
    const long long ts0 = p2p_stat_begin(g.t, tid);
    // ---- 1. local slice sum, scattered into slot [par][rank] of every inbox
    for (int i = tid; i < g.count; i += P2P_NT) {
        double v = 0.0;
        for (int sl = 0; sl < g.nslice; ++sl) v += g.src[(size_t)sl * g.ldz + i];
        for (int r = 0; r < g.t.nranks; ++r) p2p_slot(g.t.peer[r], par, g.t.rank, g.t.nranks, g.t.cap)[i] = v;
    }
    __threadfence_system();
    __syncthreads();
    // ---- 2. publish the epoch, wait for every rank's flag in the own inbox (bounded)
    p2p_publish_and_wait(g.t, tid);
    __syncthreads();
    if (tid == 0) bail = __hip_atomic_load(p2p_status(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
    __syncthreads();
    if (bail) return;
    // ---- 3. ordered sum of the nranks slots (system-scope loads: the slots were written by other devices)
    for (int i = tid; i < g.count; i += P2P_NT) {
        double s = 0.0;
        for (int r = 0; r < g.t.nranks; ++r) s += p2p_load_slot(p2p_slot(mine, par, r, g.t.nranks, g.t.cap) + i);
        g.dst[i] = s;
    }
    p2p_stat_end(g.t, tid, ts0);
}

static size_t p2p_bytes(int nranks, size_t cap) { return P2P_HDR_BYTES + sizeof(double) * 2 * (size_t)nranks * cap; }

extern "C" int32_t jch_ctx_p2p_export(jch_ctx *ctx, int32_t nranks, void *handle64)
{
    if (!ctx) return JCH_EINVAL;
    if (!handle64 || nranks < 1 || nranks > JCH_P2P_MAXR)
        return jch_fail(ctx, JCH_EINVAL, "jch_ctx_p2p_export: nranks %d outside [1, %d]", nranks, JCH_P2P_MAXR);
    jch_p2p &t = ctx->p2p;
    if (t.local) return jch_fail(ctx, JCH_EINVAL, "jch_ctx_p2p_export: inbox already allocated");
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    t.cap = 16384;
    // never below the widest fused per-LV message (the fused kernels of smallstate_fast.hip / bf16.hip write ldr + 1 + 16
    // resp. bf_ldr + 2 doubles into one slot without chunking; jch_p2p_allreduce chunks by cap)
    if (const char *e = getenv("JCH_P2P_CAP")) t.cap = (size_t)std::max(JCH_SWEEP_MAXP + 64, atoi(e));
    t.nranks = nranks;
    const size_t bytes = p2p_bytes(nranks, t.cap);
    JCH_HIP(ctx, hipExtMallocWithFlags(&t.local, bytes, hipDeviceMallocFinegrained));
    JCH_HIP(ctx, hipMemset(t.local, 0, bytes));
    JCH_HIP(ctx, hipMalloc((void **)&t.stats, 64));
    JCH_HIP(ctx, hipMemset(t.stats, 0, 64));
    JCH_HIP(ctx, hipHostMalloc((void **)&t.host_status, 64, hipHostMallocMapped));
    *t.host_status = 0ull;
    JCH_HIP(ctx, hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    JCH_HIP(ctx, hipIpcGetMemHandle(&h, t.local));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    memcpy(handle64, &h, 64);
    return JCH_OK;
}

// fills the device view of the transport for the NEXT all-reduce (advances the epoch)
void jch_p2p_next(jch_ctx *ctx, p2p_dev *out)
{
    jch_p2p &t = ctx->p2p;
    for (int r = 0; r < JCH_P2P_MAXR; ++r) out->peer[r] = r < t.nranks ? (char *)t.peer[r] : nullptr;
    out->host_status = t.host_status_dev;
    out->epoch = ++t.epoch;
    out->timeout_ticks = t.timeout_ticks;
    out->cap = t.cap;
    out->nranks = t.nranks; out->rank = t.rank;
    out->stats = (ctx->profiling && ctx->coll_in_fit && t.stats) ? t.stats + 4 * (ctx->coll_phase ? 1 : 0) : nullptr;
}

static int32_t p2p_launch(jch_ctx *ctx, const double *src, int count, int nslice, int ldz, double *dst)
{
    p2p_args g;
    jch_p2p_next(ctx, &g.t);
    g.src = src; g.dst = dst; g.count = count; g.nslice = nslice; g.ldz = ldz;
    hipLaunchKernelGGL(k_p2p_allreduce, dim3(1), dim3(P2P_NT), 0, ctx->stream, g);
    JCH_HIP(ctx, hipGetLastError());
    return JCH_OK;
}

extern "C" int32_t jch_ctx_p2p_import(jch_ctx *ctx, const void *handles, int32_t rank, int32_t nranks, uint32_t flags)
{
    if (!ctx) return JCH_EINVAL;
    (void)flags;
    jch_p2p &t = ctx->p2p;
    if (!t.local) return jch_fail(ctx, JCH_EINVAL, "jch_ctx_p2p_import: call jch_ctx_p2p_export first");
    if (!handles || nranks != t.nranks || rank < 0 || rank >= nranks)
        return jch_fail(ctx, JCH_EINVAL, "jch_ctx_p2p_import: bad rank %d / nranks %d (exported for %d)", rank, nranks, t.nranks);
    if (ctx->comm && (ctx->rank != rank || ctx->nranks != nranks))
        return jch_fail(ctx, JCH_EINVAL, "jch_ctx_p2p_import: rank/nranks differ from the RCCL communicator's");
    if (ctx->loop) return jch_fail(ctx, JCH_EINVAL, "jch_ctx_p2p_import: ctx uses the loopback communicator");
    JCH_HIP(ctx, hipSetDevice(ctx->device));
    t.rank = rank;
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) { t.peer[r] = t.local; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + 64 * (size_t)r, 64);
        void *ptr = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return jch_fail(ctx, JCH_EHIP, "hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
        t.peer[r] = ptr;
        t.opened[r] = true;
    }
    JCH_HIP(ctx, hipHostGetDevicePointer((void **)&t.host_status_dev, t.host_status, 0));
    double ms = 10000.0;   // generous: a spurious timeout would abort a fit; a lost peer still cannot hang the GPU
    if (const char *e = getenv("JCH_P2P_TIMEOUT_MS")) ms = atof(e);
    t.timeout_ticks = (long long)(ms * 1e5);   // wall_clock64 runs at 100 MHz
    if (!ctx->comm) { ctx->rank = rank; ctx->nranks = nranks; }
    // ---- self-test (collective: every rank is inside this call): both parities, exact small-integer sums
    const int cnt = 1000;
    JCH_TRY(jch_reserve(ctx, ctx->colpart, sizeof(double) * 2 * cnt));
    double *dbuf = (double *)ctx->colpart.ptr;
    std::vector<double> h(2 * cnt);
    bool ok = true;
    for (int round = 0; round < 4 && ok; ++round) {
        for (int i = 0; i < cnt; ++i) { h[i] = (double)((rank + 1) * (i % 7 + 1 + round)); h[cnt + i] = 1.0; }   // two slices
        JCH_HIP(ctx, hipMemcpyAsync(dbuf, h.data(), sizeof(double) * 2 * cnt, hipMemcpyHostToDevice, ctx->stream));
        JCH_TRY(p2p_launch(ctx, dbuf, cnt, 2, cnt, dbuf));
        JCH_HIP(ctx, hipMemcpyAsync(h.data(), dbuf, sizeof(double) * cnt, hipMemcpyDeviceToHost, ctx->stream));
        JCH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (*t.host_status != 0ull) { ok = false; break; }
        const double tri = 0.5 * nranks * (nranks + 1);
        for (int i = 0; i < cnt; ++i)
            if (h[i] != tri * (i % 7 + 1 + round) + nranks) { ok = false; break; }
    }
    t.tested = ok;
    if (!ok) return jch_fail(ctx, JCH_ERCCL, "P2P inbox self-test failed on rank %d (status %llu)", rank, (unsigned long long)*t.host_status);
    return JCH_OK;
}

extern "C" int32_t jch_ctx_p2p_enable(jch_ctx *ctx, int32_t on)
{
    if (!ctx) return JCH_EINVAL;
    if (on && !ctx->p2p.tested) return jch_fail(ctx, JCH_EINVAL, "jch_ctx_p2p_enable: the transport has not passed its self-test");
    ctx->p2p.ready = on != 0;
    return JCH_OK;
}

void jch_p2p_destroy(jch_ctx *ctx)
{
    jch_p2p &t = ctx->p2p;
    for (int r = 0; r < JCH_P2P_MAXR; ++r)
        if (t.opened[r] && t.peer[r]) (void)hipIpcCloseMemHandle(t.peer[r]);
    if (t.local) (void)hipFree(t.local);
    if (t.stats) (void)hipFree(t.stats);
    if (t.host_status) (void)hipHostFree(t.host_status);
    t = jch_p2p{};
}

// sticky error of the transport (a wait timed out): checked by the fits after their final stream sync
int32_t jch_p2p_check(jch_ctx *ctx)
{
    if (ctx->p2p.host_status && *ctx->p2p.host_status != 0ull) {
        ctx->p2p.ready = false;
        return jch_fail(ctx, JCH_ERCCL, "P2P inbox all-reduce timed out at epoch %llu (a peer did not arrive); transport disabled",
                        (unsigned long long)*ctx->p2p.host_status);
    }
    return JCH_OK;
}

// All-reduce of `count` doubles whose local value is the sum of `nslice` slices (ld ldz) at src; result -> dst[0..count)
int32_t jch_p2p_allreduce(jch_ctx *ctx, const double *src, size_t count, int nslice, int ldz, double *dst)
{
    jch_p2p &t = ctx->p2p;
    for (size_t off = 0; off < count; off += t.cap) {
        const int c = (int)std::min(t.cap, count - off);
        JCH_TRY(p2p_launch(ctx, src + off, c, nslice, ldz, dst + off));
    }
    return JCH_OK;
}
