// Device-side layout of the P2P inbox (p2p.hip) shared with the small-state kernel that fuses the all-reduce
// (smallstate_fast.hip).  Header [flags [2][JCH_P2P_MAXR] u64 | status u64 at byte 256], then slots [2][nranks][cap] f64.
#pragma once
#include "jch_internal.h"

#define P2P_MAXBLK 128           // blocks of a kernel that exchanges per block (k_lv_spread: (p + 15) / 16 <= 128 at p <= 2048)
#define P2P_HDR_BYTES (4096 + 2 * JCH_P2P_MAXR * P2P_MAXBLK * 8)   // [0, 4096): flags + status; then the per-block flags [2][MAXR][MAXBLK] u64

struct p2p_dev {
    char *peer[JCH_P2P_MAXR];       // inbox base of every rank as mapped into THIS process (peer[rank] = own)
    unsigned long long *host_status;
    unsigned long long epoch;
    long long timeout_ticks;        // wall_clock64 ticks (100 MHz)
    size_t cap;                     // doubles per (parity, rank) slot
    int nranks, rank;
    unsigned long long *stats;      // [4] tick counters of the current phase (null: not profiling): exchange, polling, calls
};
// thread 0 brackets an exchange with these (wall_clock64: 100 MHz, constant rate)
__device__ __forceinline__ long long p2p_stat_begin(const p2p_dev &g, int tid) { return (g.stats && tid == 0) ? wall_clock64() : 0; }
__device__ __forceinline__ void p2p_stat_end(const p2p_dev &g, int tid, long long t0)
{
    if (g.stats && tid == 0) {
        atomicAdd(g.stats, (unsigned long long)(wall_clock64() - t0));
        atomicAdd(g.stats + 2, 1ull);
    }
}

__device__ __forceinline__ unsigned long long *p2p_flag(char *base, int par, int r)
{
    return reinterpret_cast<unsigned long long *>(base) + par * JCH_P2P_MAXR + r;
}
__device__ __forceinline__ double *p2p_slot(char *base, int par, int r, int nranks, size_t cap)
{
    return reinterpret_cast<double *>(base + P2P_HDR_BYTES) + ((size_t)par * nranks + r) * cap;
}
// flag of block `blk` of rank r (per-block exchange: every block of a multi-block kernel runs the publish / wait protocol for its
// own piece of the message, in its own piece of the slots)
__device__ __forceinline__ unsigned long long *p2p_bflag(char *base, int par, int r, int blk)
{
    return reinterpret_cast<unsigned long long *>(base + 4096) + ((size_t)par * JCH_P2P_MAXR + r) * P2P_MAXBLK + blk;
}
__device__ __forceinline__ unsigned long long *p2p_status(char *mine) { return reinterpret_cast<unsigned long long *>(mine + 256); }

// publish this rank's epoch in every inbox (threads 0..nranks-1), then wait — bounded — for every rank's flag in the own
// inbox; a timeout sets the sticky status word (device + pinned host copy).  Call with all threads; data stores to the
// peers must be followed by __threadfence_system() + __syncthreads() before this.
__device__ __forceinline__ void p2p_publish_and_wait(const p2p_dev &g, int tid)
{
    const int par = (int)(g.epoch & 1ull);
    char *mine = g.peer[g.rank];
    if (tid < g.nranks) {
        __hip_atomic_store(p2p_flag(g.peer[tid], par, g.rank), g.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        unsigned long long *f = p2p_flag(mine, par, tid);
        const long long t0 = wall_clock64();
        bool ok = false;
        for (;;) {
            ok = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == g.epoch;
            if (ok || wall_clock64() - t0 > g.timeout_ticks) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if (g.stats) atomicAdd(g.stats + 1, (unsigned long long)(wall_clock64() - t0));   // (summed over the nranks pollers)
        if (!ok) {
            __hip_atomic_store(p2p_status(mine), g.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g.host_status, g.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__device__ __forceinline__ double p2p_load_slot(const double *p)
{
    const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return __longlong_as_double((long long)bits);
}

// per-block version of p2p_publish_and_wait: flags [par][rank][blk]
__device__ __forceinline__ void p2p_publish_and_wait_block(const p2p_dev &g, int tid, int blk)
{
    const int par = (int)(g.epoch & 1ull);
    char *mine = g.peer[g.rank];
    if (tid < g.nranks) {
        __hip_atomic_store(p2p_bflag(g.peer[tid], par, g.rank, blk), g.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        unsigned long long *f = p2p_bflag(mine, par, tid, blk);
        const long long t0 = wall_clock64();
        bool ok = false;
        for (;;) {
            ok = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == g.epoch;
            if (ok || wall_clock64() - t0 > g.timeout_ticks) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if (g.stats && blk == 0) atomicAdd(g.stats + 1, (unsigned long long)(wall_clock64() - t0));
        if (!ok) {
            __hip_atomic_store(p2p_status(mine), g.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g.host_status, g.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
