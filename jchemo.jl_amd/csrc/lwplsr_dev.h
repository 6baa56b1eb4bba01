// Arguments shared by the batched local-fit kernels of the kNN-LWPLSR prediction path (lwplsr.hip: k_locw_plskern, the
// p-space kernel; lwplsr_kspace.hip: k_locw_kspace, the Gram / k-space kernel) — src/locwlv.jl:9-48.
#pragma once
#include "jch_internal.h"

struct locw_args {
    const double *Xrm; int ldr; int p;      // row-major training X (uncentred)
    const double *Y; int64_t ldy; int q;    // training Y (column-major n x q)
    const double *Xq; int64_t ldxq; int m;  // queries, column-major m x p
    const int *ind; const double *w; int k; // neighbours / weights [m][k]
    int scal, nlv_lo, nlv_hi;
    double *scratch; size_t slab;           // per-block global scratch (p-space kernel: Xg [k][ldr], P [nlv][ldr], R [nlv][ldr]; k-space kernel: [ldr] column stds)
    double *pred;                           // [m][le][q], le = nlv_hi - nlv_lo + 1
    int dbg;                                // measurement switches (JCH_LOCW_DBG; results then wrong by design): 1 = every query gathers rows 0 .. k-1
    int *flags;                             // [m] or null.  k-space kernel: flags[i] = 1 when query i lies too far from its neighbours for the
                                            // Gram matrix about the query row (pivot check, lwplsr_kspace.hip): the caller refits those queries
};

// lwplsr_kspace.hip: _feasible: the k-space kernel can take this shape (k <= 208, q <= 8, nlv <= 48, p <= 2048);
// _supported: ... and is expected to beat the p-space kernel there
bool jch_locw_kspace_feasible(const locw_args &g);
bool jch_locw_kspace_supported(const locw_args &g);
int32_t jch_launch_locw_kspace(jch_ctx *ctx, locw_args &g);

// Arguments of the kNN + weights stage (lwplsr.hip: k_knn_scan / k_knn_finish; lwplsr_generic.hip: k_knn_generic) — src/getknn.jl:29-57,
// src/wdist.jl:64-75.
struct knn_args {
    const double *Zt; int64_t ldzt; int64_t n;   // train scores, column-major n x dd
    const double *Zq; int64_t ldzq; int m;       // query scores, column-major m x dd
    int dd, k;
    double h, cri, tol;
    int *ind;      // [m][k]
    double *dist;  // [m][k]
    double *w;     // [m][k]
    int nseg;      // the training rows are scanned in nseg segments by different workgroups (block b: segment b % nseg, query group b / nseg)
    double *ckey;  // [m][nseg][k] squared distances of every segment's k best (ascending; +inf beyond the segment's rows)
    int *cidx;     // [m][nseg][k]
    int dbg;       // measurement switch (JCH_KNN_DBG; results then wrong by design): 1 = the bar starts at -inf (no candidate is ever kept: the bare scan)
};

// lwplsr_generic.hip: the paths WITHOUT shape limits (any k <= n, any p, q, nlv) behind the batched kernels' envelope.
// kNN + weights of all m queries: exact selection of the k smallest distances per query, (distance, index) order, wdist weights
int32_t jch_launch_knn_generic(jch_ctx *ctx, const knn_args &a);
// local fits one query at a time: gather the neighbour rows, jch_plskern_fit on them, jch_predict on the query row (the
// reference's own schedule, src/locwlv.jl:18-39); dpred [m][le][q] device
int32_t jch_lw_generic_fits(jch_ctx *ctx, const locw_args &g, int64_t n, const int *only = nullptr /*host: query indices to fit (null: all)*/, int n_only = 0);

