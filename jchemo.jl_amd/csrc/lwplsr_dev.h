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
};

// lwplsr_kspace.hip: _feasible: the k-space kernel can take this shape (k <= 208, q <= 8, nlv <= 48, p <= 2048);
// _supported: ... and is expected to beat the p-space kernel there
bool jch_locw_kspace_feasible(const locw_args &g);
bool jch_locw_kspace_supported(const locw_args &g);
int32_t jch_launch_locw_kspace(jch_ctx *ctx, locw_args &g);
