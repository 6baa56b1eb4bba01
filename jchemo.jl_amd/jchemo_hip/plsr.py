"""Host-side mirror of the reference's operator interface for the PLS hot path.

Same names, argument meaning and error behaviour as Jchemo.jl (paths relative to /root/reference):
    plskern / plskern_ (= `plskern!`)      src/plskern.jl:106-178
    plsnipals / plsnipals_ (= `plsnipals!`) src/plsnipals.jl:31-97
    transform / coef / predict / summary    src/plskern.jl:187-260
    Plsr                                    src/plskern.jl:1-14
All n-sized arithmetic runs in libjchemo_hip.so (HIP, gfx950); numpy is used only for p x q glue
(`coef`) exactly as the Julia wrapper does it on the host.  X / Y may be numpy arrays (host) or torch
CUDA tensors (device-resident; column-major is Julia's layout, so a (n, p) torch tensor must satisfy
stride(0) == 1 — use `colmajor_empty`).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence, Union

import numpy as np

from . import _lib
from ._lib import Context, JchError, PlsDesc, default_context

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


@dataclass
class Plsr:
    """Field names and shapes equal the reference's `Plsr` (src/plskern.jl:1-14).  `T` and `weights` are
    numpy arrays for host fits and torch CUDA tensors for device-resident fits."""
    T: object
    P: np.ndarray
    R: np.ndarray
    W: np.ndarray
    C: np.ndarray
    TT: np.ndarray
    xmeans: np.ndarray
    xscales: np.ndarray
    ymeans: np.ndarray
    yscales: np.ndarray
    weights: object
    niter: Optional[np.ndarray] = None


# ---------------------------------------------------------------------------------- array plumbing
def _is_torch(a) -> bool:
    return torch is not None and isinstance(a, torch.Tensor)


def colmajor_empty(n: int, p: int, device="cuda:0", dtype=None):
    """(n, p) torch tensor (float64 by default) with Julia's column-major strides (1, n)."""
    return torch.empty((p, n), dtype=dtype or torch.float64, device=device).t()


def ensure_mat(X):
    """src/utility.jl:544-548: vector -> n x 1, number -> 1 x 1 (no copy when already a matrix)."""
    if _is_torch(X):
        if X.dim() == 1:
            return X.reshape(-1, 1)
        if X.dim() == 0:
            return X.reshape(1, 1)
        return X
    X = np.asarray(X)
    if X.ndim == 0:
        return X.reshape(1, 1)
    if X.ndim == 1:
        return X.reshape(-1, 1)
    return X


def _addr_ld(a, allow_bf16=False):
    """(address, leading dimension) of a column-major float64 matrix view; raises if not column-major."""
    if _is_torch(a):
        if a.dtype != torch.float64 and not (allow_bf16 and a.dtype == torch.bfloat16):
            raise TypeError("expected a float64 tensor")
        n = a.shape[0]
        if a.dim() == 1:
            if a.stride(0) != 1 and n > 1:
                raise ValueError("vector must be contiguous")
            return a.data_ptr(), max(n, 1)
        if n > 1 and a.stride(0) != 1:
            raise ValueError("device matrix must be column-major (stride(0) == 1); see colmajor_empty()")
        ld = a.stride(1) if a.shape[1] > 1 else max(n, 1)
        if ld < n:
            raise ValueError("bad leading dimension")
        return a.data_ptr(), ld
    if a.dtype != np.float64:
        raise TypeError("expected a float64 array")
    n = a.shape[0]
    if a.ndim == 1:
        if not a.flags.c_contiguous:
            raise ValueError("vector must be contiguous")
        return a.ctypes.data, max(n, 1)
    if not a.flags.f_contiguous:
        raise ValueError("host matrix must be Fortran-ordered (column-major)")
    return a.ctypes.data, max(n, 1)


def _as_colmajor_copy(a):
    a = ensure_mat(a)
    if _is_torch(a):
        out = colmajor_empty(a.shape[0], a.shape[1], a.device)
        out.copy_(a.to(torch.float64))
        return out
    return np.array(a, dtype=np.float64, order="F", copy=True)


def _as_colmajor_view(a):
    """For the `!` variants: the caller's own storage must be usable in place."""
    a = ensure_mat(a)
    _addr_ld(a)  # raises if not column-major float64
    return a


def _np(a):
    return a.ctypes.data if a is not None else None


# ---------------------------------------------------------------------------------- fits
def _fit(entry: str, X, Y, weights, nlv: int, scal: bool, inplace: bool, ctx: Optional[Context], variant: int = 0,
         wold: Optional[tuple] = None, ext_scales: Optional[tuple] = None):
    dev = _is_torch(X)
    if dev != _is_torch(Y):
        raise TypeError("X and Y must both be host arrays or both device tensors")
    if dev and not X.is_cuda:
        raise TypeError("torch inputs must live on the GPU (host data: pass numpy arrays)")
    n, p = X.shape
    q = Y.shape[1]
    if Y.shape[0] != n:
        raise ValueError(f"DimensionMismatch: X has {n} rows, Y has {Y.shape[0]}")
    device = X.device.index if dev else 0
    ctx = ctx or default_context(device or 0)
    if weights is None:
        w_arr, w_addr = None, None
    elif dev:
        w_arr = weights if _is_torch(weights) else torch.as_tensor(np.asarray(weights, dtype=np.float64), device=X.device)
        w_arr = w_arr.to(torch.float64).contiguous()
        w_addr = w_arr.data_ptr()
    else:
        w_arr = np.ascontiguousarray(np.asarray(weights, dtype=np.float64).reshape(-1))
        w_addr = w_arr.ctypes.data
    if w_arr is not None and w_arr.shape[0] != n:
        raise ValueError(f"DimensionMismatch: weights has {w_arr.shape[0]} entries, X has {n} rows")
    kmax = max(1, min(p, int(nlv)))  # global-n clamp happens in the library (nlv_out)
    if dev:
        T = colmajor_empty(n, kmax, X.device)
        wn = torch.empty(n, dtype=torch.float64, device=X.device)
        t_addr, wn_addr = T.data_ptr(), wn.data_ptr()
    else:
        T = np.empty((n, kmax), dtype=np.float64, order="F")
        wn = np.empty(n, dtype=np.float64)
        t_addr, wn_addr = T.ctypes.data, wn.ctypes.data
    P = np.zeros((p, kmax), order="F"); R = np.zeros((p, kmax), order="F"); W = np.zeros((p, kmax), order="F")
    Cm = np.zeros((q, kmax), order="F"); TT = np.zeros(kmax)
    xm = np.empty(p); xs = np.empty(p); ym = np.empty(q); ys = np.empty(q)
    bf16 = dev and X.dtype == torch.bfloat16
    if bf16 and Y.dtype != torch.bfloat16:
        raise TypeError("bf16 storage mode needs both X and Y as bfloat16 tensors")
    xa, ldx = _addr_ld(X, allow_bf16=True)
    ya, ldy = _addr_ld(Y, allow_bf16=True)
    desc = PlsDesc(n=n, p=p, q=q, nlv=int(nlv), scal=int(bool(scal)), dtype=_lib.BF16 if bf16 else _lib.F64,
                   loc=_lib.LOC_DEVICE if dev else _lib.LOC_HOST, inplace=int(inplace), reserved=int(variant))
    got = C.c_int32(0)
    if dev:
        torch.cuda.current_stream(X.device).synchronize()  # inputs produced on other streams are complete
    niter = None
    if ext_scales is not None:   # jch_plskern_fit_scaled: caller-supplied column divisors after weights
        xdiv = np.ascontiguousarray(ext_scales[0], dtype=np.float64)
        ydiv = None if ext_scales[1] is None else np.ascontiguousarray(ext_scales[1], dtype=np.float64)
        if xdiv.shape[0] != p or (ydiv is not None and ydiv.shape[0] != q):
            raise ValueError("DimensionMismatch: scale vectors")
        st = _lib.load().jch_plskern_fit_scaled(ctx._h, C.byref(desc), xa, ldx, ya, ldy, w_addr, _np(xdiv), _np(ydiv), t_addr, _np(P),
                                                _np(R), _np(W), _np(Cm), _np(TT), _np(xm), _np(xs), _np(ym), _np(ys), wn_addr,
                                                C.byref(got))
    elif wold is not None:   # jch_plswold_fit: tol, maxit after weights; niter before nlv_out
        niter = np.zeros(kmax)
        st = _lib.load().jch_plswold_fit(ctx._h, C.byref(desc), xa, ldx, ya, ldy, w_addr, float(wold[0]), int(wold[1]), t_addr,
                                         _np(P), _np(R), _np(W), _np(Cm), _np(TT), _np(xm), _np(xs), _np(ym), _np(ys), wn_addr,
                                         _np(niter), C.byref(got))
    else:
        st = getattr(_lib.load(), entry)(ctx._h, C.byref(desc), xa, ldx, ya, ldy, w_addr, t_addr, _np(P), _np(R), _np(W),
                                          _np(Cm), _np(TT), _np(xm), _np(xs), _np(ym), _np(ys), wn_addr, C.byref(got))
    ctx.check(st)
    k = got.value
    return Plsr(T[:, :k], P[:, :k], R[:, :k], W[:, :k], Cm[:, :k], TT[:k], xm, xs, ym, ys, wn,
                None if niter is None else niter[:k])


REUSE_XCOPY = 8                # include/jchemo_hip.h JCH_REUSE_XCOPY


def plskern(X, Y, weights=None, *, nlv: int, scal: bool = False, ctx: Optional[Context] = None, variant: int = 0, reuse_x: bool = False) -> Plsr:
    """`plskern(X, Y, weights = ones(n); nlv, scal = false)` — src/plskern.jl:106-110.  X, Y untouched
    (the reference copies them first; here the library simply never writes them).
    `variant=1` (not in the reference) opts into kernel algorithm #2: X'DX once, no pass over X in the LV loop.
    `reuse_x=True`: the caller's promise that X is the array (pointer AND contents) the previous fit on this ctx was given — a
    cross-validation fold, a grid combination; the library then takes X'DY from the row-major copy that fit left behind instead
    of staging and transposing X again (JCH_REUSE_XCOPY; ignored wherever it does not apply)."""
    X = ensure_mat(X); Y = ensure_mat(Y)
    try:
        _addr_ld(X, allow_bf16=True); _addr_ld(Y, allow_bf16=True)   # bf16 tensors: storage mode of BASELINE configs[2]
    except (ValueError, TypeError):
        X, Y = _as_colmajor_copy(X), _as_colmajor_copy(Y)   # layout/dtype conversion only
    return _fit("jch_plskern_fit", X, Y, weights, nlv, scal, False, ctx, variant | (REUSE_XCOPY if reuse_x else 0))


def plskern_(X, Y, weights=None, *, nlv: int, scal: bool = False, ctx: Optional[Context] = None) -> Plsr:
    """`plskern!(X::Matrix, Y::Matrix, ...)` — src/plskern.jl:112-178: X and Y are overwritten with their
    centred/scaled versions, so they must be column-major float64 matrices owned by the caller."""
    return _fit("jch_plskern_fit", _as_colmajor_view(X), _as_colmajor_view(Y), weights, nlv, scal, True, ctx)


NIPALS_ONE_PASS = 4            # include/jchemo_hip.h JCH_NIPALS_ONE_PASS


def plsnipals(X, Y, weights=None, *, nlv: int, scal: bool = False, ctx: Optional[Context] = None, one_pass: bool = False, reuse_x: bool = False) -> Plsr:
    """`plsnipals` — src/plsnipals.jl:31-35.
    `one_pass=True` (not in the reference, never the default) opts into ONE pass over X per LV: the next X'DY follows from the exact
    identity K_{a+1} = K_a - zp_raw c_raw' / tt instead of being recomputed from the deflated matrices (src/plsnipals.jl:71); same
    results up to rounding (q <= 16, p <= 2048)."""
    X = ensure_mat(X); Y = ensure_mat(Y)
    try:
        _addr_ld(X, allow_bf16=True); _addr_ld(Y, allow_bf16=True)
    except (ValueError, TypeError):
        X, Y = _as_colmajor_copy(X), _as_colmajor_copy(Y)
    return _fit("jch_plsnipals_fit", X, Y, weights, nlv, scal, False, ctx, (NIPALS_ONE_PASS if one_pass else 0) | (REUSE_XCOPY if reuse_x else 0))


def plsnipals_(X, Y, weights=None, *, nlv: int, scal: bool = False, ctx: Optional[Context] = None) -> Plsr:
    """`plsnipals!` — src/plsnipals.jl:37-97: X, Y end up centred/scaled AND deflated."""
    return _fit("jch_plsnipals_fit", _as_colmajor_view(X), _as_colmajor_view(Y), weights, nlv, scal, True, ctx)


def _copy_fit(entry, X, Y, weights, nlv, scal, ctx, wold=None, variant=0):
    X = ensure_mat(X); Y = ensure_mat(Y)
    try:
        _addr_ld(X, allow_bf16=True); _addr_ld(Y, allow_bf16=True)   # (bf16-stored device tensors: widened exactly by the library where a fit has no bf16 kernels)
    except (ValueError, TypeError):
        X, Y = _as_colmajor_copy(X), _as_colmajor_copy(Y)
    return _fit(entry, X, Y, weights, nlv, scal, False, ctx, wold=wold, variant=variant)


def plssimp(X, Y, weights=None, *, nlv: int, scal: bool = False, ctx: Optional[Context] = None, reuse_x: bool = False) -> Plsr:
    """`plssimp` — src/plssimp.jl:22-26 (SIMPLS, scores not normed; `W` is returned equal to `R`, :85-87).  `reuse_x`: see plskern."""
    return _copy_fit("jch_plssimp_fit", X, Y, weights, nlv, scal, ctx, variant=REUSE_XCOPY if reuse_x else 0)


def plssimp_(X, Y, weights=None, *, nlv: int, scal: bool = False, ctx: Optional[Context] = None) -> Plsr:
    """`plssimp!` — src/plssimp.jl:28-88: X, Y overwritten with their centred/scaled versions."""
    return _fit("jch_plssimp_fit", _as_colmajor_view(X), _as_colmajor_view(Y), weights, nlv, scal, True, ctx)


def plsrosa(X, Y, weights=None, *, nlv: int, scal: bool = False, ctx: Optional[Context] = None, reuse_x: bool = False) -> Plsr:
    """`plsrosa` — src/plsrosa.jl:26-30.  `reuse_x`: see plskern."""
    return _copy_fit("jch_plsrosa_fit", X, Y, weights, nlv, scal, ctx, variant=REUSE_XCOPY if reuse_x else 0)


def plsrosa_(X, Y, weights=None, *, nlv: int, scal: bool = False, ctx: Optional[Context] = None) -> Plsr:
    """`plsrosa!` — src/plsrosa.jl:32-96: X ends up centred/scaled, Y centred/scaled AND deflated (:87)."""
    return _fit("jch_plsrosa_fit", _as_colmajor_view(X), _as_colmajor_view(Y), weights, nlv, scal, True, ctx)


_SQRT_EPS = float(np.sqrt(np.finfo(np.float64).eps))


WOLD_REF_ZERO_WEIGHT_NAN = 2   # include/jchemo_hip.h JCH_WOLD_REF_ZERO_WEIGHT_NAN


def plswold(X, Y, weights=None, *, nlv: int, tol: float = _SQRT_EPS, maxit: int = 200, scal: bool = False,
            zero_weight_nan: bool = False, ctx: Optional[Context] = None, one_pass: bool = False, reuse_x: bool = False) -> Plsr:
    """`plswold` — src/plswold.jl:30-34; `niter` (inner passes per LV) as :93.  `zero_weight_nan = True` reproduces the
    reference's NaN scores for rows whose weight is 0 (:107); the default keeps them finite (t_i = x_i' r), which is what
    lets a cross-validation fold be ONE weighted fit (gridcvlv)."""
    return _copy_fit("jch_plswold_fit", X, Y, weights, nlv, scal, ctx, wold=(tol, maxit),
                     variant=(WOLD_REF_ZERO_WEIGHT_NAN if zero_weight_nan else 0) | (NIPALS_ONE_PASS if one_pass else 0) | (REUSE_XCOPY if reuse_x else 0))   # one_pass: see plsnipals


def plswold_(X, Y, weights=None, *, nlv: int, tol: float = _SQRT_EPS, maxit: int = 200, scal: bool = False,
             zero_weight_nan: bool = False, ctx: Optional[Context] = None) -> Plsr:
    """`plswold!` — src/plswold.jl:36-111: X, Y end up centred/scaled, carrying the row metric sqrt(w) and deflated."""
    return _fit("jch_plswold_fit", _as_colmajor_view(X), _as_colmajor_view(Y), weights, nlv, scal, True, ctx, wold=(tol, maxit),
                variant=WOLD_REF_ZERO_WEIGHT_NAN if zero_weight_nan else 0)


# ---------------------------------------------------------------------------------- accessors
def _nlv_arg(fm: Plsr, nlv) -> int:
    a = fm.P.shape[1]
    return a if nlv is None else min(int(nlv), a)


def _affine(X, shift, scale, B, bias, ctx):
    X = ensure_mat(X)
    try:
        _addr_ld(X)
    except (ValueError, TypeError):
        X = _as_colmajor_copy(X)
    dev = _is_torch(X)
    m, p = X.shape
    k = B.shape[1]
    B = np.asfortranarray(B, dtype=np.float64)
    if B.shape[0] != p:
        raise ValueError(f"DimensionMismatch: X has {p} columns, the model has {B.shape[0]}")
    ctx = ctx or default_context((X.device.index or 0) if dev else 0)
    if dev:
        out = colmajor_empty(m, k, X.device)
        torch.cuda.current_stream(X.device).synchronize()
        oa = out.data_ptr()
    else:
        out = np.empty((m, k), dtype=np.float64, order="F")
        oa = out.ctypes.data
    xa, ldx = _addr_ld(X)
    sh = None if shift is None else np.ascontiguousarray(shift, dtype=np.float64)
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float64)
    bi = None if bias is None else np.ascontiguousarray(bias, dtype=np.float64).reshape(-1)
    ctx.check(_lib.load().jch_affine_gemm(ctx._h, _lib.LOC_DEVICE if dev else _lib.LOC_HOST, xa, m, p, ldx, _np(sh), _np(sc),
                                          B.ctypes.data, k, _np(bi), oa, max(m, 1)))
    return out


def _x_out(X, k, ctx):
    """Common plumbing of the accessors: column-major X, an m x k output next to it, the ctx."""
    X = ensure_mat(X)
    try:
        _addr_ld(X)
    except (ValueError, TypeError):
        X = _as_colmajor_copy(X)
    dev = _is_torch(X)
    m = X.shape[0]
    ctx = ctx or default_context((X.device.index or 0) if dev else 0)
    if dev:
        out = colmajor_empty(m, k, X.device)
        torch.cuda.current_stream(X.device).synchronize()
        oa = out.data_ptr()
    else:
        out = np.empty((m, k), dtype=np.float64, order="F")
        oa = out.ctypes.data
    return X, out, oa, ctx, (_lib.LOC_DEVICE if dev else _lib.LOC_HOST)


def _model_vec(v):
    return np.ascontiguousarray(v, dtype=np.float64)


def transform(fm: Plsr, X, *, nlv: Optional[int] = None, ctx: Optional[Context] = None):
    """src/plskern.jl:187-195: `cscale(X, xmeans, xscales) * R[:, 1:nlv]` (nlv clamped to the model's) — jch_transform."""
    k = _nlv_arg(fm, nlv)
    if k < 1:
        raise ValueError("transform needs nlv >= 1")
    X, out, oa, ctx, loc = _x_out(X, k, ctx)
    m, p = X.shape
    if fm.R.shape[0] != p:
        raise ValueError(f"DimensionMismatch: X has {p} columns, the model has {fm.R.shape[0]}")
    R = np.asfortranarray(fm.R[:, :k], dtype=np.float64)
    xm, xs = _model_vec(fm.xmeans), _model_vec(fm.xscales)
    xa, ldx = _addr_ld(X)
    ctx.check(_lib.load().jch_transform(ctx._h, loc, xa, m, p, ldx, _np(xm), _np(xs), R.ctypes.data, k, oa, max(m, 1)))
    return out


def _predict_range(fm: Plsr, X, lo: int, hi: int, ctx):
    """jch_predict: predictions for every nlv in lo..hi, one pass over X; m x (q * (hi - lo + 1))."""
    q = fm.C.shape[0]
    X, out, oa, ctx, loc = _x_out(X, q * (hi - lo + 1), ctx)
    m, p = X.shape
    if fm.R.shape[0] != p:
        raise ValueError(f"DimensionMismatch: X has {p} columns, the model has {fm.R.shape[0]}")
    R = np.asfortranarray(fm.R, dtype=np.float64); Cm = np.asfortranarray(fm.C, dtype=np.float64)
    xm, xs, ym, ys = (_model_vec(v) for v in (fm.xmeans, fm.xscales, fm.ymeans, fm.yscales))
    xa, ldx = _addr_ld(X)
    ctx.check(_lib.load().jch_predict(ctx._h, loc, xa, m, p, ldx, _np(xm), _np(xs), _np(ym), _np(ys), R.ctypes.data,
                                      Cm.ctypes.data, q, lo, hi, oa, max(m, 1)))
    return out


def coef(fm: Plsr, *, nlv: Optional[int] = None):
    """src/plskern.jl:207-217 — (B p x q, int 1 x q); nlv = 0 gives B = 0.  p x q host glue."""
    k = _nlv_arg(fm, nlv)
    beta = fm.C[:, :k].T
    B = (fm.R[:, :k] / fm.xscales[:, None]) @ beta * fm.yscales[None, :]
    intercept = fm.ymeans[None, :] - fm.xmeans[None, :] @ B
    return B, intercept


def _pred_matrix(fm: "Plsr", X, rng, ctx):
    """m x (len(rng) * q) matrix [pred_{rng[0]} | pred_{rng[1]} | ...] (level-major columns)."""
    q = fm.C.shape[0]
    # several nlv: the library takes ONE pass over X for the scores (m x max(nlv)); every prediction block is then a running sum of
    # score x loading terms, pred_a = ymeans + sum_{l <= a} T_l (C_l .* yscales)' (k_predict_prefix, csrc/gemm.hip)
    a = fm.P.shape[1]
    if rng[-1] <= a:
        return _predict_range(fm, X, rng[0], rng[-1], ctx)   # rng is contiguous (src/plskern.jl:228)
    # levels beyond the fitted LVs repeat the last one (the reference clamps, :228)
    lo, hi = min(rng[0], a), a
    out = _predict_range(fm, X, lo, hi, ctx)
    cols = np.concatenate([np.arange(q) + (min(k, a) - lo) * q for k in rng])
    return out[:, torch.as_tensor(cols, device=out.device)] if _is_torch(out) else out[:, cols]


def predict(fm: Plsr, X, *, nlv: Union[None, int, Sequence[int]] = None, ctx: Optional[Context] = None, rank: Optional[int] = None,
            world: Optional[int] = None):
    """src/plskern.jl:226-238: a collection of nlv becomes the contiguous range max(0,min):min(a,max); one
    value -> matrix, several -> list of matrices.  All values are computed in ONE pass over X.
    `rank` / `world` (kNN-LWPLSR only): split the queries over `world` replicas, see lwplsr_predict."""
    if isinstance(fm, Lwplsr):
        return lwplsr_predict(fm, X, nlv=nlv, ctx=ctx, rank=rank, world=world)
    if isinstance(fm, Plsrda):
        return plsrda_predict(fm, X, nlv=nlv, ctx=ctx)
    if isinstance(fm, Plslda):
        return plslda_predict(fm, X, nlv=nlv, ctx=ctx)
    if isinstance(fm, Mbplsr):
        return mbplsr_predict(fm, X, nlv=nlv, ctx=ctx)
    a = fm.P.shape[1]
    if nlv is None:
        rng = [a]
    else:
        vals = np.atleast_1d(np.asarray(nlv))
        rng = list(range(max(0, int(vals.min())), min(a, int(vals.max())) + 1))
    q = fm.C.shape[0]
    out = _pred_matrix(fm, X, rng, ctx)
    preds = [out[:, i * q:(i + 1) * q] for i in range(len(rng))]
    return preds[0] if len(preds) == 1 else preds


def summary(fm: Plsr, X, *, ctx: Optional[Context] = None):
    """src/plskern.jl:246-260 — explained X-variance table as a dict of columns (nlv, var, pvar, cumpvar)."""
    X = ensure_mat(X)
    try:
        _addr_ld(X)
    except (ValueError, TypeError):
        X = _as_colmajor_copy(X)
    dev = _is_torch(X)
    n = X.shape[0]
    nlv = fm.P.shape[1]
    ctx = ctx or default_context((X.device.index or 0) if dev else 0)
    w = fm.weights
    if dev and not _is_torch(w):
        w = torch.as_tensor(np.asarray(w), device=X.device)
    if not dev and _is_torch(w):
        w = w.cpu().numpy()
    ss = C.c_double(0.0)
    xa, ldx = _addr_ld(X)
    wa = w.data_ptr() if dev else np.ascontiguousarray(w).ctypes.data
    if dev:
        torch.cuda.current_stream(X.device).synchronize()
    ctx.check(_lib.load().jch_weighted_ss(ctx._h, _lib.LOC_DEVICE if dev else _lib.LOC_HOST, xa, n, X.shape[1], ldx, wa,
                                          _np(np.ascontiguousarray(fm.xmeans)), _np(np.ascontiguousarray(fm.xscales)),
                                          C.byref(ss)))
    tt_adj = np.sum(fm.P ** 2, axis=0) * fm.TT
    pvar = tt_adj / ss.value
    return dict(nlv=np.arange(1, nlv + 1), var=tt_adj / n, pvar=pvar, cumpvar=np.cumsum(pvar))


def xfit(fm: Plsr, X, *, nlv: Optional[int] = None, ctx: Optional[Context] = None):
    """`xfit(object, X; nlv)` — src/xfit.jl:37-56: X reconstructed from nlv LVs in the original scale,
    (cscale(X) R_k) (P_k' diag(xscales)) + xmeans: the scores pass over X, then a GEMM on the m x nlv scores."""
    k = _nlv_arg(fm, nlv)
    p = fm.P.shape[0]
    if k == 0:   # the column means (src/xfit.jl:41-45); an m x p broadcast through the same device primitive
        return _affine(X, None, None, np.zeros((p, p)), fm.xmeans, ctx)
    Tq = transform(fm, X, nlv=k, ctx=ctx)
    return _affine(Tq, None, None, fm.P[:, :k].T * fm.xscales[None, :], fm.xmeans, ctx)


def xresid(fm: Plsr, X, *, nlv: Optional[int] = None, ctx: Optional[Context] = None):
    """`xresid(object, X; nlv)` — src/xfit.jl:86-93: E = X - xfit(X) = cscale(X) (I - R_k P_k') diag(xscales),
    one p x p device GEMM on X (no n x p temporary on the host)."""
    k = _nlv_arg(fm, nlv)
    p = fm.P.shape[0]
    M = np.eye(p) - fm.R[:, :k] @ fm.P[:, :k].T
    return _affine(X, fm.xmeans, fm.xscales, M * fm.xscales[None, :], None, ctx)


def vip(fm: Plsr, Y=None, *, nlv: Optional[int] = None, ctx: Optional[Context] = None):
    """`vip(object; nlv)` / `vip(object, Y; nlv)` — src/vip.jl:62-107.  Without Y everything follows from W, C and
    TT = t'Dt (p x nlv host glue); with Y the redundancies rd(Y, T, weights) (src/angles.jl:97-105) come from ONE
    weighted covariance of [Y | T] on the device (jch_weighted_cov)."""
    a = fm.P.shape[1]
    p = fm.W.shape[0]
    k = a if nlv is None else min(int(nlv), a)
    W2 = fm.W[:, :k] ** 2
    if Y is None:
        sst = np.sum(fm.C[:, :k] ** 2, axis=0) * fm.TT[:k]       # tr(C_a C_a') * t_a'D t_a  (src/vip.jl:76-82)
        A = (sst[None, :] * W2).sum(axis=1)
        return dict(imp=np.sqrt(A / (sst.sum() / p)), W2=W2, sst=sst)
    Y = ensure_mat(Y)
    q = Y.shape[1]
    T = fm.T[:, :k]
    dev = _is_torch(T)
    if dev:
        Yd = Y if _is_torch(Y) else torch.as_tensor(np.asarray(Y, dtype=np.float64), device=T.device)
        A_ = colmajor_empty(T.shape[0], q + k, T.device)
        A_[:, :q] = Yd; A_[:, q:] = T
        w = fm.weights
    else:
        A_ = np.asfortranarray(np.hstack([np.asarray(Y, dtype=np.float64), T]))   # layout only
        w = np.ascontiguousarray(fm.weights)
    ctx = ctx or default_context((T.device.index or 0) if dev else 0)
    S = np.empty((q + k, q + k), order="F")
    aa, lda = _addr_ld(A_)
    if dev:
        torch.cuda.current_stream(T.device).synchronize()
    ctx.check(_lib.load().jch_weighted_cov(ctx._h, _lib.LOC_DEVICE if dev else _lib.LOC_HOST, aa, A_.shape[0], q + k, lda,
                                           w.data_ptr() if dev else w.ctypes.data, S.ctypes.data, None))
    sd = np.sqrt(np.diag(S))
    cor = S[:q, q:] / sd[:q, None] / sd[None, q:]
    rdd = (cor ** 2).sum(axis=0, keepdims=True) / q
    A = (rdd * W2).sum(axis=1)
    return dict(imp=np.sqrt(A / (rdd.sum() / p)), W2=W2, rdd=rdd)


# ---------------------------------------------------------------------------------- kNN-LWPLSR (src/lwplsr.jl)
@dataclass
class Lwplsr:
    """src/lwplsr.jl:1-12 — same fields."""
    X: object
    Y: object
    fm: Optional[Plsr]
    metric: str
    h: float
    k: int
    nlv: int
    tol: float
    scal: bool
    verbose: bool = False


@dataclass
class LwplsrPred:
    """The NamedTuple returned by `predict(::Lwplsr, X)` (src/lwplsr.jl:165): pred is a matrix for one nlv, else a
    list of matrices; listnn holds 0-based row indices (Julia: 1-based)."""
    pred: object
    listnn: np.ndarray
    listd: np.ndarray
    listw: np.ndarray


def lwplsr(X, Y, *, nlvdis: int, metric: str, h: float, k: int, nlv: int, tol: float = 1e-4, scal: bool = False,
           verbose: bool = False, ctx: Optional[Context] = None) -> Lwplsr:
    """`lwplsr(X, Y; nlvdis, metric, h, k, nlv, tol = 1e-4, scal = false)` — src/lwplsr.jl:114-126: stores the data and
    (nlvdis > 0) a global plskern fit whose scores define the neighbourhood space."""
    X = ensure_mat(X); Y = ensure_mat(Y)
    if metric not in ("eucl", "mahal"):
        raise ValueError(f"metric must be 'eucl' or 'mahal', got {metric!r}")
    fm = None if nlvdis == 0 else plskern(X, Y, nlv=nlvdis, scal=scal, ctx=ctx)
    return Lwplsr(X, Y, fm, metric, h, k, nlv, tol, scal, verbose)


def _cov_uncorrected(A, ctx):
    A = ensure_mat(A)
    dev = _is_torch(A)
    n, d = A.shape
    S = np.empty((d, d), order="F")
    aa, lda = _addr_ld(A)
    if dev:
        torch.cuda.current_stream(A.device).synchronize()
    ctx.check(_lib.load().jch_weighted_cov(ctx._h, _lib.LOC_DEVICE if dev else _lib.LOC_HOST, aa, n, d, lda, None, S.ctypes.data, None))
    return S


def _knn_train_space(obj: Lwplsr, ctx):
    """The space neighbours are searched in (src/lwplsr.jl:139-151 + the whitening of src/getknn.jl:37-49): returns the training
    coordinates Zt (n x dd) and the map that takes a query block to the same coordinates.  Model-constant: computed once per
    `Lwplsr` object and cached with the device handle (`_lwplsr_prepared`).  Third value: the same map as a list of affine
    stages (shift, scale, B) for jch_lwplsr_add_query_map — [] when the queries are searched in their own coordinates."""
    stages = []
    if obj.fm is None:
        Zt = obj.X
        try:
            _addr_ld(Zt)
        except (ValueError, TypeError):
            Zt = _as_colmajor_copy(Zt)
        D = None
        if obj.scal:   # colstd(object.X) (unweighted) on both sides
            xs = plskern(obj.X, obj.Y, nlv=1, scal=True, ctx=ctx).xscales
            D = np.diag(1.0 / xs)
            Zt = _affine(Zt, None, None, D, None, ctx)
            stages.append((None, None, D))
        qmap = (lambda Xq: Xq) if D is None else (lambda Xq: _affine(Xq, None, None, D, None, ctx))
    else:
        fm = obj.fm          # (the closures below must not capture `obj`: they live in the module-level handle table, and a strong
        Zt = fm.T            # reference from there would keep the model — and its device handle — alive for ever)
        qmap = lambda Xq: transform(fm, Xq, ctx=ctx)
        stages.append((_model_vec(fm.xmeans), _model_vec(fm.xscales), np.asfortranarray(fm.R[:, :_nlv_arg(fm, None)], dtype=np.float64)))
    if obj.metric == "mahal":
        d = Zt.shape[1]
        S = _cov_uncorrected(Zt, ctx)
        if d == 1:
            Uinv = np.array([[1.0 / np.sqrt(S[0, 0])]])
        else:
            try:
                Uinv = np.linalg.inv(np.linalg.cholesky(S).T)        # inv(cholesky(S).U)
            except np.linalg.LinAlgError:
                Uinv = np.diag(1.0 / np.diag(S))                     # sic, src/getknn.jl:43
        Zt = _affine(Zt, None, None, Uinv, None, ctx)
        inner = qmap
        qmap = lambda Xq: _affine(inner(Xq), None, None, Uinv, None, ctx)
        stages.append((None, None, Uinv))
    return Zt, qmap, stages


def _knn_space(obj: Lwplsr, Xq, ctx):
    """(training coordinates, query coordinates) — see _knn_train_space."""
    Zt, qmap, _ = _knn_train_space(obj, ctx)
    return Zt, qmap(Xq)


def _release_lwplsr_handle(handle):
    try:
        _lib.load().jch_lwplsr_release(None, handle)
    except Exception:  # pragma: no cover  (interpreter shutdown)
        pass


# Prepared device handles live OUTSIDE the `Lwplsr` instances (keyed by id(obj), dropped by a per-object finalizer): the
# dataclass stays plain data, so copy / deepcopy / pickle of a predicted-from model work and a copy never shares — or outlives —
# the original's handle.
_LWPLSR_PREP = {}


def _drop_lwplsr_prep(oid):
    st = _LWPLSR_PREP.pop(oid, None)
    if st is not None:
        _release_lwplsr_handle(st["handle"])


def _lwplsr_prepared(obj: Lwplsr, ctx, dev: bool, qk: int):
    """Device handle of the model-constant data of `obj` (include/jchemo_hip.h jch_lwplsr_prepare: row-major Xtrain, Ytrain, the
    whitened training scores) + the query map, built on the first `predict` and kept for the object's lifetime — the reference's
    `Lwplsr` is fitted once and predicted from many times (src/lwplsr.jl:1-12).  qk: responses handed to the batched kernel.
    The cache key covers everything the handle was built from: re-assigning `obj.X`, `obj.Y`, `obj.fm`, `obj.metric` or
    `obj.scal` after a predict rebuilds it (in-place edits of the arrays themselves are not detected)."""
    import weakref
    key = (id(ctx), ctx._h.value, dev, qk, id(obj.X), id(obj.Y), id(obj.fm), obj.metric, bool(obj.scal))
    oid = id(obj)
    st = _LWPLSR_PREP.get(oid)
    if st is not None and st["key"] == key:
        return st
    if st is not None:
        _drop_lwplsr_prep(oid)                                   # another ctx / residency / model data: drop the old handle
    else:
        weakref.finalize(obj, _drop_lwplsr_prep, oid)
    Xt, Yt = obj.X, obj.Y
    try:
        _addr_ld(Xt); _addr_ld(Yt)
    except (ValueError, TypeError):
        Xt, Yt = _as_colmajor_copy(Xt), _as_colmajor_copy(Yt)
    Zt, qmap, stages = _knn_train_space(obj, ctx)
    n, p = Xt.shape
    xa, ldx = _addr_ld(Xt); ya, ldy = _addr_ld(Yt); za, ldz = _addr_ld(Zt)
    if dev:
        torch.cuda.current_stream(Xt.device).synchronize()
    h = C.c_void_p()
    ctx.check(_lib.load().jch_lwplsr_prepare(ctx._h, _lib.LOC_DEVICE if dev else _lib.LOC_HOST, xa, n, p, ldx, ya, qk, ldy, za, ldz,
                                             Zt.shape[1], C.byref(h)))
    st = {"key": key, "handle": h, "qmap": qmap, "dd": Zt.shape[1], "device_map": False}
    _LWPLSR_PREP[oid] = st
    # the query map travels with the handle: the library then takes Xq alone (two jch_affine_gemm calls per predict, each with
    # an upload of its matrix and a stream synchronisation, become two launches on the ctx stream)
    if stages and os.environ.get("JCH_LW_DEVICE_QMAP", "1") != "0":
        for shift, scale, B in stages:
            B = np.asfortranarray(B, dtype=np.float64)
            sh = None if shift is None else np.ascontiguousarray(shift, dtype=np.float64)
            sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float64)
            ctx.check(_lib.load().jch_lwplsr_add_query_map(ctx._h, h, _np(sh), _np(sc), B.ctypes.data, B.shape[0], B.shape[1], None))
        st["device_map"] = True
    return st


def query_shard(m: int, rank: int, world: int):
    """Rows [lo, hi) of the m queries that replica `rank` of `world` predicts (the queries of
    `predict(::Lwplsr)` are independent — src/locwlv.jl:18 runs them under `Threads.@threads` —, so every GPU holds a
    replica of the training data and takes a contiguous slice of the queries; nothing crosses GPUs but the results)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return (m * rank) // world, (m * (rank + 1)) // world


def _merge_lwplsr_parts(parts) -> "LwplsrPred":
    """Re-assemble the per-replica results (in rank order) into the result of the unsplit call."""
    parts = [p_ for p_ in parts if p_ is not None]
    if not parts:    # m = 0: every replica's slice was empty
        return LwplsrPred(np.empty((0, 0)), np.empty((0, 0), dtype=np.int32), np.empty((0, 0)), np.empty((0, 0)))
    single = not isinstance(parts[0].pred, list)
    if single:
        pred = np.concatenate([p_.pred for p_ in parts], axis=0)
    else:
        pred = [np.concatenate([p_.pred[a] for p_ in parts], axis=0) for a in range(len(parts[0].pred))]
    return LwplsrPred(pred, np.concatenate([p_.listnn for p_ in parts], axis=0), np.concatenate([p_.listd for p_ in parts], axis=0),
                      np.concatenate([p_.listw for p_ in parts], axis=0))


def lwplsr_predict(obj: Lwplsr, X, *, nlv=None, ctx: Optional[Context] = None, rank: Optional[int] = None,
                   world: Optional[int] = None, gather=None, lists: bool = True) -> LwplsrPred:
    """`predict(object::Lwplsr, X; nlv)` — src/lwplsr.jl:134-166.

    `lists = False` (not in the reference, whose result always carries them): predictions only — the neighbour lists, distances
    and weights (3.2 MB at cfg5) stay on the device and `listnn / listd / listw` come back as None.

    Multi-GPU ("replicas", one process per GPU holding the whole training set): pass this process' `rank` and the
    `world` size; the call predicts the slice `query_shard(m, rank, world)` on its GPU and returns the FULL result on
    every rank, gathered in rank order with `gather` (default: torch.distributed.all_gather_object on the default
    process group; any callable part -> list of parts will do, e.g. an MPI allgather)."""
    if world is not None and world > 1:
        if rank is None:
            raise ValueError("lwplsr predict: `world` > 1 needs this process' `rank`")
        X = ensure_mat(X)
        lo, hi = query_shard(X.shape[0], int(rank), int(world))
        part = None
        if hi > lo:
            part = lwplsr_predict(obj, X[lo:hi], nlv=nlv, ctx=ctx)   # (replica split: always with the lists, the merge expects them)
            if _is_torch(part.pred) or (isinstance(part.pred, list) and part.pred and _is_torch(part.pred[0])):  # pragma: no cover
                part = LwplsrPred([t.cpu().numpy() for t in part.pred] if isinstance(part.pred, list) else part.pred.cpu().numpy(),
                                  part.listnn, part.listd, part.listw)
        if gather is None:
            import torch.distributed as dist

            def gather(x):
                out = [None] * dist.get_world_size()
                dist.all_gather_object(out, x)
                return out
        return _merge_lwplsr_parts(gather(part))
    X = ensure_mat(X)
    try:
        _addr_ld(X)
    except (ValueError, TypeError):
        X = _as_colmajor_copy(X)
    Xt, Yt = obj.X, obj.Y
    dev = _is_torch(X)
    if dev != _is_torch(Xt):
        raise TypeError("training data and queries must both be host arrays or both device tensors")
    ctx = ctx or default_context((X.device.index or 0) if dev else 0)
    a = obj.nlv
    if nlv is None:
        lo = hi = a
    else:
        vals = np.atleast_1d(np.asarray(nlv))
        lo, hi = max(int(vals.min()), 0), min(int(vals.max()), a)
    n, p = Xt.shape
    m, q = X.shape[0], Yt.shape[1]
    hi = min(hi, p)                                            # src/locwlv.jl:14
    le = hi - lo + 1
    k = min(obj.k, n)
    # model-constant device data (row-major Xtrain, Ytrain, whitened training scores): prepared once per object.  No shape limits:
    # outside the batched kernels' envelope (k > 768; local fits with p > 2048, q > 16, nlv > 48) the library runs its generic
    # paths — an exact selection per query, then one jch_plskern_fit + jch_predict per query, the reference's own schedule
    # (src/locwlv.jl:18-39; csrc/lwplsr_generic.hip)
    st = _lwplsr_prepared(obj, ctx, dev, q)
    pred = np.empty((m, le, q))
    if lists:
        ind = np.empty((m, k), dtype=np.int32); dist = np.empty((m, k)); w = np.empty((m, k))
    else:
        ind = dist = w = None
    if st["device_map"]:
        qa, ldq = None, 0                                          # the handle maps the queries itself
    else:
        Zq = st["qmap"](X)
        qa, ldq = _addr_ld(Zq)
    xqa, ldxq = _addr_ld(X)
    if dev:
        torch.cuda.current_stream(X.device).synchronize()
    ctx.check(_lib.load().jch_lwplsr_predict_prepared(ctx._h, st["handle"], _lib.LOC_DEVICE if dev else _lib.LOC_HOST, qa, ldq, xqa, m, ldxq, k,
                                                      float(obj.h), float(obj.tol), int(obj.scal), lo, hi, pred.ctypes.data, _np(ind),
                                                      _np(dist), _np(w)))
    if getattr(obj, "verbose", False):                            # src/locwlv.jl:19,40 (`print(i, " ")` per query, then a newline)
        print("".join(f"{i} " for i in range(1, m + 1)))
    preds = [pred[:, i, :].copy() for i in range(le)]
    return LwplsrPred(preds[0] if le == 1 else preds, ind, dist, w)


# ---------------------------------------------------------------------------------- scores / grid search (§8f rank 1)
def _score_sums(pred, Y, mask, ctx):
    """Device statistics of jch_score_sums: returns array (levels, q, 6)."""
    pred = ensure_mat(pred); Y = ensure_mat(Y)
    try:
        _addr_ld(pred)
    except (ValueError, TypeError):
        pred = _as_colmajor_copy(pred)
    try:
        _addr_ld(Y)
    except (ValueError, TypeError):
        Y = _as_colmajor_copy(Y)
    dev = _is_torch(pred)
    if dev != _is_torch(Y):
        raise TypeError("pred and Y must both be host arrays or both device tensors")
    m, ncol = pred.shape
    q = Y.shape[1]
    if Y.shape[0] != m or ncol % q:
        raise ValueError("DimensionMismatch between predictions and Y")
    ctx = ctx or default_context((pred.device.index or 0) if dev else 0)
    pa, ldp = _addr_ld(pred); ya, ldy = _addr_ld(Y)
    ma = None
    if mask is not None:
        mask = mask if _is_torch(mask) == dev else (torch.as_tensor(np.asarray(mask, dtype=np.float64), device=pred.device) if dev else mask.cpu().numpy())
        mask = mask.to(torch.float64).contiguous() if dev else np.ascontiguousarray(mask, dtype=np.float64)
        ma = mask.data_ptr() if dev else mask.ctypes.data
    sums = np.empty((ncol, 6))
    if dev:
        torch.cuda.current_stream(pred.device).synchronize()
    ctx.check(_lib.load().jch_score_sums(ctx._h, _lib.LOC_DEVICE if dev else _lib.LOC_HOST, pa, m, ncol, ldp, ya, q, ldy, ma, sums.ctypes.data))
    return sums.reshape(ncol // q, q, 6)


def _score_sums_lv(T, fm, Y, mask, rng, ctx):
    """jch_score_sums_lv: the statistics of `_score_sums` for the predictions with nlv = rng[0]..rng[-1] (contiguous) straight from
    the rows' scores T (m x k) and the model's C / ymeans / yscales — the prediction matrix is never formed.  (levels, q, 6)."""
    T = ensure_mat(T); Y = ensure_mat(Y)
    dev = _is_torch(T)
    if dev != _is_torch(Y):
        Y = torch.as_tensor(np.asarray(Y), device=T.device) if dev else Y.cpu().numpy()
    try:
        _addr_ld(T)
    except (ValueError, TypeError):
        T = _as_colmajor_copy(T)
    try:
        _addr_ld(Y)
    except (ValueError, TypeError):
        Y = _as_colmajor_copy(Y)
    m, k = T.shape
    q = Y.shape[1]
    if Y.shape[0] != m or fm.C.shape[0] != q:
        raise ValueError("DimensionMismatch between the scores, Y and the model")
    k = min(k, fm.C.shape[1])
    lo, hi = int(rng[0]), int(rng[-1])
    ctx = ctx or default_context((T.device.index or 0) if dev else 0)
    ta, ldt = _addr_ld(T); ya, ldy = _addr_ld(Y)
    ma = None
    if mask is not None:
        mask = mask if _is_torch(mask) == dev else (torch.as_tensor(np.asarray(mask, dtype=np.float64), device=T.device) if dev else mask.cpu().numpy())
        mask = mask.to(torch.float64).contiguous() if dev else np.ascontiguousarray(mask, dtype=np.float64)
        ma = mask.data_ptr() if dev else mask.ctypes.data
    Cm = np.asfortranarray(fm.C[:, :k], dtype=np.float64)
    ym, ys = _model_vec(fm.ymeans), _model_vec(fm.yscales)
    sums = np.empty(((hi - lo + 1) * q, 6))
    if dev:
        torch.cuda.current_stream(T.device).synchronize()
    ctx.check(_lib.load().jch_score_sums_lv(ctx._h, _lib.LOC_DEVICE if dev else _lib.LOC_HOST, ta, m, k, ldt, Cm.ctypes.data, _np(ym), _np(ys),
                                            ya, q, ldy, ma, lo, hi, sums.ctypes.data))
    return sums.reshape(hi - lo + 1, q, 6)


def _score_from_sums(name: str, S: np.ndarray) -> np.ndarray:
    """S: (levels, q, 6) = {sum e, sum e^2, sum y e, sum y, sum y^2, count}.  Formulas: src/scores.jl."""
    se, see, sye, sy, syy, cnt = (S[..., i] for i in range(6))
    if name == "ssr":
        return see
    if name == "msep":
        return see / cnt
    if name == "rmsep":
        return np.sqrt(see / cnt)
    if name == "bias":
        return -se / cnt
    if name == "r2":
        return 1 - (see / cnt) / (syy / cnt - (sy / cnt) ** 2)
    if name == "cor2":
        sp, spp, spy = sy - se, syy - 2 * sye + see, syy - sye        # sums of pred, pred^2, pred*y
        cov = spy / cnt - (sp / cnt) * (sy / cnt)
        return cov ** 2 / ((spp / cnt - (sp / cnt) ** 2) * (syy / cnt - (sy / cnt) ** 2))
    raise ValueError(name)


def _make_score(name):
    def score(pred, Y, *, ctx: Optional[Context] = None):
        return _score_from_sums(name, _score_sums(pred, Y, None, ctx))[0].reshape(1, -1)
    score.__name__ = name
    score.__doc__ = f"`{name}(pred, Y)` — src/scores.jl; 1 x q, computed from device-side sums."
    score._jch_name = name
    return score


msep, rmsep, ssr, bias, r2, cor2 = (_make_score(nm) for nm in ("msep", "rmsep", "ssr", "bias", "r2", "cor2"))


def segmkf(n: int, K: int, *, rep: int = 1, seed=None):
    """src/segm.jl:44-57 — K-fold segments (0-based indices), `rep` replications."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(rep):
        perm = rng.permutation(n)
        out.append([np.sort(perm[j::K]) for j in range(K)])
    return out


def segmts(n: int, m: int, *, rep: int = 1, seed=None):
    """src/segm.jl:135-149 — one test segment of m rows per replication."""
    rng = np.random.default_rng(seed)
    return [[np.sort(rng.choice(n, size=m, replace=False))] for _ in range(rep)]


def _nlv_range(nlv, p):
    vals = np.atleast_1d(np.asarray(nlv))
    return list(range(max(0, int(vals.min())), min(p, int(vals.max())) + 1))


def mpar(**kwargs):
    """`mpar(; kwargs...)` — src/mpar.jl:15-24: every combination of the parameter values (first keyword fastest,
    `Base.product` order) as {name: list of ncomb values}."""
    import itertools
    names = list(kwargs)
    vals = [list(v) if isinstance(v, (list, tuple, np.ndarray, range)) else [v] for v in kwargs.values()]
    out = {nm: [] for nm in names}
    for c in itertools.product(*reversed(vals)):
        for nm, x in zip(names, reversed(c)):
            out[nm].append(x)
    return out


def _pars_rows(pars):
    """Element-wise combinations of the `pars` vectors (src/gridscore.jl:192-195); one empty combination for None."""
    if pars is None:
        return [dict()]
    names = list(pars)
    if "nlv" in names:
        raise ValueError("Argument `pars` must not contain `nlv` (src/gridscore.jl:163)")
    ncomb = len(pars[names[0]])
    return [{nm: pars[nm][i] for nm in names} for i in range(ncomb)]


def _grid_table(pars, rng, res):
    """Column layout of the reference's result DataFrame (src/gridscore.jl:204-216): combination-major rows."""
    rows = _pars_rows(pars)
    out = dict(nlv=list(rng) * len(rows), res=res)
    if pars is not None:
        for nm in pars:
            out[nm] = [r[nm] for r in rows for _ in rng]
    return out


def gridscorelv(Xtrain, Ytrain, X, Y, *, score, fun, nlv, pars=None, verbose: bool = False, ctx: Optional[Context] = None, **kwargs):
    """`gridscorelv(Xtrain, Ytrain, X, Y; score, fun, nlv, pars, verbose)` — src/gridscore.jl:167-221: per parameter combination
    one fit at max(nlv), predictions for the whole range in ONE pass over X, scores from device-side sums.
    `verbose` prints what the reference prints (:178,189,191,217).
    Returns dict(nlv=[...], <one list per pars key>, res=(ncomb * le_nlv, q)), rows combination-major."""
    rng = _nlv_range(nlv, ensure_mat(Xtrain).shape[1])
    name = getattr(score, "_jch_name", None)
    blocks = []
    if verbose:
        print("-- Nb. combinations = 0." if pars is None else f"-- Nb. combinations = {len(_pars_rows(pars))}")
    for kw in _pars_rows(pars):
        if verbose and pars is not None:
            print("".join(f"{k_} => {v_}" for k_, v_ in kw.items()))
        fm = fun(Xtrain, Ytrain, nlv=max(rng), ctx=ctx, **kwargs, **kw)
        if name is None or not isinstance(fm, Plsr):   # arbitrary score(pred, Y) or a non-Plsr model: what predict returns
            pred = predict(fm, X, nlv=rng, ctx=ctx)
            if isinstance(pred, LwplsrPred):
                pred = pred.pred
            pred = [pred] if len(rng) == 1 else pred
            blocks.append(np.vstack([np.asarray(score(pr, Y)).reshape(1, -1) for pr in pred]))
        else:
            # the scores of X once (one pass), then every level's statistics from running sums over the score columns
            kmax = min(max(rng), fm.P.shape[1])
            Tq = transform(fm, X, nlv=kmax, ctx=ctx) if kmax > 0 else np.zeros((ensure_mat(X).shape[0], 0))
            if kmax == 0:
                blocks.append(_score_from_sums(name, _score_sums(_pred_matrix(fm, X, rng, ctx), Y, None, ctx)))
            else:
                blocks.append(_score_from_sums(name, _score_sums_lv(Tq, fm, Y, None, rng, ctx)))
    if verbose:
        print("-- End.")
    return _grid_table(pars, rng, np.vstack(blocks))


_CV_FUNS = None


def gridcvlv(X, Y, *, segm, score, fun, nlv, pars=None, verbose: bool = False, ctx: Optional[Context] = None, **kwargs):
    """`gridcvlv(X, Y; segm, score, fun, nlv, pars)` — src/gridcv.jl:187-228.  The reference copies rmrow(X, s) for
    every segment; here X stays where it is and each fold is ONE weighted fit with weight 0 on the held-out rows
    (identical means / XtY / loadings), whose scores T on the held-out rows already are their transformed rows:
    predictions for every nlv are T[:, :a] * C[:, :a]' — running sums over the n x nlv scores, no second pass over X.
    Returns dict(nlv, <pars columns>, res (ncomb * le, q) mean over folds, res_rep (nrep, nsegm, ncomb * le, q))."""
    name = getattr(score, "_jch_name", None)
    if name is None or fun not in (plskern, plsnipals, plssimp, plsrosa, plswold):
        raise NotImplementedError("gridcvlv: score must be one of msep/rmsep/ssr/bias/r2/cor2 and fun a PLS fit of this module")
    X = ensure_mat(X); Y = ensure_mat(Y)
    try:
        _addr_ld(X); _addr_ld(Y)
    except (ValueError, TypeError):
        X, Y = _as_colmajor_copy(X), _as_colmajor_copy(Y)
    dev = _is_torch(X)
    n, p = X.shape
    q = Y.shape[1]
    rng = _nlv_range(nlv, p)
    combos = _pars_rows(pars)
    rep_out = []
    # every fit of this call sees the SAME X (only the weights change): from the second one on the library is told so and takes X'DY
    # from the row-major copy the previous fit left in its workspace (JCH_REUSE_XCOPY) instead of staging / transposing X again
    ctx = ctx or default_context((X.device.index or 0) if dev else 0)
    same_x = False
    for irep, listsegm in enumerate(segm):
        if verbose:                                     # src/gridcv.jl:197,202,226
            print(f"/ repl={irep + 1} ", end="")
        zres = []
        for jseg, s in enumerate(listsegm):
            if verbose:
                print(f"segm={jseg + 1} ", end="")
            s = np.asarray(s)
            if dev:   # the fold's 0/1 vectors are made where they are used: only the held-out row numbers cross the bus
                held = torch.zeros(n, dtype=torch.float64, device=X.device)
                held[torch.as_tensor(s, dtype=torch.int64).to(X.device)] = 1.0
                w = 1.0 - held
            else:
                held = np.zeros(n); held[s] = 1.0
                w = 1.0 - held
            kfit = min(max(rng), n - len(s))                                   # the reference clamps with the TRAINING rows
            blocks = []
            for kw in combos:
                fm = fun(X, Y, w, nlv=kfit, ctx=ctx, reuse_x=same_x, **kwargs, **kw)
                same_x = True
                # pred_a = ymeans + sum_{l < min(a, k)} T_l (C_l .* yscales)' on the held-out rows: running sums over the score
                # columns inside jch_score_sums_lv — the n x (levels q) prediction matrix is never formed
                blocks.append(_score_from_sums(name, _score_sums_lv(fm.T, fm, Y, held, rng, ctx)))
            zres.append(np.vstack(blocks))
        rep_out.append(np.stack(zres))
    if verbose:
        print("/ End.")
    res_rep = np.stack(rep_out)
    out = _grid_table(pars, rng, res_rep.mean(axis=(0, 1)))
    out["res_rep"] = res_rep
    return out


# ---------------------------------------------------------------------------------- PLSR-DA (§8f rank 4)
@dataclass
class Plsrda:
    """src/plsrda.jl: struct Plsrda (fm::Plsr, lev, ni)."""
    fm: Plsr
    lev: np.ndarray
    ni: np.ndarray


def dummy(y):
    """`dummy(y)` — src/utility.jl:509-519: (n x nlev 0/1 table, sorted levels).  Host-side label handling."""
    y = np.asarray(y.cpu() if _is_torch(y) else y).reshape(-1)
    lev = np.unique(y)
    return (y[:, None] == lev[None, :]).astype(np.float64), lev


def plsrda(X, y, weights=None, *, nlv: int, scal: bool = False, ctx: Optional[Context] = None) -> Plsrda:
    """`plsrda(X, y, weights; nlv, scal)` — src/plsrda.jl:71-77: plskern on the dummy table of the classes."""
    Yd, lev = dummy(y)
    yv = np.asarray(y.cpu() if _is_torch(y) else y).reshape(-1)
    ni = np.array([(yv == l).sum() for l in lev])
    X = ensure_mat(X)
    if _is_torch(X):
        Yt = colmajor_empty(Yd.shape[0], Yd.shape[1], X.device); Yt.copy_(torch.from_numpy(Yd)); Yd = Yt
    return Plsrda(plskern(X, Yd, weights, nlv=nlv, scal=scal, ctx=ctx), lev, ni)


def plsrda_predict(obj: Plsrda, X, *, nlv=None, ctx: Optional[Context] = None):
    """`predict(object::Plsrda, X; nlv)` — src/plsrda.jl:95-120: (pred, posterior); lists when several nlv."""
    post = predict(obj.fm, X, nlv=nlv, ctx=ctx)
    many = isinstance(post, list)
    posts = post if many else [post]
    preds = [obj.lev[np.argmax(z.cpu().numpy() if _is_torch(z) else z, axis=1)].reshape(-1, 1) for z in posts]
    return (preds, posts) if many else (preds[0], posts[0])


# ---------------------------------------------------------------------------------- PLS-LDA / PLS-QDA (§8f rank 4)
@dataclass
class Plslda:
    """src/plslda.jl:1-5 — struct Plslda (fm = (fm_pls, fm_da), lev, ni); also what plsqda returns.  fm_da[i-1] is the
    discriminant model on the first i scores: dict(mu (nlev, i), Uinv [nlev] (i, i), detS (nlev), wprior)."""
    fm_pls: Plsr
    fm_da: list
    lev: np.ndarray
    ni: np.ndarray


def _class_score_stats(fm: Plsr, yv, lev, ctx):
    """Per class: mean and uncorrected covariance of the scores T (src/matW.jl:27-57 on T), from ONE device pass over T
    per class (jch_weighted_cov with the class indicator as weights); a single-observation class gets the covariance of
    all rows."""
    T = fm.T
    dev = _is_torch(T)
    n, a = T.shape
    ctx = ctx or default_context((T.device.index or 0) if dev else 0)
    aa, lda_ = _addr_ld(T)
    if dev:
        torch.cuda.current_stream(T.device).synchronize()

    def cov_of(mask):
        S = np.empty((a, a), order="F"); mu = np.empty(a)
        if mask is None:
            wa = None
        elif dev:
            mask = torch.as_tensor(mask, device=T.device); wa = mask.data_ptr()
        else:
            mask = np.ascontiguousarray(mask); wa = mask.ctypes.data
        ctx.check(_lib.load().jch_weighted_cov(ctx._h, _lib.LOC_DEVICE if dev else _lib.LOC_HOST, aa, n, a, lda_, wa, S.ctypes.data,
                                               mu.ctypes.data))
        return mu, S

    ni = np.array([(yv == l).sum() for l in lev])
    all_cov = cov_of(None)[1] if np.any(ni == 1) else None
    mus, Wi = [], []
    for i, l in enumerate(lev):
        mu, S = cov_of((yv == l).astype(np.float64))
        mus.append(mu); Wi.append(all_cov if ni[i] == 1 else S)
    return np.stack(mus), Wi, ni


def _dmnorm(mu, S):
    """src/dmnorm.jl:112-128."""
    U = np.linalg.cholesky(S).T
    detS = float(np.prod(np.diag(U)) ** 2)
    return np.linalg.inv(U), (1e-20 if detS == 0 else detS)


def _plsda(X, y, weights, nlv, prior, scal, ctx, quadratic):
    if prior not in ("unif", "prop"):
        raise ValueError("prior must be 'unif' or 'prop'")
    Yd, lev = dummy(y)
    yv = np.asarray(y.cpu() if _is_torch(y) else y).reshape(-1)
    X = ensure_mat(X)
    if _is_torch(X):
        Yt = colmajor_empty(Yd.shape[0], Yd.shape[1], X.device); Yt.copy_(torch.from_numpy(Yd)); Yd = Yt
    fm = plskern(X, Yd, weights, nlv=nlv, scal=scal, ctx=ctx)
    mus, Wi, ni = _class_score_stats(fm, yv, lev, ctx)
    n, a = fm.T.shape
    nlev = len(lev)
    wprior = np.ones(nlev) / nlev if prior == "unif" else ni / ni.sum()
    fm_da = []
    for i in range(1, a + 1):
        if quadratic:   # src/qda.jl:68-75
            cs = [Wi[c][:i, :i] if ni[c] == 1 else Wi[c][:i, :i] * ni[c] / (ni[c] - 1) for c in range(nlev)]
        else:           # src/lda.jl:64-75: pooled, unbiased
            W = sum((ni[c] / n) * Wi[c][:i, :i] for c in range(nlev)) * n / (n - nlev)
            cs = [W] * nlev
        dm = [_dmnorm(mus[c, :i], cs[c]) for c in range(nlev)]
        fm_da.append(dict(mu=mus[:, :i].copy(), Uinv=[d[0] for d in dm], detS=np.array([d[1] for d in dm]), wprior=wprior))
    return Plslda(fm, fm_da, lev, ni)


def plslda(X, y, weights=None, *, nlv: int, prior: str = "unif", scal: bool = False, ctx: Optional[Context] = None) -> Plslda:
    """`plslda(X, y, weights; nlv, prior, scal)` — src/plslda.jl:76-88: plskern on the class dummy table, then one LDA
    per number of LVs on the scores.  The class statistics of the n x nlv scores come from the device."""
    return _plsda(X, y, weights, nlv, prior, scal, ctx, False)


def plsqda(X, y, weights=None, *, nlv: int, prior: str = "unif", scal: bool = False, ctx: Optional[Context] = None) -> Plslda:
    """`plsqda` — src/plsqda.jl:23-34: as plslda with one covariance per class (QDA)."""
    return _plsda(X, y, weights, nlv, prior, scal, ctx, True)


def plslda_predict(obj: Plslda, X, *, nlv=None, ctx: Optional[Context] = None):
    """`predict(object::Plslda, X; nlv)` — src/plslda.jl:107-130: (pred, posterior); lists over the clamped nlv range.
    The scores of X come from ONE device pass (transform at the largest nlv: its leading columns are the scores of every
    smaller model); the Gaussian posteriors on the m x nlv scores are p x nlev host glue like the reference's."""
    a = obj.fm_pls.P.shape[1]
    if nlv is None:
        rng = [a]
    else:
        vals = np.atleast_1d(np.asarray(nlv))
        rng = list(range(max(int(vals.min()), 0), min(int(vals.max()), a) + 1))
    if not rng or rng[0] < 1:
        raise ValueError("BoundsError: a discriminant model needs nlv >= 1 (src/plslda.jl:120 indexes fm_da[nlv])")
    T = transform(obj.fm_pls, X, nlv=rng[-1], ctx=ctx)
    T = T.cpu().numpy() if _is_torch(T) else T
    preds, posts = [], []
    for k in rng:
        da = obj.fm_da[k - 1]
        nlev = len(obj.lev)
        dens = np.empty((T.shape[0], nlev))
        for c in range(nlev):
            z = (T[:, :k] - da["mu"][c][None, :]) @ da["Uinv"][c]
            dens[:, c] = (2 * np.pi) ** (-k / 2) / np.sqrt(da["detS"][c]) * np.exp(-0.5 * np.sum(z * z, axis=1))
        A = da["wprior"][None, :] * dens
        post = A / A.sum(axis=1, keepdims=True)
        preds.append(obj.lev[np.argmax(post, axis=1)].reshape(-1, 1)); posts.append(post)
    return (preds[0], posts[0]) if len(rng) == 1 else (preds, posts)


# ---------------------------------------------------------------------------------- multiblock PLSR (§8f rank 4)
@dataclass
class Mbplsr:
    """src/mbplsr.jl:1-12 — same fields."""
    fm: Plsr
    T: object
    R: np.ndarray
    C: np.ndarray
    bscales: np.ndarray
    xmeans: list
    xscales: list
    ymeans: np.ndarray
    yscales: np.ndarray
    weights: object


def _col_stats(X, weights, want_std, ctx):
    """jch_col_stats: weighted column means (and uncorrected stds) from the device."""
    X = ensure_mat(X)
    try:
        _addr_ld(X)
    except (ValueError, TypeError):
        X = _as_colmajor_copy(X)
    dev = _is_torch(X)
    n, p = X.shape
    ctx = ctx or default_context((X.device.index or 0) if dev else 0)
    if weights is None:
        wa = None
    elif dev:
        weights = (weights if _is_torch(weights) else torch.as_tensor(np.asarray(weights, dtype=np.float64), device=X.device)).to(torch.float64).contiguous()
        wa = weights.data_ptr()
    else:
        weights = np.ascontiguousarray(np.asarray(weights.cpu() if _is_torch(weights) else weights, dtype=np.float64).reshape(-1))
        wa = weights.ctypes.data
    m = np.empty(p); sd = np.empty(p) if want_std else None
    xa, ldx = _addr_ld(X)
    if dev:
        torch.cuda.current_stream(X.device).synchronize()
    ctx.check(_lib.load().jch_col_stats(ctx._h, _lib.LOC_DEVICE if dev else _lib.LOC_HOST, xa, n, p, ldx, wa, m.ctypes.data, _np(sd)))
    return m, sd


def _hcat(blocks):
    """`reduce(hcat, Xbl)`: one column-major matrix (a copy of the blocks side by side; no arithmetic)."""
    blocks = [ensure_mat(b) for b in blocks]
    if _is_torch(blocks[0]):
        n = blocks[0].shape[0]
        out = colmajor_empty(n, sum(b.shape[1] for b in blocks), blocks[0].device)
        j = 0
        for b in blocks:
            out[:, j:j + b.shape[1]] = b.to(torch.float64); j += b.shape[1]
        return out
    return np.asfortranarray(np.hstack([np.asarray(b, dtype=np.float64) for b in blocks]))


def mbplsr(Xbl, Y, weights=None, *, nlv: int, bscal: str = "none", scal: bool = False, ctx: Optional[Context] = None) -> Mbplsr:
    """`mbplsr(Xbl, Y, weights; nlv, bscal, scal)` — src/mbplsr.jl:64-113.  The reference materialises every centred /
    scaled / block-scaled block and concatenates them; here the blocks are concatenated RAW and the whole scaling is ONE
    vector of column divisors (column std x block scale) handed to jch_plskern_fit_scaled.  Column stds / block
    Frobenius norms come from jch_col_stats."""
    if bscal not in ("none", "frob"):
        raise ValueError("bscal must be 'none' or 'frob' (src/mbplsr.jl:26)")
    Y = ensure_mat(Y)
    X = _hcat(Xbl)
    dev = _is_torch(X)
    widths = [ensure_mat(b).shape[1] for b in Xbl]
    need_std = scal or bscal == "frob"
    xm, xsd = _col_stats(X, weights, need_std, ctx)
    edges = np.concatenate([[0], np.cumsum(widths)])
    xscales = [xsd[a:b].copy() if scal else np.ones(b - a) for a, b in zip(edges[:-1], edges[1:])]
    if bscal == "frob":   # || centred (/ scaled) block ||_F in the weight metric = sqrt(sum_j var_j / scale_j^2)
        bscales = np.array([np.sqrt(np.sum((xsd[a:b] / xs) ** 2)) for (a, b), xs in zip(zip(edges[:-1], edges[1:]), xscales)])
    else:
        bscales = np.ones(len(widths))
    div = np.concatenate([xs * bs for xs, bs in zip(xscales, bscales)])
    ym, ysd = _col_stats(Y, weights, scal, ctx) if scal else (None, None)
    if dev and not _is_torch(Y):
        Yt = colmajor_empty(Y.shape[0], Y.shape[1], X.device); Yt.copy_(torch.as_tensor(np.asarray(Y, dtype=np.float64))); Y = Yt
    try:
        _addr_ld(Y)
    except (ValueError, TypeError):
        Y = _as_colmajor_copy(Y)
    fm = _fit("jch_plskern_fit_scaled", X, Y, weights, nlv, False, False, ctx, ext_scales=(div, ysd))
    xmeans = [fm.xmeans[a:b].copy() for a, b in zip(edges[:-1], edges[1:])]
    yscales = fm.yscales.copy()
    # the reference's inner fit sees pre-scaled data with scal = false: its Plsr carries unit scales and zero means
    inner = Plsr(fm.T, fm.P, fm.R, fm.W, fm.C, fm.TT, np.zeros_like(fm.xmeans), np.ones_like(fm.xscales), np.zeros_like(fm.ymeans),
                 np.ones_like(fm.yscales), fm.weights, None)
    return Mbplsr(inner, fm.T, fm.R, fm.C, bscales, xmeans, xscales, fm.ymeans.copy(), yscales, fm.weights)


def mbplsr_transform(obj: Mbplsr, Xbl, *, nlv: Optional[int] = None, ctx: Optional[Context] = None):
    """`transform(object::Mbplsr, Xbl; nlv)` — src/mbplswest.jl:220-231, as one device GEMM on the raw concatenation."""
    a = obj.R.shape[1]
    k = a if nlv is None else min(int(nlv), a)
    if k < 1:
        raise ValueError("transform needs nlv >= 1")
    shift = np.concatenate(obj.xmeans)
    div = np.concatenate([xs * bs for xs, bs in zip(obj.xscales, obj.bscales)])
    return _affine(_hcat(Xbl), shift, div, obj.R[:, :k], None, ctx)


def mbplsr_predict(obj: Mbplsr, Xbl, *, nlv=None, ctx: Optional[Context] = None):
    """`predict(object::Mbplsr, Xbl; nlv)` — src/mbplswest.jl:239-254: ymeans .+ T[:, 1:nlv] * C[:, 1:nlv]' (the reference
    does not multiply by yscales here; reproduced as is)."""
    a = obj.R.shape[1]
    if nlv is None:
        rng = [a]
    else:
        vals = np.atleast_1d(np.asarray(nlv))
        rng = list(range(max(0, int(vals.min())), min(a, int(vals.max())) + 1))
    q = obj.C.shape[0]
    T = mbplsr_transform(obj, Xbl, ctx=ctx)
    Bc = np.zeros((a, len(rng) * q))
    for i, k in enumerate(rng):
        Bc[:k, i * q:(i + 1) * q] = obj.C[:, :k].T
    out = _affine(T, None, None, Bc, np.tile(obj.ymeans, len(rng)), ctx)
    preds = [out[:, i * q:(i + 1) * q] for i in range(len(rng))]
    return preds[0] if len(preds) == 1 else preds
