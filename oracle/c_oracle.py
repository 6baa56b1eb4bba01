"""ctypes loader for oracle/libplsr_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY —
see plsr_oracle.c.  Built by `make -C oracle` (also by __graft_entry__.build())."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .plsr_oracle import Plsr

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libplsr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "plsr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libplsr_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        i64 = C.c_int64
        L.orc_fill_uniform.argtypes = [dp, i64, i64, i64, i64, i64, C.c_uint64]
        L.orc_fill_uniform.restype = None
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        sig = [dp, i64, dp, i64, dp, i64, i64, i64, C.c_int, C.c_int] + [dp] * 11
        for f in (L.orc_plskern, L.orc_plsnipals):
            f.argtypes = sig
            f.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def fill_uniform(seed: int, n: int, p: int, row0: int = 0, n_total=None) -> np.ndarray:
    n_total = n if n_total is None else n_total
    out = np.empty((n, p), dtype=np.float64, order="F")
    lib().orc_fill_uniform(_p(out), n, p, max(n, 1), row0, n_total, seed)
    return out


def _fit(fn, X, Y, weights, nlv, scal):
    """In place on Fortran-ordered float64 X (n,p), Y (n,q) like the `!` variants."""
    assert X.flags.f_contiguous and Y.flags.f_contiguous and X.dtype == np.float64 and Y.dtype == np.float64
    n, p = X.shape
    q = Y.shape[1]
    k = min(n, p, nlv)
    T = np.empty((n, k), order="F"); P = np.empty((p, k), order="F"); R = np.empty((p, k), order="F")
    W = np.empty((p, k), order="F"); Cm = np.empty((q, k), order="F"); TT = np.empty(k)
    xm = np.empty(p); xs = np.empty(p); ym = np.empty(q); ys = np.empty(q); wn = np.empty(n)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    got = fn(_p(X), max(n, 1), _p(Y), max(n, 1), _p(w), n, p, q, nlv, int(bool(scal)),
             _p(T), _p(P), _p(R), _p(W), _p(Cm), _p(TT), _p(xm), _p(xs), _p(ym), _p(ys), _p(wn))
    assert got == k
    return Plsr(T, P, R, W, Cm, TT, xm, xs, ym, ys, wn, None)


def plskern_(X, Y, weights=None, *, nlv, scal=False):
    return _fit(lib().orc_plskern, X, Y, weights, nlv, scal)


def plsnipals_(X, Y, weights=None, *, nlv, scal=False):
    return _fit(lib().orc_plsnipals, X, Y, weights, nlv, scal)


def plskern(X, Y, weights=None, *, nlv, scal=False):
    Y = np.asarray(Y, dtype=np.float64)
    Y = Y.reshape(-1, 1) if Y.ndim == 1 else Y
    return plskern_(np.array(X, dtype=np.float64, order="F", copy=True),
                    np.array(Y, dtype=np.float64, order="F", copy=True), weights, nlv=nlv, scal=scal)


def plsnipals(X, Y, weights=None, *, nlv, scal=False):
    Y = np.asarray(Y, dtype=np.float64)
    Y = Y.reshape(-1, 1) if Y.ndim == 1 else Y
    return plsnipals_(np.array(X, dtype=np.float64, order="F", copy=True),
                      np.array(Y, dtype=np.float64, order="F", copy=True), weights, nlv=nlv, scal=scal)
