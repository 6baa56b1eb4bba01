"""CPU oracle (numpy) for the plskern / plsnipals hot path of Jchemo.jl.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product path
(``jchemo.jl_amd/``) never imports this module and has no CPU fallback.

PARITY UNPINNED.  The reference (`/root/reference`, Jchemo.jl v0.1.23) is pure
Julia; there is no Julia toolchain in the build container and the reference
ships no tests, golden vectors or fixtures for this path
(`test/runtests.jl:1-2` only loads the package).  This restatement is pinned
instead by (i) the algebraic invariants of a PLS fit, (ii) agreement between
the two independent algorithms restated here (plskern vs plsnipals), (iii)
scikit-learn's ``PLSRegression`` and LAPACK ``dgesdd`` through numpy (the same
LAPACK routine Julia's ``svd`` calls), see ``tests/test_oracle.py``.

Every function cites the reference lines it restates (paths relative to
`/root/reference/`).  Arrays are float64; matrices are (n, p) numpy arrays in
either memory order (the arithmetic is order-independent).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence, Union

import numpy as np

# --------------------------------------------------------------------------
# Portable synthetic-input generator (shared with the HIP library and the C
# oracle): element k of a stream is a pure function of (seed, k), so it can be
# produced in parallel on the device and vectorised here.
#   z_k = seed + (k + 1) * 0x9E3779B97F4A7C15        (splitmix64 state walk)
#   z ^= z >> 30; z *= 0xBF58476D1CE4E5B9; z ^= z >> 27; z *= 0x94D049BB133111EB; z ^= z >> 31
#   u_k = (z >> 11) * 2**-53  in [0, 1)
# Matrices are filled in COLUMN-MAJOR order (Julia's `rand(n, p)` layout):
# element (i, j) of an n_total x p matrix is u_{i + j * n_total}.
# --------------------------------------------------------------------------
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64_uniform(seed: int, start: int, count: int) -> np.ndarray:
    """u_k for k = start .. start+count-1 (float64 in [0,1))."""
    with np.errstate(over="ignore"):
        k = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def rand_matrix(seed: int, n: int, p: int, row0: int = 0, n_total: Optional[int] = None) -> np.ndarray:
    """Rows [row0, row0+n) of the n_total x p column-major-filled matrix, as an
    (n, p) Fortran-ordered float64 array."""
    n_total = n if n_total is None else n_total
    out = np.empty((n, p), dtype=np.float64, order="F")
    for j in range(p):
        out[:, j] = splitmix64_uniform(seed, row0 + j * n_total, n)
    return out


# --------------------------------------------------------------------------
# Column utilities  (src/utility.jl)
# --------------------------------------------------------------------------
def ensure_mat(X) -> np.ndarray:
    """src/utility.jl:544-548 — vector -> (n,1) matrix, number -> (1,1)."""
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 0:
        return X.reshape(1, 1)
    if X.ndim == 1:
        return X.reshape(-1, 1)
    return X


def mweight(w) -> np.ndarray:
    """src/utility.jl:715-723 — w / sum(w) (copy)."""
    w = np.array(w, dtype=np.float64).reshape(-1)
    return w / w.sum()


def colmean(X, w) -> np.ndarray:
    """src/utility.jl:195 — vec(mweight(w)' * X)."""
    return mweight(w) @ ensure_mat(X)


def colvar(X, w) -> np.ndarray:
    """src/utility.jl:314-323 — two-pass weighted, uncorrected variance."""
    X = ensure_mat(X)
    w = mweight(w)
    z = colmean(X, w)
    return np.array([np.dot(w, (X[:, j] - z[j]) ** 2) for j in range(X.shape[1])])


def colstd(X, w) -> np.ndarray:
    """src/utility.jl:264."""
    return np.sqrt(colvar(X, mweight(w)))


def center_(X: np.ndarray, v: np.ndarray) -> None:
    """src/utility.jl:76-81 — in place."""
    X -= v[None, :]


def cscale_(X: np.ndarray, u: np.ndarray, v: np.ndarray) -> None:
    """src/utility.jl:482-487 — in place (X - u) / v."""
    X -= u[None, :]
    X /= v[None, :]


# --------------------------------------------------------------------------
# Result record  (src/plskern.jl:1-14)
# --------------------------------------------------------------------------
@dataclass
class Plsr:
    T: np.ndarray        # n x nlv
    P: np.ndarray        # p x nlv
    R: np.ndarray        # p x nlv
    W: np.ndarray        # p x nlv
    C: np.ndarray        # q x nlv
    TT: np.ndarray       # nlv
    xmeans: np.ndarray   # p
    xscales: np.ndarray  # p
    ymeans: np.ndarray   # q
    yscales: np.ndarray  # q
    weights: np.ndarray  # n (normalised)
    niter: Optional[np.ndarray] = None


def _preamble(X, Y, weights, nlv, scal):
    """src/plskern.jl:114-130 (identical in src/plsnipals.jl:39-56).  Mutates X, Y."""
    n, p = X.shape
    q = Y.shape[1]
    nlv = min(n, p, nlv)
    weights = mweight(weights)
    xmeans = colmean(X, weights)
    ymeans = colmean(Y, weights)
    xscales = np.ones(p)
    yscales = np.ones(q)
    if scal:
        xscales = colstd(X, weights)
        yscales = colstd(Y, weights)
        cscale_(X, xmeans, xscales)
        cscale_(Y, ymeans, yscales)
    else:
        center_(X, xmeans)
        center_(Y, ymeans)
    return n, p, q, nlv, weights, xmeans, ymeans, xscales, yscales


def dominant_left_sv(K: np.ndarray) -> np.ndarray:
    """src/plskern.jl:150-155 / src/plsnipals.jl:72-77 — q==1: normalised
    column; else `svd(K).U[:, 1]` (LAPACK dgesdd via numpy; sign is LAPACK's)."""
    if K.shape[1] == 1:
        w = K[:, 0].copy()
        return w / np.linalg.norm(w)
    U, _, _ = np.linalg.svd(K, full_matrices=False)
    return U[:, 0].copy()


def plskern_(X: np.ndarray, Y: np.ndarray, weights=None, *, nlv: int, scal: bool = False) -> Plsr:
    """`plskern!` — src/plskern.jl:112-178.  X and Y are overwritten with their
    centred/scaled versions (F5)."""
    assert X.dtype == np.float64 and Y.dtype == np.float64 and X.ndim == 2 and Y.ndim == 2
    if weights is None:
        weights = np.ones(X.shape[0])
    n, p, q, nlv, weights, xmeans, ymeans, xscales, yscales = _preamble(X, Y, weights, nlv, scal)
    XtY = X.T @ (weights[:, None] * Y)                      # :131-132
    T = np.empty((n, nlv)); W = np.empty((p, nlv)); P = np.empty((p, nlv))
    R = np.empty((p, nlv)); C = np.empty((q, nlv)); TT = np.empty(nlv)
    for a in range(nlv):                                    # :149-175
        w = dominant_left_sv(XtY)                           # :150-155
        r = w.copy()
        for j in range(a):                                  # :156-161
            r -= np.dot(w, P[:, j]) * R[:, j]
        t = X @ r                                           # :162
        dt = weights * t                                    # :163
        tt = np.dot(t, dt)                                  # :164
        c = (XtY.T @ r) / tt                                # :165-166
        zp = X.T @ dt                                       # :167
        XtY -= np.outer(zp, c)                              # :168
        P[:, a] = zp / tt; T[:, a] = t; W[:, a] = w         # :169-171
        R[:, a] = r; C[:, a] = c; TT[a] = tt                # :172-174
    return Plsr(T, P, R, W, C, TT, xmeans, xscales, ymeans, yscales, weights, None)


def plskern(X, Y, weights=None, *, nlv: int, scal: bool = False) -> Plsr:
    """`plskern` — src/plskern.jl:106-110 (copies, then `plskern!`)."""
    Xc = np.array(ensure_mat(X), dtype=np.float64, copy=True)
    Yc = np.array(ensure_mat(Y), dtype=np.float64, copy=True)
    return plskern_(Xc, Yc, weights, nlv=nlv, scal=scal)


def plsnipals_(X: np.ndarray, Y: np.ndarray, weights=None, *, nlv: int, scal: bool = False) -> Plsr:
    """`plsnipals!` — src/plsnipals.jl:37-97.  X and Y end up centred AND deflated."""
    assert X.dtype == np.float64 and Y.dtype == np.float64 and X.ndim == 2 and Y.ndim == 2
    if weights is None:
        weights = np.ones(X.shape[0])
    n, p, q, nlv, weights, xmeans, ymeans, xscales, yscales = _preamble(X, Y, weights, nlv, scal)
    T = np.empty((n, nlv)); W = np.empty((p, nlv)); P = np.empty((p, nlv))
    C = np.empty((q, nlv)); TT = np.empty(nlv)
    for a in range(nlv):                                    # :70-94
        XtY = X.T @ (weights[:, None] * Y)                  # :71
        w = dominant_left_sv(XtY)                           # :72-77
        t = X @ w                                           # :78
        dt = weights * t                                    # :79
        tt = np.dot(t, dt)                                  # :80
        zp = (X.T @ dt) / tt                                # :81-82
        c = (Y.T @ dt) / tt                                 # :83-84
        X -= np.outer(t, zp)                                # :86
        Y -= np.outer(t, c)                                 # :87
        P[:, a] = zp; T[:, a] = t; W[:, a] = w; C[:, a] = c; TT[a] = tt
    R = W @ np.linalg.inv(P.T @ W)                          # :95
    return Plsr(T, P, R, W, C, TT, xmeans, xscales, ymeans, yscales, weights, None)


def plsnipals(X, Y, weights=None, *, nlv: int, scal: bool = False) -> Plsr:
    """`plsnipals` — src/plsnipals.jl:31-35."""
    Xc = np.array(ensure_mat(X), dtype=np.float64, copy=True)
    Yc = np.array(ensure_mat(Y), dtype=np.float64, copy=True)
    return plsnipals_(Xc, Yc, weights, nlv=nlv, scal=scal)


# --------------------------------------------------------------------------
# Sibling algorithms: SIMPLS, ROSA, Wold's NIPALS
# --------------------------------------------------------------------------
def plssimp_(X: np.ndarray, Y: np.ndarray, weights=None, *, nlv: int, scal: bool = False) -> Plsr:
    """`plssimp!` — src/plssimp.jl:28-88 (de Jong 1993, scores not normed).  XtY is never deflated: each LV
    projects it on the orthogonal complement of the loadings found so far (:65-70); W does not exist in SIMPLS and
    is returned as R (:85-87)."""
    assert X.dtype == np.float64 and Y.dtype == np.float64 and X.ndim == 2 and Y.ndim == 2
    if weights is None:
        weights = np.ones(X.shape[0])
    n, p, q, nlv, weights, xmeans, ymeans, xscales, yscales = _preamble(X, Y, weights, nlv, scal)
    XtY = X.T @ (weights[:, None] * Y)                      # :47-48
    T = np.empty((n, nlv)); P = np.empty((p, nlv)); R = np.empty((p, nlv))
    C = np.empty((q, nlv)); TT = np.empty(nlv)
    for a in range(nlv):                                    # :64-83
        if a == 0:
            tmp = XtY.copy()
        else:
            zP = P[:, :a]
            tmp = XtY - zP @ np.linalg.inv(zP.T @ zP) @ (zP.T @ XtY)   # :69
        U, _, _ = np.linalg.svd(tmp, full_matrices=False)   # :71 (also for q == 1)
        r = U[:, 0].copy()
        t = X @ r
        dt = weights * t
        tt = np.dot(t, dt)
        c = (XtY.T @ r) / tt                                # :75-76
        zp = X.T @ dt                                       # :77
        P[:, a] = zp / tt; T[:, a] = t; R[:, a] = r; C[:, a] = c; TT[a] = tt
    return Plsr(T, P, R, R.copy(), C, TT, xmeans, xscales, ymeans, yscales, weights, None)


def plssimp(X, Y, weights=None, *, nlv: int, scal: bool = False) -> Plsr:
    """`plssimp` — src/plssimp.jl:22-26."""
    Xc = np.array(ensure_mat(X), dtype=np.float64, copy=True)
    Yc = np.array(ensure_mat(Y), dtype=np.float64, copy=True)
    return plssimp_(Xc, Yc, weights, nlv=nlv, scal=scal)


def plsrosa_(X: np.ndarray, Y: np.ndarray, weights=None, *, nlv: int, scal: bool = False) -> Plsr:
    """`plsrosa!` — src/plsrosa.jl:32-96 (Liland et al. 2016).  X stays centred (never deflated); Y is deflated in
    place (:87); scores are orthogonalised against the previous ones in the D metric (:75-76), weights against the
    previous weights (:77-79); R = W inv(P'W) (:94)."""
    assert X.dtype == np.float64 and Y.dtype == np.float64 and X.ndim == 2 and Y.ndim == 2
    if weights is None:
        weights = np.ones(X.shape[0])
    n, p, q, nlv, weights, xmeans, ymeans, xscales, yscales = _preamble(X, Y, weights, nlv, scal)
    T = np.empty((n, nlv)); W = np.empty((p, nlv)); P = np.empty((p, nlv))
    C = np.empty((q, nlv)); TT = np.empty(nlv)
    for a in range(nlv):                                    # :65-93
        XtY = X.T @ (weights[:, None] * Y)                  # :66
        w = dominant_left_sv(XtY)                           # :67-72
        t = X @ w                                           # :73
        if a > 0:
            z = T[:, :a]
            t = t - z @ (np.linalg.inv(z.T @ (weights[:, None] * z)) @ (z.T @ (weights * t)))   # :76
            z = W[:, :a]
            w = w - z @ (z.T @ w)                           # :78
            w = w / np.sqrt(np.dot(w, w))                   # :79
        dt = weights * t
        tt = np.dot(t, dt)
        c = (Y.T @ dt) / tt                                 # :83-84
        zp = (X.T @ dt) / tt                                # :85-86
        Y -= np.outer(t, c)                                 # :87
        P[:, a] = zp; T[:, a] = t; W[:, a] = w; C[:, a] = c; TT[a] = tt
    R = W @ np.linalg.inv(P.T @ W)                          # :94
    return Plsr(T, P, R, W, C, TT, xmeans, xscales, ymeans, yscales, weights, None)


def plsrosa(X, Y, weights=None, *, nlv: int, scal: bool = False) -> Plsr:
    """`plsrosa` — src/plsrosa.jl:26-30."""
    Xc = np.array(ensure_mat(X), dtype=np.float64, copy=True)
    Yc = np.array(ensure_mat(Y), dtype=np.float64, copy=True)
    return plsrosa_(Xc, Yc, weights, nlv=nlv, scal=scal)


def plswold_(X: np.ndarray, Y: np.ndarray, weights=None, *, nlv: int, tol: float = float(np.sqrt(np.finfo(float).eps)),
             maxit: int = 200, scal: bool = False) -> Plsr:
    """`plswold!` — src/plswold.jl:36-111 (Wold's NIPALS with the inner power iteration; X, Y carry the row metric
    sqrt(w) (:57-58) and are deflated in place).  The reference starts every LV from `wx .= rand(p)` (:78): that random
    vector is only ever used as the `w0` of the FIRST convergence check, which therefore never passes (dif ~ p/3);
    here the first check is skipped outright, which is the same thing without the generator.  `niter` as :93."""
    assert X.dtype == np.float64 and Y.dtype == np.float64 and X.ndim == 2 and Y.ndim == 2
    if weights is None:
        weights = np.ones(X.shape[0])
    n, p, q, nlv, weights, xmeans, ymeans, xscales, yscales = _preamble(X, Y, weights, nlv, scal)
    sqrtw = np.sqrt(weights)
    X *= sqrtw[:, None]                                     # :57
    Y *= sqrtw[:, None]                                     # :58
    Tx = np.empty((n, nlv)); Wx = np.empty((p, nlv)); Px = np.empty((p, nlv))
    Wyt = np.empty((q, nlv)); TTx = np.empty(nlv); niter = np.zeros(nlv)
    for a in range(nlv):                                    # :73-106
        ty = Y[:, 0].copy()                                 # :75
        it = 1
        wx = None
        while True:                                         # :79-92
            w0 = wx
            wx = X.T @ ty / np.dot(ty, ty)                  # :81
            wx = wx / np.linalg.norm(wx)                    # :82
            tx = X @ wx                                     # :83
            wytild = Y.T @ tx / np.dot(tx, tx)              # :84
            wy = wytild / np.linalg.norm(wytild)            # :85
            ty = Y @ wy                                     # :86
            dif = np.inf if w0 is None else float(np.sum((wx - w0) ** 2))   # :87
            it += 1
            if dif < tol or it > maxit:                     # :89-91
                break
        niter[a] = it - 1                                   # :93
        ttx = np.dot(tx, tx)
        px = (X.T @ tx) / ttx                               # :95-96
        X -= np.outer(tx, px)                               # :98
        Y -= np.outer(tx, wytild)                           # :99
        Tx[:, a] = tx; Wx[:, a] = wx; Px[:, a] = px; Wyt[:, a] = wytild; TTx[a] = ttx
    Tx = Tx / sqrtw[:, None]                                # :107
    Rx = Wx @ np.linalg.inv(Px.T @ Wx)                      # :108
    return Plsr(Tx, Px, Rx, Wx, Wyt, TTx, xmeans, xscales, ymeans, yscales, weights, niter)


def plswold(X, Y, weights=None, *, nlv: int, tol: float = float(np.sqrt(np.finfo(float).eps)), maxit: int = 200,
            scal: bool = False) -> Plsr:
    """`plswold` — src/plswold.jl:30-34."""
    Xc = np.array(ensure_mat(X), dtype=np.float64, copy=True)
    Yc = np.array(ensure_mat(Y), dtype=np.float64, copy=True)
    return plswold_(Xc, Yc, weights, nlv=nlv, tol=tol, maxit=maxit, scal=scal)


# --------------------------------------------------------------------------
# Accessors  (src/plskern.jl:187-260)
# --------------------------------------------------------------------------
def transform(fm: Plsr, X, *, nlv: Optional[int] = None) -> np.ndarray:
    """src/plskern.jl:187-195."""
    X = ensure_mat(X)
    a = fm.T.shape[1]
    nlv = a if nlv is None else min(nlv, a)
    return ((X - fm.xmeans[None, :]) / fm.xscales[None, :]) @ fm.R[:, :nlv]


def coef(fm: Plsr, *, nlv: Optional[int] = None):
    """src/plskern.jl:207-217 — returns (B p x q, int 1 x q)."""
    a = fm.T.shape[1]
    nlv = a if nlv is None else min(nlv, a)
    beta = fm.C[:, :nlv].T
    B = (fm.R[:, :nlv] / fm.xscales[:, None]) @ beta * fm.yscales[None, :]
    intercept = fm.ymeans[None, :] - fm.xmeans[None, :] @ B
    return B, intercept


def predict(fm: Plsr, X, *, nlv: Union[None, int, Sequence[int]] = None):
    """src/plskern.jl:226-238 — a collection of nlv is replaced by the contiguous
    range max(0,min):min(a,max); one value -> matrix, several -> list."""
    X = ensure_mat(X)
    a = fm.T.shape[1]
    if nlv is None:
        rng = [a]
    else:
        vals = np.atleast_1d(np.asarray(nlv))
        rng = list(range(max(0, int(vals.min())), min(a, int(vals.max())) + 1))
    preds = []
    for k in rng:
        B, intercept = coef(fm, nlv=k)
        preds.append(intercept + X @ B)
    return preds[0] if len(preds) == 1 else preds


def summary(fm: Plsr, X):
    """src/plskern.jl:246-260 — dict(nlv, var, pvar, cumpvar)."""
    X = ensure_mat(X)
    n, nlv = fm.T.shape
    Xs = (X - fm.xmeans[None, :]) / fm.xscales[None, :]
    sstot = float(np.sum(fm.weights @ (Xs ** 2)))
    tt_adj = np.sum(fm.P ** 2, axis=0) * fm.TT
    pvar = tt_adj / sstot
    return dict(nlv=np.arange(1, nlv + 1), var=tt_adj / n, pvar=pvar, cumpvar=np.cumsum(pvar))


# --------------------------------------------------------------------------
# Host-simulated row sharding: the same plskern, with every
# n-length reduction computed as G per-shard partials summed in rank order.
# `allreduce` may be replaced by a real collective (the gloo tests do).
# --------------------------------------------------------------------------
def plskern_sharded(shards_X, shards_Y, shards_w, *, nlv: int, scal: bool = False, allreduce=None):
    """shards_*: lists (one entry per rank held by THIS process) of row blocks.
    allreduce(vec) -> vec summed over every rank of every process; default sums
    the local list only.  Returns (Plsr with T = list of per-shard blocks)."""
    def ar(parts):
        s = np.sum(np.stack(parts, 0), axis=0)
        return allreduce(s) if allreduce is not None else s

    Xs = [np.array(x, dtype=np.float64, copy=True) for x in shards_X]
    Ys = [np.array(ensure_mat(y), dtype=np.float64, copy=True) for y in shards_Y]
    p = Xs[0].shape[1]; q = Ys[0].shape[1]
    head = ar([np.array([float(w.sum()), float(len(w))]) for w in shards_w])
    wsum, n_total = head[0], int(round(head[1]))
    nlv = min(n_total, p, nlv)
    ds = [np.asarray(w, dtype=np.float64) / wsum for w in shards_w]
    mom = ar([np.concatenate([d @ x, d @ y]) for d, x, y in zip(ds, Xs, Ys)])
    xmeans, ymeans = mom[:p], mom[p:]
    xscales, yscales = np.ones(p), np.ones(q)
    if scal:
        var = ar([np.concatenate([d @ (x - xmeans) ** 2, d @ (y - ymeans) ** 2]) for d, x, y in zip(ds, Xs, Ys)])
        xscales, yscales = np.sqrt(var[:p]), np.sqrt(var[p:])
    for x, y in zip(Xs, Ys):
        cscale_(x, xmeans, xscales); cscale_(y, ymeans, yscales)
    K = ar([(x.T @ (d[:, None] * y)).ravel() for d, x, y in zip(ds, Xs, Ys)]).reshape(p, q)
    W = np.empty((p, nlv)); P = np.empty((p, nlv)); R = np.empty((p, nlv))
    C = np.empty((q, nlv)); TT = np.empty(nlv); Ts = [np.empty((x.shape[0], nlv)) for x in Xs]
    for a in range(nlv):
        w = dominant_left_sv(K)
        r = w.copy()
        for j in range(a):
            r -= np.dot(w, P[:, j]) * R[:, j]
        parts = []
        for g, (d, x) in enumerate(zip(ds, Xs)):
            t = x @ r
            Ts[g][:, a] = t
            dt = d * t
            parts.append(np.concatenate([x.T @ dt, [np.dot(t, dt)]]))
        red = ar(parts)                                     # ONE collective per LV: [zp (p), tt]
        zp, tt = red[:p], red[p]
        c = (K.T @ r) / tt
        K -= np.outer(zp, c)
        P[:, a] = zp / tt; W[:, a] = w; R[:, a] = r; C[:, a] = c; TT[a] = tt
    wn = [d for d in ds]
    return Plsr(Ts, P, R, W, C, TT, xmeans, xscales, ymeans, yscales, wn, None)


# --------------------------------------------------------------------------
# Comparison helpers used by the parity tests
# --------------------------------------------------------------------------
def sign_align(ref_W: np.ndarray, new_W: np.ndarray) -> np.ndarray:
    """Per-LV sign s_a = sign(<w_ref,a , w_new,a>) (F3)."""
    s = np.sign(np.sum(ref_W * new_W, axis=0))
    s[s == 0] = 1.0
    return s


def rel_fro(a: np.ndarray, b: np.ndarray) -> float:
    den = np.linalg.norm(a)
    return float(np.linalg.norm(a - b) / (den if den > 0 else 1.0))


# --------------------------------------------------------------------------
# kNN-LWPLSR (BASELINE.json configs[4])
# --------------------------------------------------------------------------
def mad(x) -> float:
    """src/utility.jl:679 — 1.4826 * median(|x - median(x)|)."""
    x = np.asarray(x, dtype=np.float64)
    return 1.4826 * float(np.median(np.abs(x - np.median(x))))


def wdist(d, *, h: float = 2.0, cri: float = 4.0, squared: bool = False) -> np.ndarray:
    """src/wdist.jl:58-75."""
    d = np.array(d, dtype=np.float64, copy=True)
    if squared:
        d = d ** 2
    zmed = float(np.median(d))
    zmad = mad(d)
    cutoff = zmed + cri * zmad
    with np.errstate(all="ignore"):
        w = np.where(d <= cutoff, np.exp(-d / (h * zmad)), 0.0)
        w = w / np.max(w)
    w[np.isnan(w)] = 1.0
    return w


def getknn(Xtrain, X, *, k: int = 1, metric: str = "eucl"):
    """src/getknn.jl:29-57 — brute-force k nearest neighbours, sorted by increasing distance (unsquared).
    Returns (ind (m, k) 0-based, d (m, k)).  Ties are broken by index (NearestNeighbors leaves them arbitrary)."""
    Xtrain = ensure_mat(Xtrain); X = ensure_mat(X)
    n, p = Xtrain.shape
    k = min(k, n)
    if metric == "mahal":
        S = np.cov(Xtrain, rowvar=False, bias=True).reshape(p, p)
        if p == 1:
            Uinv = np.array([[1.0 / np.sqrt(S[0, 0])]])
        else:
            try:
                U = np.linalg.cholesky(S).T                      # S = U'U
                Uinv = np.linalg.inv(U)
            except np.linalg.LinAlgError:
                Uinv = np.diag(1.0 / np.diag(S))                 # sic (getknn.jl:43)
        Xtrain = Xtrain @ Uinv
        X = X @ Uinv
    elif metric != "eucl":
        raise ValueError(metric)
    ind = np.empty((X.shape[0], k), dtype=np.int64)
    dist = np.empty((X.shape[0], k))
    for i in range(X.shape[0]):
        d2 = np.sum((Xtrain - X[i]) ** 2, axis=1)
        # the k smallest by (distance, index): candidates = everything not beyond the k-th smallest distance, ordered
        # by the same two keys — identical to lexsort over all n rows, without sorting all n
        cand = np.nonzero(d2 <= np.partition(d2, k - 1)[k - 1])[0]
        order = cand[np.lexsort((cand, d2[cand]))][:k]
        ind[i] = order
        dist[i] = np.sqrt(d2[order])
    return ind, dist


def locwlv(Xtrain, Ytrain, X, *, listnn, listw=None, nlv, scal: bool = False):
    """src/locwlv.jl:9-48 with fun = plskern.  Returns pred (m, q, le_nlv) for nlv = max(0,min):min(p,max)."""
    Xtrain = ensure_mat(Xtrain); Ytrain = ensure_mat(Ytrain); X = ensure_mat(X)
    p = Xtrain.shape[1]; m = X.shape[0]; q = Ytrain.shape[1]
    vals = np.atleast_1d(np.asarray(nlv))
    rng = list(range(max(0, int(vals.min())), min(p, int(vals.max())) + 1))
    zpred = np.empty((m, q, len(rng)))
    for i in range(m):
        s = np.asarray(listnn[i])
        zY = Ytrain[s, :]
        if q == 1 and len(np.unique(zY)) == 1:
            zpred[i, :, :] = zY[0, 0]
            continue
        w = None if listw is None else listw[i]
        fm = plskern(Xtrain[s, :], zY, w, nlv=max(rng), scal=scal)
        for a, kk in enumerate(rng):
            zpred[i, :, a] = predict(fm, X[i:i + 1, :], nlv=kk)[0]
    return zpred, rng


@dataclass
class Lwplsr:
    """src/lwplsr.jl:1-12."""
    X: np.ndarray
    Y: np.ndarray
    fm: Optional[Plsr]
    metric: str
    h: float
    k: int
    nlv: int
    tol: float
    scal: bool


def lwplsr(X, Y, *, nlvdis: int, metric: str, h: float, k: int, nlv: int, tol: float = 1e-4, scal: bool = False) -> Lwplsr:
    """src/lwplsr.jl:114-126."""
    X = ensure_mat(X); Y = ensure_mat(Y)
    fm = None if nlvdis == 0 else plskern(X, Y, nlv=nlvdis, scal=scal)
    return Lwplsr(np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64), fm, metric, h, k, nlv, tol, scal)


def lwplsr_predict(obj: Lwplsr, X, *, nlv=None):
    """src/lwplsr.jl:134-166.  Returns dict(pred (m,q,le) , rng, listnn, listd, listw)."""
    X = ensure_mat(X)
    a = obj.nlv
    if nlv is None:
        rng_req = [a]
    else:
        vals = np.atleast_1d(np.asarray(nlv))
        rng_req = list(range(max(int(vals.min()), 0), min(int(vals.max()), a) + 1))
    if obj.fm is None:
        if obj.scal:
            xs = np.sqrt(colvar(obj.X, np.ones(obj.X.shape[0])))
            ind, d = getknn(obj.X / xs, X / xs, k=obj.k, metric=obj.metric)
        else:
            ind, d = getknn(obj.X, X, k=obj.k, metric=obj.metric)
    else:
        ind, d = getknn(obj.fm.T, transform(obj.fm, X), k=obj.k, metric=obj.metric)
    listw = np.empty_like(d)
    for i in range(d.shape[0]):
        w = wdist(d[i], h=obj.h)
        w[w < obj.tol] = obj.tol
        listw[i] = w
    pred, rng = locwlv(obj.X, obj.Y, X, listnn=ind, listw=listw, nlv=rng_req, scal=obj.scal)
    return dict(pred=pred, rng=rng, listnn=ind, listd=d, listw=listw)


# --------------------------------------------------------------------------
# Validation / tuning helpers: scores, segments, gridscorelv, gridcvlv
# --------------------------------------------------------------------------
def residreg(pred, Y):
    """src/scores.jl:241."""
    return ensure_mat(Y) - ensure_mat(pred)


def msep(pred, Y):
    """src/scores.jl:155-158."""
    return np.mean(residreg(pred, Y) ** 2, axis=0).reshape(1, -1)


def rmsep(pred, Y):
    """src/scores.jl:268."""
    return np.sqrt(msep(pred, Y))


def ssr(pred, Y):
    """src/scores.jl:426-429."""
    return np.sum(residreg(pred, Y) ** 2, axis=0).reshape(1, -1)


def bias(pred, Y):
    """src/scores.jl:25-28."""
    return (-np.mean(residreg(pred, Y), axis=0)).reshape(1, -1)


def r2(pred, Y):
    """src/scores.jl:190-196."""
    Y = ensure_mat(Y)
    M = np.tile(Y.mean(axis=0), (Y.shape[0], 1))
    return 1 - msep(pred, Y) / msep(M, Y)


def cor2(pred, Y):
    """src/scores.jl:54-62."""
    pred = ensure_mat(pred); Y = ensure_mat(Y)
    return np.array([[np.corrcoef(pred[:, k], Y[:, k])[0, 1] ** 2 for k in range(Y.shape[1])]])


def rmrow(X, s):
    """src/utility.jl:1020-1023 (s: 0-based indices here)."""
    keep = np.setdiff1d(np.arange(np.asarray(X).shape[0]), np.asarray(s))
    return np.asarray(X)[keep]


def mpar(**kwargs):
    """src/mpar.jl:15-24 — all combinations of the parameter values; `Base.product` order: the FIRST keyword varies
    fastest.  Returns {name: list of ncomb values}."""
    import itertools
    names = list(kwargs)
    vals = [list(v) if isinstance(v, (list, tuple, np.ndarray, range)) else [v] for v in kwargs.values()]
    out = {nm: [] for nm in names}
    for c in itertools.product(*reversed(vals)):          # last factor fastest == first keyword fastest
        for nm, x in zip(names, reversed(c)):
            out[nm].append(x)
    return out


def _pars_rows(pars):
    if pars is None:
        return [dict()]
    names = list(pars)
    ncomb = len(pars[names[0]])       # src/gridscore.jl:192 `length(pars[1])`
    return [{nm: pars[nm][i] for nm in names} for i in range(ncomb)]


def gridscorelv(Xtrain, Ytrain, X, Y, *, score, fun, nlv, pars=None):
    """src/gridscore.jl:167-221: per parameter combination (element-wise over the `pars` vectors, :192-195; one
    implicit combination when pars = nothing) fit once at max(nlv), predict for the whole (clamped) range, score each.
    Returns (nlv values, res (ncomb * le_nlv, q)) — rows ordered combination-major like :204-216."""
    Xtrain = ensure_mat(Xtrain); Ytrain = ensure_mat(Ytrain)
    p = Xtrain.shape[1]
    vals = np.atleast_1d(np.asarray(nlv))
    rng = list(range(max(0, int(vals.min())), min(p, int(vals.max())) + 1))
    blocks = []
    for kw in _pars_rows(pars):
        fm = fun(Xtrain, Ytrain, nlv=max(rng), **kw)
        if isinstance(fm, Plsr):
            pred = predict(fm, X, nlv=rng)
            pred = [pred] if len(rng) == 1 else pred
        else:                                   # Lwplsr: pred (m, q, le) over its own clamped range
            out = lwplsr_predict(fm, X, nlv=rng)
            pred = [out["pred"][:, :, i] for i in range(out["pred"].shape[2])]
        blocks.append(np.vstack([score(pr, Y) for pr in pred]))
    return rng, np.vstack(blocks)


def gridcvlv(X, Y, *, segm, score, fun, nlv, pars=None):
    """src/gridcv.jl:187-228.  segm: list (replications) of lists (segments) of 0-based row indices.  Returns
    (nlv values, res = mean over all (repl, segm) (ncomb * le_nlv, q), res_rep (nrep, nsegm, ncomb * le_nlv, q))."""
    X = ensure_mat(X); Y = ensure_mat(Y)
    p = X.shape[1]
    vals = np.atleast_1d(np.asarray(nlv))
    rng = list(range(max(0, int(vals.min())), min(p, int(vals.max())) + 1))
    rep = []
    for listsegm in segm:
        zres = []
        for s in listsegm:
            s = np.asarray(s)
            _, r = gridscorelv(rmrow(X, s), rmrow(Y, s), X[s, :], Y[s, :], score=score, fun=fun, nlv=rng, pars=pars)
            zres.append(r)
        rep.append(np.stack(zres))
    res_rep = np.stack(rep)
    return rng, res_rep.mean(axis=(0, 1)), res_rep


# --------------------------------------------------------------------------
# vip / xfit / xresid (thin accessors of a fitted Plsr)
# --------------------------------------------------------------------------
def corm(X, Y, w):
    """src/utility.jl:361-374 — weighted correlation between the columns of X and of Y."""
    X = np.array(ensure_mat(X), dtype=np.float64, copy=True); Y = np.array(ensure_mat(Y), dtype=np.float64, copy=True)
    w = mweight(w)
    X = (X - colmean(X, w)) / colstd(X, w)
    Y = (Y - colmean(Y, w)) / colstd(Y, w)
    return X.T @ (w[:, None] * Y)


def vip(fm: Plsr, Y=None, *, nlv: Optional[int] = None):
    """src/vip.jl:62-107 — variable importance in projection.  Without Y: sst_a = tr(C_a C_a') t_a'D t_a (:76-82);
    with Y: the redundancies rd(Y, T, weights) (:101, src/angles.jl:97-105).  Returns dict(imp, W2, sst | rdd)."""
    a = fm.T.shape[1]
    p = fm.W.shape[0]
    nlv = a if nlv is None else min(nlv, a)
    W2 = fm.W[:, :nlv] ** 2
    if Y is None:
        sqrtw = np.sqrt(fm.weights)
        sst = np.zeros(nlv)
        for i in range(nlv):
            t = sqrtw * fm.T[:, i]
            sst[i] = np.sum(fm.C[:, i] ** 2) * np.dot(t, t)
        A = (sst[None, :] * W2).sum(axis=1)
        return dict(imp=np.sqrt(A / (sst.sum() / p)), W2=W2, sst=sst)
    Y = ensure_mat(Y)
    rdd = (corm(Y, fm.T[:, :nlv], fm.weights) ** 2).sum(axis=0, keepdims=True) / Y.shape[1]
    A = (rdd * W2).sum(axis=1)
    return dict(imp=np.sqrt(A / (rdd.sum() / p)), W2=W2, rdd=rdd)


def xfit(fm: Plsr, X, *, nlv: Optional[int] = None) -> np.ndarray:
    """src/xfit.jl:37-56 — X reconstructed from nlv LVs, in the original scale (nlv = 0: the column means)."""
    X = np.array(ensure_mat(X), dtype=np.float64, copy=True)
    a = fm.T.shape[1]
    nlv = a if nlv is None else min(nlv, a)
    if nlv == 0:
        X[:] = fm.xmeans[None, :]
        return X
    return transform(fm, X, nlv=nlv) @ fm.P[:, :nlv].T * fm.xscales[None, :] + fm.xmeans[None, :]


def xresid(fm: Plsr, X, *, nlv: Optional[int] = None) -> np.ndarray:
    """src/xfit.jl:86-93 — E = X - xfit(X)."""
    X = np.array(ensure_mat(X), dtype=np.float64, copy=True)
    return X - xfit(fm, X, nlv=nlv)


# --------------------------------------------------------------------------
# PLSR-DA: plskern on the dummy table of the classes
# --------------------------------------------------------------------------
def dummy(y):
    """src/utility.jl:509-519 — (Y n x nlev of 0/1, sorted levels)."""
    y = np.asarray(y).reshape(-1)
    lev = np.unique(y)
    return (y[:, None] == lev[None, :]).astype(np.float64), lev


def plsrda(X, y, weights=None, *, nlv: int, scal: bool = False):
    """src/plsrda.jl:71-77 — returns (fm, lev, ni)."""
    Y, lev = dummy(y)
    ni = np.array([(np.asarray(y).reshape(-1) == l).sum() for l in lev])
    return plskern(X, Y, weights, nlv=nlv, scal=scal), lev, ni


def plsrda_predict(model, X, *, nlv=None):
    """src/plsrda.jl:95-120 — (pred list of (m,1) labels, posterior list of (m, nlev)); one nlv -> bare arrays."""
    fm, lev, _ = model
    a = fm.T.shape[1]
    if nlv is None:
        rng = [a]
    else:
        vals = np.atleast_1d(np.asarray(nlv))
        rng = list(range(max(int(vals.min()), 0), min(int(vals.max()), a) + 1))
    post = [predict(fm, X, nlv=k) for k in rng]
    pred = [lev[np.argmax(z, axis=1)].reshape(-1, 1) for z in post]      # ties: the first maximum, like Julia's argmax
    return (pred[0], post[0]) if len(rng) == 1 else (pred, post)


# --------------------------------------------------------------------------
# PLS-LDA / PLS-QDA: plskern on the class dummy table, then LDA / QDA on the scores
# --------------------------------------------------------------------------
def matW(X, y):
    """src/matW.jl:27-57 — within-class covariances (uncorrected) Wi and their pooled W = sum (ni / n) Wi; a class with
    a single observation gets the covariance of the whole X."""
    X = ensure_mat(X); y = np.asarray(y).reshape(-1)
    lev = np.unique(y)
    ni = np.array([(y == l).sum() for l in lev])
    sigma_1 = np.cov(X, rowvar=False, bias=True).reshape(X.shape[1], X.shape[1]) if np.any(ni == 1) else None
    Wi = [sigma_1 if ni[i] == 1 else np.cov(X[y == lev[i]], rowvar=False, bias=True).reshape(X.shape[1], X.shape[1])
          for i in range(len(lev))]
    w = ni / ni.sum()
    W = sum(w[i] * Wi[i] for i in range(len(lev)))
    return W, Wi, lev, ni


def dmnorm(mu, S):
    """src/dmnorm.jl:112-128 — (mu, inv(chol(S).U), det S) of a Gaussian density."""
    U = np.linalg.cholesky(np.asarray(S, dtype=np.float64)).T
    detS = float(np.prod(np.diag(U)) ** 2)
    if detS == 0:
        detS = 1e-20
    return np.asarray(mu, dtype=np.float64), np.linalg.inv(U), detS


def dmnorm_predict(dm, X):
    """src/dmnorm.jl:136-143."""
    mu, Uinv, detS = dm
    X = ensure_mat(X)
    z = (X - mu[None, :]) @ Uinv
    d = np.sum(z * z, axis=1)
    return (2 * np.pi) ** (-X.shape[1] / 2) / np.sqrt(detS) * np.exp(-0.5 * d)


def _wprior(prior, ni):
    if prior == "unif":
        return np.ones(len(ni)) / len(ni)
    if prior == "prop":
        return ni / ni.sum()
    raise ValueError("prior must be 'unif' or 'prop'")


def lda(X, y, *, prior="unif"):
    """src/lda.jl:56-78 — class centres, pooled W * n / (n - nlev), one Gaussian per class."""
    X = ensure_mat(X); y = np.asarray(y).reshape(-1)
    n = X.shape[0]
    W, Wi, lev, ni = matW(X, y)
    ct = np.stack([X[y == l].mean(axis=0) for l in lev])
    W = W * n / (n - len(lev))
    return dict(fm=[dmnorm(ct[i], W) for i in range(len(lev))], wprior=_wprior(prior, ni), lev=lev, ni=ni)


def qda(X, y, *, prior="unif"):
    """src/qda.jl:55-79 — one covariance per class, Wi * ni / (ni - 1)."""
    X = ensure_mat(X); y = np.asarray(y).reshape(-1)
    W, Wi, lev, ni = matW(X, y)
    ct = np.stack([X[y == l].mean(axis=0) for l in lev])
    fm = [dmnorm(ct[i], Wi[i] if ni[i] == 1 else Wi[i] * ni[i] / (ni[i] - 1)) for i in range(len(lev))]
    return dict(fm=fm, wprior=_wprior(prior, ni), lev=lev, ni=ni)


def da_predict(model, X):
    """src/lda.jl:85-99 / src/qda.jl:87-102 — (pred, dens, posterior)."""
    X = ensure_mat(X)
    dens = np.stack([dmnorm_predict(dm, X) for dm in model["fm"]], axis=1)
    A = model["wprior"][None, :] * dens
    posterior = A / A.sum(axis=1, keepdims=True)
    pred = model["lev"][np.argmax(posterior, axis=1)].reshape(-1, 1)
    return pred, dens, posterior


def plslda(X, y, weights=None, *, nlv: int, prior="unif", scal: bool = False, da=lda):
    """src/plslda.jl:76-88 (plsqda: src/plsqda.jl:23-34 with da = qda): plskern on dummy(y), then one discriminant
    model per number of LVs on the scores T[:, 1:i]."""
    Yd, lev = dummy(y)
    fm_pls = plskern(X, Yd, weights, nlv=nlv, scal=scal)
    yv = np.asarray(y).reshape(-1)
    fm_da = [da(fm_pls.T[:, :i + 1], yv, prior=prior) for i in range(fm_pls.T.shape[1])]
    return dict(fm_pls=fm_pls, fm_da=fm_da, lev=lev, ni=np.array([(yv == l).sum() for l in lev]))


def plsqda(X, y, weights=None, *, nlv: int, prior="unif", scal: bool = False):
    return plslda(X, y, weights, nlv=nlv, prior=prior, scal=scal, da=qda)


def plslda_predict(model, X, *, nlv=None):
    """src/plslda.jl:107-130 — per requested nlv: scores, discriminant prediction on them.  Returns lists (pred,
    posterior) over the clamped range (single entry -> the bare arrays)."""
    a = model["fm_pls"].T.shape[1]
    if nlv is None:
        rng = [a]
    else:
        vals = np.atleast_1d(np.asarray(nlv))
        rng = list(range(max(int(vals.min()), 0), min(int(vals.max()), a) + 1))
    if not rng or rng[0] < 1:
        raise ValueError("BoundsError: fm_da[nlv] needs nlv >= 1 (src/plslda.jl:120)")
    preds, posts = [], []
    for k in rng:
        T = transform(model["fm_pls"], X, nlv=k)
        pr, _, po = da_predict(model["fm_da"][k - 1], T)
        preds.append(pr); posts.append(po)
    return (preds[0], posts[0]) if len(rng) == 1 else (preds, posts)


# --------------------------------------------------------------------------
# Multiblock PLSR: block-scaled concatenation, then plskern
# --------------------------------------------------------------------------
def mbplsr(Xbl, Y, weights=None, *, nlv: int, bscal: str = "none", scal: bool = False):
    """src/mbplsr.jl:64-113 — per-block centring (/ column scaling), optional "frob" block scaling
    (src/blockscal.jl:86-94: each block divided by its weighted Frobenius norm), plskern(scal = false) on the
    concatenation.  Returns a dict with the fields of `Mbplsr` (:1-12)."""
    Xbl = [np.array(ensure_mat(b), dtype=np.float64, copy=True) for b in Xbl]
    Y = np.array(ensure_mat(Y), dtype=np.float64, copy=True)
    n = Xbl[0].shape[0]
    w = mweight(np.ones(n) if weights is None else weights)
    xmeans, xscales = [], []
    for k, b in enumerate(Xbl):
        m = colmean(b, w); sc = colstd(b, w) if scal else np.ones(b.shape[1])
        Xbl[k] = (b - m) / sc
        xmeans.append(m); xscales.append(sc)
    ymeans = colmean(Y, w); yscales = colstd(Y, w) if scal else np.ones(Y.shape[1])
    Y = (Y - ymeans) / yscales
    if bscal == "none":
        bscales = np.ones(len(Xbl))
    elif bscal == "frob":
        bscales = np.array([np.sqrt(np.sum(w[:, None] * b ** 2)) for b in Xbl])     # frob(X, w): src/utility.jl:591-599
    else:
        raise ValueError("bscal must be 'none' or 'frob'")
    X = np.hstack([b / bs for b, bs in zip(Xbl, bscales)])
    fm = plskern(X, Y, w, nlv=nlv, scal=False)
    return dict(fm=fm, T=fm.T, R=fm.R, C=fm.C, bscales=bscales, xmeans=xmeans, xscales=xscales, ymeans=ymeans, yscales=yscales,
                weights=w)


def mbplsr_transform(obj, Xbl, *, nlv: Optional[int] = None):
    """src/mbplswest.jl:220-231."""
    a = obj["T"].shape[1]
    nlv = a if nlv is None else min(nlv, a)
    Z = np.hstack([(np.asarray(ensure_mat(b), dtype=np.float64) - m) / sc / bs
                   for b, m, sc, bs in zip(Xbl, obj["xmeans"], obj["xscales"], obj["bscales"])])
    return Z @ obj["R"][:, :nlv]


def mbplsr_predict(obj, Xbl, *, nlv=None):
    """src/mbplswest.jl:239-254 — `int .+ T[:, 1:nlv] * C[:, 1:nlv]'` with int = ymeans: the reference does NOT multiply
    by yscales here (with scal = true its predictions stay in the scaled Y units, shifted by ymeans); reproduced as is."""
    a = obj["T"].shape[1]
    if nlv is None:
        rng = [a]
    else:
        vals = np.atleast_1d(np.asarray(nlv))
        rng = list(range(max(0, int(vals.min())), min(a, int(vals.max())) + 1))
    T = mbplsr_transform(obj, Xbl)
    pred = [obj["ymeans"][None, :] + T[:, :k] @ obj["C"][:, :k].T for k in rng]
    return pred[0] if len(rng) == 1 else pred
