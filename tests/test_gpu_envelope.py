"""Round 4 (-m gpu): no reference-valid call may fail.  The reference has no limits on k, p, q, nlv (src/locwlv.jl:9-48,
src/getknn.jl:29-57, src/plskern.jl:112-178, src/plsrda.jl:71-77); every shape limit the batched / LDS-resident kernels have is
backed by a slower generic path inside the library (csrc/lwplsr_generic.hip, csrc/smallstate.hip with its matrices in global
memory, the tiled SYRK behind jch_weighted_cov).  Each test takes one lifted limit through the C ABI and compares with the oracle."""
import numpy as np
import pytest

from oracle import plsr_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def J():
    import jchemo_hip
    return jchemo_hip


@pytest.fixture(scope="module")
def ctx(J):
    c = J.Context(0)
    yield c
    c.close()


def _spectra(seed, n, p, nsrc=6, noise=0.02):
    """latent sources + noise (spectra-like: the local fits stay well conditioned)"""
    S = O.rand_matrix(seed, nsrc, p)
    A = O.rand_matrix(seed + 1, n, nsrc)
    return A @ S + noise * O.rand_matrix(seed + 2, n, p), A


def _cmp_lw(ref, res, tol_pred=1e-7):
    assert np.array_equal(res.listnn, ref["listnn"])
    assert O.rel_fro(ref["listd"], res.listd) < 1e-9
    assert O.rel_fro(ref["listw"], res.listw) < 1e-7
    pred = np.stack(res.pred, axis=2) if isinstance(res.pred, list) else res.pred[:, :, None]
    assert O.rel_fro(ref["pred"], pred) < tol_pred


def test_lwplsr_wide_rows_p2500(J, ctx):
    """p > 2048: the batched local-fit kernels hold a row in 16 register chunks of 128; beyond, one jch_plskern_fit per query
    (wide two-pass sweep) + jch_predict — src/locwlv.jl:18-39."""
    n, p, m = 500, 2500, 5
    X, A = _spectra(11, n + m, p)
    y = A[:, 0] - 2.0 * A[:, 1] + 0.5 * A[:, 2] ** 2
    kw = dict(nlvdis=5, metric="mahal", h=2.0, k=60, nlv=4)
    ref = O.lwplsr_predict(O.lwplsr(X[:n], y[:n], **kw), X[n:], nlv=range(0, 5))
    res = J.predict(J.lwplsr(X[:n], y[:n], ctx=ctx, **kw), X[n:], nlv=range(0, 5), ctx=ctx)
    _cmp_lw(ref, res)


def test_lwplsr_k1000_neighbours(J, ctx):
    """k > 768: beyond the kNN scan's LDS candidate buffers the selection is exact per query (radix select + sort in global
    memory), and the k x p local fit runs through jch_plskern_fit."""
    n, p, m = 3000, 40, 6
    X, A = _spectra(21, n + m, p, noise=0.05)
    y = A[:, 0] + A[:, 1] * A[:, 2]
    kw = dict(nlvdis=4, metric="eucl", h=3.0, k=1000, nlv=3)
    ref = O.lwplsr_predict(O.lwplsr(X[:n], y[:n], **kw), X[n:], nlv=range(0, 4))
    res = J.predict(J.lwplsr(X[:n], y[:n], ctx=ctx, **kw), X[n:], nlv=range(0, 4), ctx=ctx)
    _cmp_lw(ref, res)
    # k = n: every training row is a neighbour of every query (src/getknn.jl:33 clamps k to n)
    kw["k"] = n + 50
    ref = O.lwplsr_predict(O.lwplsr(X[:n], y[:n], **kw), X[n:n + 2], nlv=2)
    res = J.predict(J.lwplsr(X[:n], y[:n], ctx=ctx, **kw), X[n:n + 2], nlv=2, ctx=ctx)
    assert res.listnn.shape == (2, n) and np.array_equal(np.sort(res.listnn, axis=1), np.tile(np.arange(n), (2, 1)))
    assert np.array_equal(res.listnn, ref["listnn"])
    assert O.rel_fro(ref["pred"][:, :, 0], res.pred) < 1e-7


def test_generic_knn_is_the_same_selection_as_the_scan(J, ctx, monkeypatch):
    """JCH_KNN_GENERIC=1 forces the generic selection where the scan applies: same neighbours in the same order, the same
    distances to the bit (same expression, same column order), weights to rounding; with ties (lattice coordinates) too."""
    n, p, m = 5000, 12, 9
    X = np.floor(O.rand_matrix(31, n + m, p) * 3.0)            # integer lattice: many equal distances
    y = X[:, 0] - X[:, 1] + 0.01 * O.rand_matrix(32, n + m, 1)[:, 0]
    kw = dict(nlvdis=0, metric="eucl", h=2.0, k=150, nlv=2)
    a = J.predict(J.lwplsr(X[:n], y[:n], ctx=ctx, **kw), X[n:], nlv=2, ctx=ctx)
    monkeypatch.setenv("JCH_KNN_GENERIC", "1")
    b = J.predict(J.lwplsr(X[:n], y[:n], ctx=ctx, **kw), X[n:], nlv=2, ctx=ctx)
    assert np.array_equal(a.listnn, b.listnn) and np.array_equal(a.listd, b.listd)
    assert O.rel_fro(a.listw, b.listw) < 1e-12 and O.rel_fro(a.pred, b.pred) < 1e-10
    ref = O.lwplsr_predict(O.lwplsr(X[:n], y[:n], **kw), X[n:], nlv=2)
    assert np.array_equal(b.listnn, ref["listnn"])


def test_generic_local_fits_match_the_batched_kernels(J, ctx, monkeypatch):
    """JCH_LOCW_GENERIC=1 forces the per-query fits where the batched kernels apply: predictions agree to rounding; the
    constant-y shortcut (src/locwlv.jl:25-28) is taken by both."""
    n, p, m = 1500, 60, 7
    X, A = _spectra(41, n + m, p, noise=0.05)
    y = np.round(A[:, 0] * 2.0)                                   # few levels: some neighbourhoods have a constant response
    kw = dict(nlvdis=3, metric="mahal", h=2.0, k=25, nlv=3)
    a = J.predict(J.lwplsr(X[:n], y[:n], ctx=ctx, **kw), X[n:], nlv=range(0, 4), ctx=ctx)
    monkeypatch.setenv("JCH_LOCW_GENERIC", "1")
    b = J.predict(J.lwplsr(X[:n], y[:n], ctx=ctx, **kw), X[n:], nlv=range(0, 4), ctx=ctx)
    assert np.array_equal(a.listnn, b.listnn)
    assert O.rel_fro(np.stack(a.pred), np.stack(b.pred)) < 1e-8
    ref = O.lwplsr_predict(O.lwplsr(X[:n], y[:n], **kw), X[n:], nlv=range(0, 4))
    _cmp_lw(ref, b)


def test_lwplsr_many_lvs_and_many_responses(J, ctx):
    """nlv > 48 and q > 16 in the local fits (generic per-query path; q = 20 also exercises the generic small-state kernel)."""
    n, p, m, q = 600, 90, 4, 20
    X, A = _spectra(51, n + m, p, nsrc=60, noise=0.05)
    B = O.rand_matrix(54, p, q) - 0.5
    Y = X @ B + 0.05 * O.rand_matrix(55, n + m, q)
    kw = dict(nlvdis=8, metric="mahal", h=2.0, k=150, nlv=52)
    ref = O.lwplsr_predict(O.lwplsr(X[:n], Y[:n, 0], **kw), X[n:], nlv=[0, 10, 52])
    res = J.predict(J.lwplsr(X[:n], Y[:n, 0], ctx=ctx, **kw), X[n:], nlv=[0, 10, 52], ctx=ctx)
    assert np.array_equal(res.listnn, ref["listnn"]) and len(res.pred) == 53
    for a in (0, 5, 10, 20):                                      # (later LVs of a 150-row local model fit noise: conditioning)
        assert O.rel_fro(ref["pred"][:, :, a], res.pred[a]) < 1e-6, a
    kw = dict(nlvdis=8, metric="eucl", h=2.0, k=80, nlv=4)
    ref = O.lwplsr_predict(O.lwplsr(X[:n], Y[:n], **kw), X[n:], nlv=range(0, 5))
    res = J.predict(J.lwplsr(X[:n], Y[:n], ctx=ctx, **kw), X[n:], nlv=range(0, 5), ctx=ctx)
    _cmp_lw(ref, res)


def test_lwplsr_mahalanobis_on_raw_spectra_p120(J, ctx):
    """nlvdis = 0, metric = "mahal" with p > 64: the reference whitens the raw X (src/getknn.jl:37-49, src/lwplsr.jl:21); the
    p x p covariance comes from the tiled MFMA SYRK behind jch_weighted_cov."""
    n, p, m = 2000, 120, 6
    X = O.rand_matrix(61, n + m, p) + 0.3 * O.rand_matrix(62, n + m, 1)          # full-rank covariance
    y = X[:, :5] @ np.array([1.0, -2.0, 0.5, 3.0, -1.0]) + 0.05 * O.rand_matrix(63, n + m, 1)[:, 0]
    S = J.plsr._cov_uncorrected(np.asfortranarray(X[:n]), ctx)
    assert O.rel_fro(np.cov(X[:n].T, bias=True), S) < 1e-11
    kw = dict(nlvdis=0, metric="mahal", h=2.0, k=100, nlv=3)
    ref = O.lwplsr_predict(O.lwplsr(X[:n], y[:n], **kw), X[n:], nlv=range(0, 4))
    res = J.predict(J.lwplsr(X[:n], y[:n], ctx=ctx, **kw), X[n:], nlv=range(0, 4), ctx=ctx)
    _cmp_lw(ref, res)


@pytest.mark.parametrize("alg", ["kern", "nipals", "simp", "rosa", "wold"])
def test_q70_responses(alg, J, ctx):
    """q > 64: the generic small-state kernel keeps its q x q eigen-solver matrices in global memory; plsnipals / plswold take
    c_raw = Y'Dt from a column accumulation of their own (the NIPALS sweep has one lane per response up to 64)."""
    n, p, q, nlv = 700, 50, 70, 6
    X = O.rand_matrix(71, n, p)
    B = O.rand_matrix(72, p, q) - 0.5
    Y = X @ B + np.sin(3.0 * X[:, :1]) + 0.1 * O.rand_matrix(73, n, q)
    fn = dict(kern=J.plskern, nipals=J.plsnipals, simp=J.plssimp, rosa=J.plsrosa, wold=J.plswold)[alg]
    ofn = dict(kern=O.plskern, nipals=O.plsnipals, simp=O.plssimp, rosa=O.plsrosa, wold=O.plswold)[alg]
    fm, ref = fn(X, Y, nlv=nlv, ctx=ctx), ofn(X, Y, nlv=nlv)
    s = O.sign_align(ref.R, fm.R)
    for f in ("T", "P", "R", "C"):
        assert O.rel_fro(getattr(ref, f), getattr(fm, f) * s) < TOL, f
    assert O.rel_fro(O.predict(ref, X[:20], nlv=nlv), J.predict(fm, X[:20], nlv=nlv, ctx=ctx)) < 1e-8


def test_plsrda_with_70_classes(J, ctx):
    """The call the round-3 review named: plsrda (src/plsrda.jl:71-77) on more than 64 classes = plskern on a 70-column dummy
    table."""
    n, p, ncl = 1400, 30, 70
    X = O.rand_matrix(81, n, p)
    y = np.array([f"c{int(v):02d}" for v in np.floor(O.rand_matrix(82, n, 1)[:, 0] * ncl)])
    assert len(set(y)) == ncl
    Xq = O.rand_matrix(83, 40, p)
    ref = O.plsrda(X, y, nlv=5)
    rp, rpost = O.plsrda_predict(ref, Xq, nlv=5)
    fm = J.plsrda(X, y, nlv=5, ctx=ctx)
    gp, gpost = J.predict(fm, Xq, nlv=5, ctx=ctx)
    assert O.rel_fro(rpost, gpost) < 1e-8 and np.array_equal(rp, gp)


@pytest.mark.parametrize("alg", ["simp", "wold"])
def test_sibling_nlv_beyond_256_outside_the_lds_envelope(alg, J, ctx):
    """plssimp / plswold with nlv > 256 where their LDS-resident kernels do not apply (here p > 2048): the generic small-state
    kernel keeps its per-LV dot products in global memory.  Leading LVs against the oracle; for plswold (X deflated: the scores
    stay D-orthogonal) the invariant on the first 100 columns.  (SIMPLS itself loses orthogonality once X'Y is exhausted — the oracle's
    own scores are at 5e-5 by LV 50 and O(1) by LV 100 on this data — so there is no such invariant to assert for it.)"""
    n, p, q, nlv = 300, 2100, 2, 260
    X = O.rand_matrix(91, n, p)
    Y = X[:, :q] * 2.0 + O.rand_matrix(92, n, q)
    fn, ofn = (J.plssimp, O.plssimp) if alg == "simp" else (J.plswold, O.plswold)
    fm, ref = fn(X, Y, nlv=nlv, ctx=ctx), ofn(X, Y, nlv=nlv)
    assert fm.T.shape[1] == nlv and np.isfinite(fm.T).all() and np.isfinite(fm.P).all()
    key = "R" if alg == "simp" else "W"
    s = O.sign_align(getattr(ref, key)[:, :20], getattr(fm, key)[:, :20])
    # (plswold: R = W inv(P'W) involves ALL 260 LVs — its leading columns inherit the conditioning of the noise LVs in any
    # implementation —, so the per-LV quantities T, P, W, C are compared)
    for f in (("T", "P", "R", "C") if alg == "simp" else ("T", "P", "W", "C")):
        assert O.rel_fro(getattr(ref, f)[:, :20], getattr(fm, f)[:, :20] * s) < TOL, f
    if alg == "wold":   # (beyond ~LV 100 the Y residual of this data is rounding noise and NIPALS degenerates — in the oracle as well: 2e-3 by LV 150)
        d = fm.weights
        G = (fm.T[:, :100] * d[:, None]).T @ fm.T[:, :100]
        assert np.abs(G - np.diag(np.diag(G))).max() < 1e-10 * np.abs(np.diag(G)).max()


def test_vip_and_plslda_beyond_64_columns(J, ctx):
    """vip(object, Y) with q + nlv > 64 and plslda with more than 64 LVs go through the wide jch_weighted_cov."""
    n, p, q, nlv = 500, 80, 60, 8
    X = O.rand_matrix(101, n, p)
    Y = X @ (O.rand_matrix(102, p, q) - 0.5) + 0.1 * O.rand_matrix(103, n, q)
    fm, ref = J.plskern(X, Y, nlv=nlv, ctx=ctx), O.plskern(X, Y, nlv=nlv)
    got, exp = J.vip(fm, Y, ctx=ctx), O.vip(ref, Y)
    assert O.rel_fro(exp["imp"], got["imp"]) < 1e-8


def test_lwplsr_kspace_outlying_query_is_refitted(J, ctx, monkeypatch):
    """the neighbour-space local-fit kernel forms the Gram matrix of the gathered rows about the QUERY row.  Neighbours
    are chosen in the nlvdis-dimensional score space, so a query may sit ~1e4 spreads away from them in full p-space (here: shifted
    along a direction the global scores do not see); the centring then comes out of the Gram matrix by cancellation.  The kernel's
    pivot check flags such queries and the library refits them with the per-query path: predictions equal the oracle's, the ordinary
    queries of the same call stay on the batched kernel, and the refit counter says how many were redone."""
    import ctypes as C
    n, p, m = 3000, 150, 6
    X, A = _spectra(111, n + m, p, nsrc=5, noise=0.05)
    y = A[:, 0] - A[:, 1] + 0.3 * A[:, 2]
    kw = dict(nlvdis=4, metric="eucl", h=2.0, k=160, nlv=5)
    ofm = O.lwplsr(X[:n], y[:n], **kw)
    # a direction invisible to the global scores: orthogonal to the columns of R (so the query keeps its neighbours), large in p-space
    R = ofm.fm.R
    v = O.rand_matrix(112, p, 1)[:, 0] - 0.5
    v -= R @ np.linalg.lstsq(R, v, rcond=None)[0]
    v /= np.linalg.norm(v)
    Xq = X[n:].copy()
    Xq[1] += 1e4 * X[:n].std() * v
    Xq[4] -= 3e3 * X[:n].std() * v
    ref = O.lwplsr_predict(ofm, Xq, nlv=range(0, 6))
    monkeypatch.setenv("JCH_LOCW_KSPACE", "2")                    # the neighbour-space kernel wherever the shape fits
    fm = J.lwplsr(X[:n], y[:n], ctx=ctx, **kw)
    cnt = C.c_int64(0)
    lib = J.load()
    ctx.check(lib.jch_ctx_get_counter(ctx._h, 1, C.byref(cnt))); before = cnt.value
    res = J.predict(fm, Xq, nlv=range(0, 6), ctx=ctx)
    ctx.check(lib.jch_ctx_get_counter(ctx._h, 1, C.byref(cnt)))
    assert cnt.value - before == 2                                # exactly the two shifted queries were refitted
    assert np.array_equal(res.listnn, ref["listnn"])
    pred = np.stack(res.pred, axis=2)
    for i in range(m):
        assert O.rel_fro(ref["pred"][i], pred[i]) < 1e-7, i
