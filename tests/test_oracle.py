"""Pins the oracle (CPU, no GPU): the reference ships no fixtures (test/runtests.jl:1-2), so the
numpy and C restatements are held to (i) PLS invariants, (ii) plskern == plsnipals, (iii) LAPACK /
scikit-learn, (iv) the committed golden vectors.  SURVEY.md §4."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import c_oracle as CO
from oracle import plsr_oracle as O

FIELDS = ("T", "P", "R", "W", "C")


def _aligned_err(a, b):
    s = O.sign_align(a.W, b.W)
    return max(O.rel_fro(getattr(a, f), getattr(b, f) * s) for f in FIELDS)


def test_generator_known_answers():
    # splitmix64 known-answer: first outputs for seed 0 (Vigna's reference sequence) mapped to [0,1)
    z = [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]
    want = np.array([(v >> 11) / 2.0**53 for v in z])
    assert np.array_equal(O.splitmix64_uniform(0, 0, 3), want)
    a = O.rand_matrix(5, 17, 4, row0=3, n_total=40)
    full = O.rand_matrix(5, 40, 4)
    assert np.array_equal(a, full[3:20])
    assert np.array_equal(CO.fill_uniform(5, 17, 4, row0=3, n_total=40), a)


@pytest.mark.parametrize("name", ["cfg1", "cfg1_scal_w", "q1", "ragged", "wide_q"])
def test_invariants_and_cross_agreement(name, golden_cases):
    c = golden_cases.CASES[name]
    X, Y, Xt, w = golden_cases.inputs(c)
    fk = O.plskern(X, Y, w, nlv=c["nlv"], scal=c["scal"])
    fn = O.plsnipals(X, Y, w, nlv=c["nlv"], scal=c["scal"])
    k = fk.T.shape[1]
    assert k == min(c["n"], c["p"], c["nlv"])
    D = np.diag(fk.weights)
    Xs = (X - fk.xmeans) / fk.xscales
    Ys = (Y - fk.ymeans) / fk.yscales
    assert abs(fk.weights.sum() - 1) < 1e-14
    assert np.abs(fk.T.T @ D @ fk.T - np.diag(fk.TT)).max() < 1e-12 * fk.TT.max()
    assert np.abs(fk.R.T @ fk.P - np.eye(k)).max() < 1e-10
    assert O.rel_fro(fk.T, Xs @ fk.R) < 1e-12
    assert np.abs(np.linalg.norm(fk.W, axis=0) - 1).max() < 1e-13
    assert O.rel_fro(fk.P, Xs.T @ D @ fk.T / fk.TT) < 1e-12
    assert O.rel_fro(fk.C, Ys.T @ D @ fk.T / fk.TT) < 1e-10
    B0, i0 = O.coef(fk, nlv=0)
    assert np.all(B0 == 0) and np.allclose(i0, fk.ymeans[None, :])
    cum = O.summary(fk, X)["cumpvar"]
    assert np.all(np.diff(cum) >= -1e-14) and cum[-1] <= 1 + 1e-12
    # weight rescaling invariance
    if w is not None:
        f2 = O.plskern(X, Y, 3.7 * w, nlv=c["nlv"], scal=c["scal"])
        assert _aligned_err(fk, f2) < 1e-12
    # two independent algorithms agree (src/plskern.jl:73-76 presents them as interchangeable)
    assert _aligned_err(fk, fn) < 1e-9
    assert O.rel_fro(fk.TT, fn.TT) < 1e-10


@pytest.mark.parametrize("name", ["cfg1", "wide_q"])
def test_sklearn_and_lapack(name, golden_cases):
    from sklearn.cross_decomposition import PLSRegression
    c = golden_cases.CASES[name]
    X, Y, Xt, w = golden_cases.inputs(c)
    fk = O.plskern(X, Y, nlv=c["nlv"])            # unweighted: sklearn has no row weights
    sk = PLSRegression(n_components=c["nlv"], scale=False, tol=1e-14, max_iter=100000).fit(X, Y)
    B, intercept = O.coef(fk)
    coef = sk.coef_.T if sk.coef_.shape == (c["q"], c["p"]) else sk.coef_
    assert O.rel_fro(B, coef) < 1e-5
    assert O.rel_fro(O.predict(fk, Xt), sk.predict(Xt)) < 1e-6
    # Gram-eigen route (what the HIP small-state kernel does) == LAPACK dgesdd U[:,0]
    K = (X - X.mean(0)).T @ (Y - Y.mean(0)) / c["n"]
    u = O.dominant_left_sv(K)
    lam, V = np.linalg.eigh(K.T @ K)
    w2 = K @ V[:, -1]
    w2 /= np.linalg.norm(w2)
    assert min(np.linalg.norm(u - w2), np.linalg.norm(u + w2)) < 1e-12


@pytest.mark.parametrize("name", ["cfg1", "cfg1_scal_w", "q1", "ragged", "wide_q"])
@pytest.mark.parametrize("alg", ["kern", "nipals"])
def test_golden_vectors(name, alg, golden_cases):
    g = load_golden(name)
    c = golden_cases.CASES[name]
    X, Y, Xt, w = golden_cases.inputs(c)
    assert np.array_equal(g["gen_probe"], np.concatenate([X[:3, 0], X[0, :3], Y[:2, 0]]))
    for impl in (O, CO):
        fn = getattr(impl, "plskern" if alg == "kern" else "plsnipals")
        fm = fn(X, Y, w, nlv=c["nlv"], scal=c["scal"])
        s = O.sign_align(g[f"{alg}_W"], fm.W)
        for f in FIELDS:
            assert O.rel_fro(g[f"{alg}_{f}"], getattr(fm, f) * s) < 1e-10, (impl.__name__, f)
        for f in ("TT", "xmeans", "xscales", "ymeans", "yscales", "weights"):
            assert O.rel_fro(g[f"{alg}_{f}"], getattr(fm, f)) < 1e-12, (impl.__name__, f)
        k = fm.T.shape[1]
        assert O.rel_fro(g[f"{alg}_transform"], O.transform(fm, Xt) * s) < 1e-10
        assert O.rel_fro(g[f"{alg}_pred"], np.stack(O.predict(fm, Xt, nlv=range(0, k + 1)))) < 1e-10
        sm = O.summary(fm, X)
        assert O.rel_fro(g[f"{alg}_summary"], np.stack([sm["var"], sm["pvar"], sm["cumpvar"]])) < 1e-10


def test_inplace_semantics(golden_cases):
    """`plskern!` / `plsnipals!` overwrite X, Y (src/plskern.jl:122-130, src/plsnipals.jl:86-87)."""
    g = load_golden("cfg1_scal_w")
    c = golden_cases.CASES["cfg1_scal_w"]
    X, Y, Xt, w = golden_cases.inputs(c)
    for impl in (O, CO):
        Xk, Yk = np.asfortranarray(X.copy()), np.asfortranarray(Y.copy())
        impl.plskern_(Xk, Yk, w, nlv=c["nlv"], scal=c["scal"])
        assert np.allclose(Xk[:4, :4], g["kern_Xinplace_probe"], rtol=1e-12, atol=1e-14)
        assert np.allclose(Yk, g["kern_Yinplace"], rtol=1e-12, atol=1e-14)
        Xn, Yn = np.asfortranarray(X.copy()), np.asfortranarray(Y.copy())
        impl.plsnipals_(Xn, Yn, w, nlv=c["nlv"], scal=c["scal"])
        assert np.allclose(Xn[:4, :4], g["nipals_Xinplace_probe"], rtol=1e-9, atol=1e-12)
        assert abs(np.linalg.norm(Xn) - g["nipals_Xinplace_fro"][0]) < 1e-9 * g["nipals_Xinplace_fro"][0]


def test_predict_range_and_clamps():
    """src/plskern.jl:226-238: collection -> contiguous clamped range; one value -> matrix."""
    X = O.rand_matrix(1, 40, 9); Y = O.rand_matrix(2, 40, 2); Xt = O.rand_matrix(3, 5, 9)
    fm = O.plskern(X, Y, nlv=4)
    assert isinstance(O.predict(fm, Xt), np.ndarray)
    assert len(O.predict(fm, Xt, nlv=[1, 3])) == 3                  # expanded to 1:3
    assert len(O.predict(fm, Xt, nlv=range(-2, 99))) == 5           # clamped to 0:4
    assert O.transform(fm, Xt, nlv=99).shape == (5, 4)
    y1 = O.plskern(X, Y[:, 0], nlv=3)                               # vector y -> n x 1 (ensure_mat)
    assert y1.C.shape == (1, 3)


def test_degenerate_rank_propagates_nan():
    """H9: no guard in the reference — when Y is exhausted, tt -> 0 and NaN/Inf propagate."""
    X = O.rand_matrix(1, 30, 6)
    Y = np.zeros((30, 1))
    with np.errstate(all="ignore"):
        fm = O.plskern(X, Y, nlv=2)
    assert not np.all(np.isfinite(fm.C))


def test_sharded_equals_unsharded():
    X = O.rand_matrix(1, 101, 23); Y = O.rand_matrix(2, 101, 4); w = 0.5 + O.splitmix64_uniform(9, 0, 101)
    ref = O.plskern(X, Y, w, nlv=7, scal=True)
    cuts = [0, 25, 50, 76, 101]
    sh = O.plskern_sharded([X[a:b] for a, b in zip(cuts, cuts[1:])], [Y[a:b] for a, b in zip(cuts, cuts[1:])],
                           [w[a:b] for a, b in zip(cuts, cuts[1:])], nlv=7, scal=True)
    s = O.sign_align(ref.W, sh.W)
    assert O.rel_fro(ref.T, np.vstack(sh.T) * s) < 1e-11
    for f in ("P", "R", "W", "C"):
        assert O.rel_fro(getattr(ref, f), getattr(sh, f) * s) < 1e-11


def test_lwplsr_oracle_pieces():
    """Pins the kNN-LWPLSR restatement (src/getknn.jl, src/wdist.jl, src/locwlv.jl) against third-party code."""
    from scipy.spatial.distance import cdist
    from sklearn.neighbors import NearestNeighbors
    Xt = O.rand_matrix(1, 300, 6); Xq = O.rand_matrix(2, 9, 6)
    ind, d = O.getknn(Xt, Xq, k=7, metric="eucl")
    nn = NearestNeighbors(n_neighbors=7, algorithm="brute").fit(Xt)
    d2, i2 = nn.kneighbors(Xq)
    assert np.array_equal(ind, i2) and np.allclose(d, d2, rtol=1e-12)
    ind, d = O.getknn(Xt, Xq, k=7, metric="mahal")
    VI = np.linalg.inv(np.cov(Xt, rowvar=False, bias=True))
    D = cdist(Xq, Xt, "mahalanobis", VI=VI)
    assert np.array_equal(ind, np.argsort(D, axis=1)[:, :7]) and np.allclose(d, np.sort(D, axis=1)[:, :7], rtol=1e-9)
    assert O.getknn(Xt, Xq, k=10_000)[0].shape == (9, 300)                       # k clamped to n (getknn.jl:33)
    # wdist: max weight 1 at the smallest distance, zero beyond median + 4 MAD, NaN -> 1 for identical distances
    dd = np.array([0.1, 0.2, 0.25, 0.3, 5.0])
    w = O.wdist(dd, h=2.0)
    assert w[0] == 1.0 and w[-1] == 0.0 and np.all(np.diff(w) <= 0)
    assert np.allclose(w[:4], np.exp(-dd[:4] / (2.0 * O.mad(dd))) / np.exp(-dd[0] / (2.0 * O.mad(dd))))
    assert np.all(O.wdist(np.full(5, 0.3)) == 1.0)
    # locwlv: the prediction for nlv = 0 is the weighted neighbour mean; more LVs reduce the training residual
    X = O.rand_matrix(3, 200, 8); y = X[:, 0] - 2 * X[:, 1] + 0.01 * O.rand_matrix(4, 200, 1)[:, 0]
    fm = O.lwplsr(X, y, nlvdis=0, metric="eucl", h=2.0, k=30, nlv=4)
    r = O.lwplsr_predict(fm, X[:20], nlv=range(0, 5))
    w0 = r["listw"][0] / r["listw"][0].sum()
    assert abs(r["pred"][0, 0, 0] - w0 @ y[r["listnn"][0]]) < 1e-12
    err = [np.abs(r["pred"][:, 0, a] - y[:20]).mean() for a in range(5)]
    assert err[4] < err[0] * 0.2


def test_scores_and_grid_oracle():
    """Pins the §8f restatements (src/scores.jl, src/gridscore.jl:167-221, src/gridcv.jl:187-228)."""
    from sklearn.metrics import mean_squared_error, r2_score
    X = O.rand_matrix(1, 300, 12); B = O.rand_matrix(2, 12, 2) - 0.5
    Y = X @ B + 0.1 * O.rand_matrix(3, 300, 2)
    Xt = O.rand_matrix(4, 80, 12); Yt = Xt @ B + 0.1 * O.rand_matrix(5, 80, 2)
    fm = O.plskern(X, Y, nlv=5)
    pred = O.predict(fm, Xt)
    assert np.allclose(O.msep(pred, Yt)[0], mean_squared_error(Yt, pred, multioutput="raw_values"))
    assert np.allclose(O.r2(pred, Yt)[0], r2_score(Yt, pred, multioutput="raw_values"))
    assert np.allclose(O.rmsep(pred, Yt) ** 2, O.ssr(pred, Yt) / 80)
    assert np.allclose(O.bias(pred, Yt)[0], (pred - Yt).mean(axis=0))
    rng, res = O.gridscorelv(X, Y, Xt, Yt, score=O.rmsep, fun=O.plskern, nlv=range(-3, 99))
    assert rng == list(range(0, 13)) and res.shape == (13, 2)                    # clamped to 0:p
    assert np.allclose(res[5], O.rmsep(pred, Yt)[0])
    assert np.allclose(res[0], O.rmsep(np.tile(Y.mean(0), (80, 1)), Yt)[0])      # nlv = 0 predicts the training mean
    segm = [[np.arange(0, 300, 3), np.arange(1, 300, 3), np.arange(2, 300, 3)]]
    rng, cv, rep = O.gridcvlv(X, Y, segm=segm, score=O.msep, fun=O.plskern, nlv=range(0, 6))
    assert rep.shape == (1, 3, 6, 2) and np.allclose(cv, rep.mean(axis=(0, 1)))
    s = segm[0][1]
    _, one = O.gridscorelv(O.rmrow(X, s), O.rmrow(Y, s), X[s], Y[s], score=O.msep, fun=O.plskern, nlv=range(0, 6))
    assert np.allclose(rep[0, 1], one)


# ------------------------------------------------------------------ sibling algorithms
@pytest.mark.parametrize("name", ["cfg1", "cfg1_scal_w", "q1", "ragged", "wide_q"])
def test_sibling_oracles(name, golden_cases):
    """plsrosa == plskern and a fully converged plswold == plsnipals (all four are the same PLS2 up to rounding /
    inner convergence); plssimp: invariants, and the PLS1 identity for q == 1.  Fixtures pin the committed values."""
    c = golden_cases.CASES[name]
    X, Y, Xt, w = golden_cases.inputs(c)
    ks = min(c["nlv"], golden_cases.SIB_NLV)
    fk = O.plskern(X, Y, w, nlv=ks, scal=c["scal"])
    fr = O.plsrosa(X, Y, w, nlv=ks, scal=c["scal"])
    fs = O.plssimp(X, Y, w, nlv=ks, scal=c["scal"])
    fw = O.plswold(X, Y, w, nlv=ks, scal=c["scal"])
    g = load_golden(name + "_siblings")
    for alg, fm in (("simp", fs), ("rosa", fr), ("wold", fw)):
        for f in ("T", "P", "R", "W", "C", "TT"):
            assert O.rel_fro(g[f"{alg}_{f}"], getattr(fm, f)) < 1e-9, (alg, f)
    assert np.array_equal(g["wold_niter"], fw.niter)
    assert _aligned_err(fk, fr) < 1e-9
    if name != "ragged":      # (its first LV needs > 20000 passes: two nearly equal singular values)
        fwc = O.plswold(X, Y, w, nlv=ks, scal=c["scal"], tol=1e-30, maxit=20000)
        assert _aligned_err(O.plsnipals(X, Y, w, nlv=ks, scal=c["scal"]), fwc) < 1e-6
    # SIMPLS invariants: D-orthogonal scores, T = Xs R, unit r, P'R = I, W == R
    D = np.diag(fs.weights)
    Xs = (X - fs.xmeans) / fs.xscales
    assert np.abs(fs.T.T @ D @ fs.T - np.diag(fs.TT)).max() < 1e-11 * fs.TT.max()
    assert O.rel_fro(fs.T, Xs @ fs.R) < 1e-12
    assert np.abs(np.linalg.norm(fs.R, axis=0) - 1).max() < 1e-12
    assert np.array_equal(fs.W, fs.R)
    if c["q"] == 1:           # PLS1: SIMPLS and the kernel algorithm give the same regression
        for a in range(ks + 1):
            assert O.rel_fro(O.coef(fk, nlv=a)[0], O.coef(fs, nlv=a)[0]) < 1e-8 or a == 0
    assert fw.niter.min() >= 1 and fw.niter.max() <= 200
    # maxit caps the inner loop (src/plswold.jl:89)
    assert np.all(O.plswold(X, Y, w, nlv=2, scal=c["scal"], maxit=3).niter <= 3)


def test_mpar_and_pars_grids():
    """src/mpar.jl:15-24 (first keyword fastest) and the `pars` branch of gridscorelv / gridcvlv
    (src/gridscore.jl:191-216, src/gridcv.jl:206-224): element-wise combinations, combination-major rows."""
    pars = O.mpar(scal=[False, True], k=[3, 4, 5])
    assert pars == {"scal": [False, True] * 3, "k": [3, 3, 4, 4, 5, 5]}
    assert O.mpar(a=7, b="m") == {"a": [7], "b": ["m"]}
    X = O.rand_matrix(1, 60, 12); Y = O.rand_matrix(2, 60, 2); Xt = O.rand_matrix(3, 9, 12); Yt = O.rand_matrix(4, 9, 2)
    pars = O.mpar(scal=[False, True])
    rng, res = O.gridscorelv(X, Y, Xt, Yt, score=O.rmsep, fun=O.plskern, nlv=range(0, 4), pars=pars)
    assert res.shape == (2 * 4, 2)
    for i, sc in enumerate(pars["scal"]):
        _, one = O.gridscorelv(X, Y, Xt, Yt, score=O.rmsep, fun=lambda a, b, nlv: O.plskern(a, b, nlv=nlv, scal=sc), nlv=range(0, 4))
        assert np.array_equal(res[4 * i:4 * i + 4], one)
    segm = [[np.arange(0, 20), np.arange(20, 60)]]
    _, cv, rep = O.gridcvlv(X, Y, segm=segm, score=O.msep, fun=O.plskern, nlv=range(0, 4), pars=pars)
    assert cv.shape == (8, 2) and rep.shape == (1, 2, 8, 2) and np.allclose(cv, rep.mean(axis=(0, 1)))


def test_vip_xfit_xresid_oracle(golden_cases):
    """src/vip.jl:62-107, src/xfit.jl:37-93: identities that pin the restatements."""
    c = golden_cases.CASES["ragged"]
    X, Y, Xt, w = golden_cases.inputs(c)
    fm = O.plskern(X, Y, w, nlv=c["nlv"], scal=True)          # nlv = p = 13: full rank -> exact reconstruction
    assert O.rel_fro(X, O.xfit(fm, X)) < 1e-9
    assert np.allclose(O.xfit(fm, Xt, nlv=0), np.tile(fm.xmeans, (Xt.shape[0], 1)))
    for k in (1, 4):
        assert O.rel_fro(Xt, O.xfit(fm, Xt, nlv=k) + O.xresid(fm, Xt, nlv=k)) < 1e-14
    v = O.vip(fm, nlv=5)
    assert v["imp"].shape == (13,) and abs(np.mean(v["imp"] ** 2) - 1) < 1e-12      # VIP^2 average to 1 (unit-norm W)
    vy = O.vip(fm, Y, nlv=5)
    assert abs(np.mean(vy["imp"] ** 2) - 1) < 1e-12 and vy["rdd"].shape == (1, 5) and np.all(vy["rdd"] <= 1 + 1e-12)
    y1 = Y[:, :1]
    f1 = O.plskern(X, y1, w, nlv=4)
    # univariate y: rd(y, t_a) = cor(y, t_a)^2 is proportional to c_a^2 tt_a -> both VIP definitions coincide
    assert O.rel_fro(O.vip(f1)["imp"], O.vip(f1, y1)["imp"]) < 1e-10


def test_plslda_plsqda_oracle():
    """src/lda.jl, src/qda.jl, src/matW.jl, src/dmnorm.jl restatements: QDA posteriors equal scikit-learn's; LDA equals
    scikit-learn's once its pooled covariance gets the reference's n / (n - nlev) factor (src/lda.jl:65)."""
    from sklearn.discriminant_analysis import QuadraticDiscriminantAnalysis
    rng = np.random.default_rng(0)
    n, p = 300, 20
    y = rng.integers(0, 3, n)
    X = rng.standard_normal((n, p)) + y[:, None] * np.linspace(0, 1, p)[None, :]
    m = O.plsqda(X, y, nlv=4, prior="prop")
    T = m["fm_pls"].T
    sk = QuadraticDiscriminantAnalysis(priors=np.bincount(y) / n).fit(T, y)
    assert np.abs(sk.predict_proba(T) - O.da_predict(m["fm_da"][3], T)[2]).max() < 1e-12
    ml = O.plslda(X, y, nlv=4, prior="unif")
    W, Wi, lev, ni = O.matW(T, y)
    assert np.allclose(W, sum((ni[i] / n) * np.cov(T[y == lev[i]], rowvar=False, bias=True) for i in range(3)))
    mu, Uinv, detS = ml["fm_da"][3]["fm"][1]
    assert np.allclose(np.linalg.inv(Uinv @ Uinv.T), W * n / (n - 3)) and abs(detS - np.linalg.det(W * n / (n - 3))) < 1e-12 * detS
    pr, po = O.plslda_predict(ml, X[:40], nlv=range(1, 99))
    assert len(pr) == 4 and all(np.allclose(q_.sum(axis=1), 1) for q_ in po)          # nlv clamped to the model's 4
    with pytest.raises(ValueError):
        O.plslda_predict(ml, X[:40], nlv=range(0, 3))                                  # fm_da[0]: BoundsError in the reference
