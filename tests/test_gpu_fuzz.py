"""Seeded random-shape parity sweep (-m gpu): every fit entry point against the oracle on shapes that cross the kernel
variants (row widths 2..2300: KC = 1..16 and the wide fallback; q = 1..17: padded q, the raw-mode limit q <= 15, the
generic small-state kernel; odd n / p; nlv up to the rank; weights on/off; scal on/off)."""
import numpy as np
import pytest

from oracle import plsr_oracle as O

pytestmark = pytest.mark.gpu


def _cases(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        p = int(rng.choice([2, 3, 7, 31, 64, 65, 127, 128, 129, 200, 255, 257, 500, 513, 777, 1024, 1500, 2048, 2300]))
        n = int(rng.integers(max(8, min(p, 40)), 1500))
        q = int(rng.choice([1, 1, 2, 3, 4, 5, 8, 10, 15, 16, 17]))
        nlv = int(rng.integers(1, min(n - 1, p, 9) + 1))
        out.append((i, n, p, q, nlv, bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
    return out


@pytest.fixture(scope="module")
def J():
    import jchemo_hip
    return jchemo_hip


@pytest.fixture(scope="module")
def ctx(J):
    c = J.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("case", _cases(int(__import__("os").environ.get("JCH_FUZZ_SEED", "20250112")), int(__import__("os").environ.get("JCH_FUZZ_COUNT", "48"))), ids=lambda c: "n%d_p%d_q%d_a%d_%s%s" % (c[1], c[2], c[3], c[4], "s" if c[5] else "", "w" if c[6] else ""))
def test_random_shapes_all_algorithms(case, J, ctx):
    i, n, p, q, nlv, scal, weighted = case
    rng = np.random.default_rng(1000 + i)
    k = max(nlv + 2, 4)
    Lt = rng.standard_normal((n, k))
    X = np.asfortranarray(Lt @ rng.standard_normal((k, p)) * rng.uniform(0.5, 2.0, p) + 0.3 * rng.standard_normal((n, p)) + rng.uniform(-3, 3, p))
    Y = np.asfortranarray(Lt[:, :min(k, max(q, 2))] @ rng.standard_normal((min(k, max(q, 2)), q)) + 0.2 * rng.standard_normal((n, q)) + 1.0)
    w = rng.uniform(0.3, 1.7, n) if weighted else None
    for alg in ("plskern", "plsnipals", "plsrosa", "plssimp", "plswold"):
        ref = getattr(O, alg)(X, Y, w, nlv=nlv, scal=scal)
        fm = getattr(J, alg)(X, Y, w, nlv=nlv, scal=scal, ctx=ctx)
        s = O.sign_align(ref.R, fm.R)
        for f in ("T", "P", "R", "C"):
            e = O.rel_fro(getattr(ref, f), getattr(fm, f) * s)
            assert e < 1e-7, (alg, f, e)
        assert O.rel_fro(ref.TT, fm.TT) < 1e-7 and O.rel_fro(ref.xmeans, fm.xmeans) < 1e-10 and O.rel_fro(ref.xscales, fm.xscales) < 1e-10
        if alg == "plswold":
            assert np.array_equal(ref.niter, fm.niter)
    # accessors on fresh rows
    Xn = np.asfortranarray(X[: min(n, 37)] * 1.01)
    fk = J.plskern(X, Y, w, nlv=nlv, scal=scal, ctx=ctx)
    rk = O.plskern(X, Y, w, nlv=nlv, scal=scal)
    s = O.sign_align(rk.R, fk.R)
    assert O.rel_fro(O.transform(rk, Xn), J.transform(fk, Xn, ctx=ctx) * s) < 1e-7
    assert O.rel_fro(np.stack(O.predict(rk, Xn, nlv=range(0, nlv + 1))), np.stack(J.predict(fk, Xn, nlv=range(0, nlv + 1), ctx=ctx))) < 1e-7


def _range_cases(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        p = int(rng.choice([3, 17, 64, 65, 130, 257, 500]))
        m = int(rng.integers(4096, 9000))
        q = int(rng.choice([1, 2, 3, 5, 8, 10, 16, 17, 21]))
        nlv = int(rng.integers(3, min(p, 34) + 1))
        lo = int(rng.integers(0, nlv - 1))
        hi = int(rng.integers(lo + 2, nlv + 1)) if lo + 2 <= nlv else nlv
        out.append((i, m, p, q, nlv, lo, hi, bool(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("case", _range_cases(int(__import__("os").environ.get("JCH_FUZZ_SEED", "20250112")) + 7, 16),
                         ids=lambda c: "m%d_p%d_q%d_a%d_%d-%d%s" % (c[1], c[2], c[3], c[4], c[5], c[6], "m" if c[7] else ""))
def test_random_shapes_prediction_ranges_and_their_scores(case, J, ctx):
    """`predict(fm, X; nlv = lo:hi)` on long inputs (running sums over the score columns, csrc/gemm.hip k_predict_prefix) against
    numpy, and the per-level score statistics straight from the scores (jch_score_sums_lv) against numpy on those predictions, on
    seeded random shapes: odd and even m, one response ... more than a 16-response slice, ranges anywhere inside 0..nlv (more than
    32 levels included), with and without a row mask."""
    from jchemo_hip import plsr as PL
    i, m, p, q, nlv, lo, hi, masked = case
    rng = np.random.default_rng(5000 + i)
    k = min(nlv + 2, p)
    Lt = rng.standard_normal((1200, k))
    X = np.asfortranarray(Lt @ rng.standard_normal((k, p)) + 0.3 * rng.standard_normal((1200, p)) + rng.uniform(-2, 2, p))
    Y = np.asfortranarray(Lt[:, :min(k, q)] @ rng.standard_normal((min(k, q), q)) + 0.3 * rng.standard_normal((1200, q)) + 1.0)
    fm = J.plskern(X, Y, nlv=nlv, scal=bool(i & 1), ctx=ctx)
    a = fm.P.shape[1]
    hi = min(hi, a)
    lo = min(lo, hi)
    Xn = np.asfortranarray(rng.standard_normal((m, p)) + rng.uniform(-2, 2, p))
    Yn = np.asfortranarray(rng.standard_normal((m, q)) + 1.0)
    got = J.predict(fm, Xn, nlv=range(lo, hi + 1), ctx=ctx)
    got = got if isinstance(got, list) else [got]
    refs = []
    for lv in range(lo, hi + 1):
        B = (fm.R[:, :lv] @ fm.C[:, :lv].T) / fm.xscales[:, None] * fm.yscales[None, :]
        refs.append(fm.ymeans + (Xn - fm.xmeans) @ B)
        assert O.rel_fro(refs[-1], np.asarray(got[lv - lo])) < 1e-10, lv
    mask = (rng.random(m) < 0.4).astype(np.float64) if masked else None
    S = PL._score_sums_lv(J.transform(fm, Xn, nlv=hi, ctx=ctx) if hi > 0 else np.zeros((m, 0)), fm, Yn, mask, list(range(lo, hi + 1)), ctx) \
        if hi > 0 else None
    if S is not None:
        wv = np.ones(m) if mask is None else mask
        for li, pr in enumerate(refs):
            e = Yn - pr
            want = np.stack([(wv[:, None] * e).sum(0), (wv[:, None] * e * e).sum(0), (wv[:, None] * Yn * e).sum(0), (wv[:, None] * Yn).sum(0),
                             (wv[:, None] * Yn * Yn).sum(0), np.full(q, wv.sum())], axis=1)
            assert np.allclose(S[li], want, rtol=1e-9, atol=1e-8 * np.abs(want).max()), (li, np.abs(S[li] - want).max())
