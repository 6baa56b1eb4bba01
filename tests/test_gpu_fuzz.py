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
