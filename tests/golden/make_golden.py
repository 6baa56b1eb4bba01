"""Regenerates tests/golden/*.npz from the numpy oracle (oracle/plsr_oracle.py).

The reference holds no fixtures for this path (test/runtests.jl:1-2 is empty) and cannot be run
here (Julia source, no Julia toolchain), so these vectors are produced by the repo's own fp64
restatement AFTER it has passed tests/test_oracle.py (invariants, plskern==plsnipals, LAPACK SVD,
scikit-learn).  They pin the oracle and the HIP path against silent drift; they are NOT outputs of
the reference itself ("parity unpinned").

Inputs are not stored: they are regenerated from the portable splitmix64 generator
(seed, shape) — the file keeps a few generator values so a generator change is caught.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import plsr_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name: (n, p, q, nlv, scal, weighted, m_test)
CASES = {
    "cfg1": dict(n=150, p=200, q=2, nlv=5, scal=False, weighted=False, m=50),          # BASELINE.json configs[0]
    "cfg1_scal_w": dict(n=150, p=200, q=2, nlv=5, scal=True, weighted=True, m=50),
    "q1": dict(n=120, p=64, q=1, nlv=6, scal=False, weighted=False, m=10),             # q == 1 branch (plskern.jl:150-152)
    "ragged": dict(n=37, p=13, q=3, nlv=20, scal=True, weighted=True, m=5),            # nlv clamped to min(n,p)=13; odd p
    "wide_q": dict(n=300, p=70, q=10, nlv=12, scal=False, weighted=True, m=7),         # q = 10 as in cfg2
}
SEED_X, SEED_Y, SEED_T, SEED_W = 20250112, 20250113, 20250114, 20250116
SIB_NLV = 8   # LVs of the sibling-algorithm fixtures (the late LVs of `ragged` are rank-exhausted noise)


def inputs(c):
    X = O.rand_matrix(SEED_X, c["n"], c["p"])
    Y = O.rand_matrix(SEED_Y, c["n"], c["q"])
    Xt = O.rand_matrix(SEED_T, c["m"], c["p"])
    w = 0.25 + O.splitmix64_uniform(SEED_W, 0, c["n"]) if c["weighted"] else None
    return X, Y, Xt, w


def main():
    for name, c in CASES.items():
        X, Y, Xt, w = inputs(c)
        out = {"gen_probe": np.concatenate([X[:3, 0], X[0, :3], Y[:2, 0]])}
        for alg, fn in (("kern", O.plskern), ("nipals", O.plsnipals)):
            fm = fn(X, Y, w, nlv=c["nlv"], scal=c["scal"])
            k = fm.T.shape[1]
            for f in ("T", "P", "R", "W", "C", "TT", "xmeans", "xscales", "ymeans", "yscales", "weights"):
                out[f"{alg}_{f}"] = getattr(fm, f)
            out[f"{alg}_transform"] = O.transform(fm, Xt)
            out[f"{alg}_B"] = np.stack([O.coef(fm, nlv=a)[0] for a in range(k + 1)])
            out[f"{alg}_int"] = np.stack([O.coef(fm, nlv=a)[1] for a in range(k + 1)])
            out[f"{alg}_pred"] = np.stack(O.predict(fm, Xt, nlv=range(0, k + 1)))
            s = O.summary(fm, X)
            out[f"{alg}_summary"] = np.stack([s["var"], s["pvar"], s["cumpvar"]])
        # in-place results of the `!` variants (centred/scaled X, Y; deflated for nipals)
        Xk, Yk = X.copy(), Y.copy(); O.plskern_(Xk, Yk, w, nlv=c["nlv"], scal=c["scal"])
        Xn, Yn = X.copy(), Y.copy(); O.plsnipals_(Xn, Yn, w, nlv=c["nlv"], scal=c["scal"])
        out["kern_Xinplace_probe"] = Xk[:4, :4].copy(); out["kern_Yinplace"] = Yk
        out["nipals_Xinplace_probe"] = Xn[:4, :4].copy(); out["nipals_Yinplace"] = Yn
        out["nipals_Xinplace_fro"] = np.array([np.linalg.norm(Xn)])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in list(out.items())[:4]}, "...")
        # sibling algorithms, separate file so the fixtures above stay byte-identical
        sib = {}
        ks = min(c["nlv"], SIB_NLV)
        for alg, fn in (("simp", O.plssimp), ("rosa", O.plsrosa), ("wold", O.plswold)):
            fm = fn(X, Y, w, nlv=ks, scal=c["scal"])
            for f in ("T", "P", "R", "W", "C", "TT"):
                sib[f"{alg}_{f}"] = getattr(fm, f)
            if fm.niter is not None:
                sib[f"{alg}_niter"] = fm.niter
        Xr_, Yr_ = X.copy(), Y.copy(); O.plsrosa_(Xr_, Yr_, w, nlv=ks, scal=c["scal"])
        Xw_, Yw_ = X.copy(), Y.copy(); O.plswold_(Xw_, Yw_, w, nlv=ks, scal=c["scal"])
        sib["rosa_Yinplace"] = Yr_; sib["wold_Yinplace"] = Yw_; sib["wold_Xinplace_fro"] = np.array([np.linalg.norm(Xw_)])
        np.savez_compressed(os.path.join(HERE, name + "_siblings.npz"), **sib)


if __name__ == "__main__":
    main()
