import os, sys
sys.path.insert(0, "jchemo.jl_amd"); sys.path.insert(0, ".")
import numpy as np
from oracle import plsr_oracle as O
import jchemo_hip as J
ctx = J.Context(0)
n, p, q, nlv = 5000, 120, 3, 6
for offset in (0.0, 1e2, 1e4, 1e6):
    rng = np.random.default_rng(4)
    Lt = rng.standard_normal((n, 8))
    X = np.asfortranarray(0.3 * (Lt @ rng.standard_normal((8, p))) + 0.1 * rng.standard_normal((n, p)) + offset)
    Y = np.asfortranarray(Lt[:, :3] @ rng.standard_normal((3, q)) + 0.1 * rng.standard_normal((n, q)) + offset / 7)
    w = rng.uniform(0.5, 1.5, n)
    ref = O.plskern(X, Y, w, nlv=nlv)
    for mode in ("raw", "centred"):
        if mode == "centred": os.environ["JCH_CENTRED_COPY"] = "1"
        else: os.environ.pop("JCH_CENTRED_COPY", None)
        fm = J.plskern(X, Y, w, nlv=nlv, ctx=ctx)
        s = O.sign_align(ref.W, fm.W)
        print(offset, mode, {f: "%.1e" % O.rel_fro(getattr(ref, f), getattr(fm, f) * s) for f in ("T", "P", "R", "C")})
