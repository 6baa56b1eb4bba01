"""T'DT = diag(TT) drift of the split small-state test's shapes, first sweep walked forwards / backwards (JCH_SWEEP_FIRST_REV is read once per process:
run this script once per mode)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jchemo.jl_amd")); sys.path.insert(0, ROOT)
import jchemo_hip as J
from oracle import c_oracle as CO
ctx = J.Context(0)
shapes = [(3000, 500, 10, 25), (900, 37, 1, 12), (1200, 130, 2, 9), (2000, 257, 3, 20), (2500, 64, 7, 30), (1500, 1000, 16, 15), (700, 16, 5, 16), (400, 301, 12, 40)]
for variant in ("raw", "scal_w", "centred", "rosa"):
    if variant == "centred": os.environ["JCH_CENTRED_COPY"] = "1"
    else: os.environ.pop("JCH_CENTRED_COPY", None)
    row = []
    for (n, p, q, nlv) in shapes:
        X = CO.fill_uniform(401, n, p) + 3.0; B0 = CO.fill_uniform(402, p, q) - 0.5; Y = X @ B0 + 0.1 * CO.fill_uniform(403, n, q)
        w = CO.fill_uniform(404, n, 1)[:, 0] + 0.2 if variant == "scal_w" else None
        fn = J.plsrosa if variant == "rosa" else J.plskern
        fm = fn(X, Y, w, nlv=nlv, scal=variant == "scal_w", ctx=ctx)
        G = (fm.T * fm.weights[:, None]).T @ fm.T
        row.append(np.abs(G - np.diag(fm.TT)).max() / np.abs(fm.TT).max())
    print(f"FIRST_REV={os.environ.get('JCH_SWEEP_FIRST_REV', '1')} {variant:8s}", " ".join(f"{v:.2e}" for v in row))
