import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jchemo.jl_amd")]
import numpy as np
import jchemo_hip as J
from jchemo_hip import plsr as P
which = sys.argv[1]
ctx = J.Context(0)
rng = np.random.default_rng(17)
n, p, m = 8000, 10, 37
X = rng.standard_normal((n, p)); Xq = rng.standard_normal((m, p))
y = X[:, 0] - X[:, 1] + 0.1 * rng.standard_normal(n)
if which == "nan": Xq[3, 2] = np.nan
if which == "inf": Xq[20, :] = np.inf
if which == "inf1": Xq[20, 1] = np.inf
if which == "train": X[77, 4] = np.nan
fm = P.lwplsr(X, y, ctx=ctx, nlvdis=0, metric="eucl", h=2.0, k=30, nlv=2)
print(which, "fit ok", flush=True)
a = P.lwplsr_predict(fm, Xq, nlv=range(0, 3), ctx=ctx)
print(which, "predict ok", ctx.counter(2), ctx.counter(3), flush=True)
