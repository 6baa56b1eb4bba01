import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jchemo.jl_amd")]
import numpy as np
import jchemo_hip as J
from jchemo_hip import plsr as P
ctx = J.Context(0)
rng = np.random.default_rng(23)
n, p, m = 60000, 3, 45
X = rng.standard_normal((n, p)) * np.array([50.0, 0.1, 0.1])
X = X[np.argsort(X[:, 0])]
Xq = rng.standard_normal((m, p)) * np.array([50.0, 0.1, 0.1])
y = X[:, 0] + rng.standard_normal(n)
X1 = X / np.array([50.0, 0.1, 0.1]); Xq1 = Xq / np.array([50.0, 0.1, 0.1])
fm = P.lwplsr(X1, y, ctx=ctx, nlvdis=0, metric="eucl", h=2.0, k=200, nlv=2)
a = P.lwplsr_predict(fm, Xq1, nlv=range(0, 3), ctx=ctx)
print("screened", ctx.counter(2), "redone", ctx.counter(3))
d2 = ((X1[None, :, :] - Xq1[:, None, :]) ** 2).sum(axis=2)
kth = np.sort(d2, axis=1)[:, 199]
print("k-th d2 per query:", np.round(kth, 4))
print("query norms^2:", np.round((Xq1 ** 2).sum(axis=1), 2))
print("max row norm^2:", (X1 ** 2).sum(axis=1).max())
