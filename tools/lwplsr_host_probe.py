"""Where the host-side time of a cfg5 predict call goes (wall vs the library call vs the device spans)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jchemo.jl_amd")]
import numpy as np, torch
import jchemo_hip as J
from jchemo_hip import plsr as P, _lib
n, p, m, k, nlvdis, nlv, r = 100000, 500, 1000, 200, 20, 15, 30
ctx = J.Context(0, stream="torch"); lib = J.load(); dev = torch.device("cuda", 0)
def gen(rows, seed):
    S = J.colmajor_empty(rows, r, dev); E = J.colmajor_empty(rows, p, dev); L = J.colmajor_empty(r, p, dev)
    ctx.check(lib.jch_fill_uniform(ctx._h, S.data_ptr(), rows, r, rows, 0, rows, seed))
    ctx.check(lib.jch_fill_uniform(ctx._h, E.data_ptr(), rows, p, rows, 0, rows, seed + 100))
    ctx.check(lib.jch_fill_uniform(ctx._h, L.data_ptr(), r, p, r, 0, r, 777))
    out = J.colmajor_empty(rows, p, dev); out.copy_(S @ L + 0.1 * E); return out
X = gen(n, 1); Xq = gen(m, 2)
y = J.colmajor_empty(n, 1, dev); y.copy_((X[:, :5].sum(1) + torch.sin(3 * X[:, 5])).reshape(-1, 1))
fm = J.lwplsr(X, y, nlvdis=nlvdis, metric="mahal", h=1.0, k=k, nlv=nlv, ctx=ctx)
orig = lib.jch_lwplsr_predict_prepared
acc = {"lib": 0.0}
class Wrap:
    def __call__(self, *a):
        t0 = time.perf_counter(); r_ = orig(*a); acc["lib"] += time.perf_counter() - t0; return r_
lib.jch_lwplsr_predict_prepared = Wrap()
for _ in range(3): J.predict(fm, Xq, nlv=range(0, nlv + 1), ctx=ctx)
acc["lib"] = 0.0; devms = 0.0; calls = 20
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(calls):
    res = J.predict(fm, Xq, nlv=range(0, nlv + 1), ctx=ctx); devms += ctx.profile().fit_ms
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / calls
print(f"wall per call {dt*1e3:.3f} ms; inside jch_lwplsr_predict_prepared {acc['lib']/calls*1e3:.3f} ms; device spans (kNN + local fits) {devms/calls:.3f} ms; Python around the call {(dt - acc['lib']/calls)*1e3:.3f} ms")
