"""Throughput of the accessors at cfg2 size on device-resident data:
transform (k = 25), predict for one nlv (k = q = 10), predict for nlv = 0..25 (k = 260), summary."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jchemo.jl_amd")]
import numpy as np, torch
import jchemo_hip as J
n, p, q, nlv = 1_000_000, 500, 10, 25
ctx = J.Context(0, stream="torch"); lib = J.load()
X = J.colmajor_empty(n, p); Y = J.colmajor_empty(n, q)
ctx.check(lib.jch_fill_uniform(ctx._h, X.data_ptr(), n, p, n, 0, n, 20250112))
ctx.check(lib.jch_fill_uniform(ctx._h, Y.data_ptr(), n, q, n, 0, n, 20250113))
fm = J.plskern(X, Y, nlv=nlv, ctx=ctx)
def timeit(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
out = {}
for name, f in (("transform_k25", lambda: J.transform(fm, X, ctx=ctx)), ("predict_nlv25_k10", lambda: J.predict(fm, X, nlv=25, ctx=ctx)),
                ("predict_range_0_25_k260", lambda: J.predict(fm, X, nlv=range(0, 26), ctx=ctx)), ("summary", lambda: J.summary(fm, X, ctx=ctx))):
    t = timeit(f)
    out[name] = {"ms": t * 1e3, "X_GBps": n * p * 8 / t / 1e9}
print(json.dumps(out))
