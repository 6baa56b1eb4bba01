"""Latency of small fits (launch-bound regime): cfg1 (n=150, p=200, q=2, nlv=5) and a CV-fold-sized fit, host and
device-resident inputs.  Prints ms per fit."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jchemo.jl_amd"))
import numpy as np
import torch
import jchemo_hip as J

ctx = J.Context(0)
out = {}
for name, (n, p, q, nlv) in {"cfg1": (150, 200, 2, 5), "n2000_p500_q10_nlv25": (2000, 500, 10, 25)}.items():
    rng = np.random.default_rng(0)
    X = np.asfortranarray(rng.random((n, p))); Y = np.asfortranarray(rng.random((n, q)))
    Xd = J.colmajor_empty(n, p); Xd.copy_(torch.from_numpy(X)); Yd = J.colmajor_empty(n, q); Yd.copy_(torch.from_numpy(Y))
    torch.cuda.synchronize()
    for tag, (a, b) in {"host": (X, Y), "device": (Xd, Yd)}.items():
        for _ in range(5):
            J.plskern(a, b, nlv=nlv, ctx=ctx)
        t0 = time.perf_counter()
        for _ in range(50):
            J.plskern(a, b, nlv=nlv, ctx=ctx)
        out[f"{name}_{tag}_ms"] = (time.perf_counter() - t0) / 50 * 1e3
print(json.dumps(out))
