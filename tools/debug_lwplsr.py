import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jchemo.jl_amd")]
import numpy as np, torch
import jchemo_hip as J
from oracle import plsr_oracle as O, c_oracle as CO
n, p, m, k, nlvdis, nlv = [int(v) for v in os.environ.get("LW_CASE", "100000,500,64,200,20,15").split(",")]
metric = os.environ.get("LW_METRIC", "mahal")
ctx = J.Context(0)
X = CO.fill_uniform(20250112, n, p); Xq = CO.fill_uniform(20250115, m, p)
y = (X[:, :5] @ np.array([1.0, -2.0, 0.5, 3.0, 1.5]) + np.sin(3 * X[:, 5]) + 0.05 * CO.fill_uniform(20250113, n, 1)[:, 0]).reshape(-1, 1)
ofm = CO.plskern(X, y, nlv=nlvdis)
gfm = J.plskern(X, y, nlv=nlvdis, ctx=ctx)
s = O.sign_align(ofm.W, gfm.W)
print("global fit T err per LV:", [f"{O.rel_fro(ofm.T[:, a], gfm.T[:, a] * s[a]):.1e}" for a in range(nlvdis)])
print("TT", ofm.TT[:8], ofm.TT[-3:])
# kNN given identical Z (oracle's whitened scores)
olw = O.Lwplsr(X, y, ofm, metric, 1.0, k, nlv, 1e-4, False)
ref = O.lwplsr_predict(olw, Xq, nlv=range(0, nlv + 1))
S = np.cov(ofm.T, rowvar=False, bias=True).reshape(nlvdis, nlvdis); Uinv = np.linalg.inv(np.linalg.cholesky(S).T) if metric == "mahal" else np.eye(nlvdis)
Zt = np.asfortranarray(ofm.T @ Uinv); Zq = np.asfortranarray(O.transform(ofm, Xq) @ Uinv)
import ctypes as C
pred = np.empty((m, nlv + 1)); ind = np.empty((m, k), dtype=np.int32); dist = np.empty((m, k)); w = np.empty((m, k))
Xf = np.asfortranarray(X); Xqf = np.asfortranarray(Xq)
ctx.check(J.load().jch_lwplsr_predict(ctx._h, 0, Xf.ctypes.data, n, p, n, y.ctypes.data, 1, n, Zt.ctypes.data, n, Zq.ctypes.data, m, nlvdis,
                                      Xqf.ctypes.data, m, m, k, 1.0, 1e-4, 0, 0, nlv, pred.ctypes.data, ind.ctypes.data, dist.ctypes.data, w.ctypes.data))
print("same Z: neighbours equal", np.mean(ind == ref["listnn"]), "dist err", O.rel_fro(ref["listd"], dist), "pred err", O.rel_fro(ref["pred"][:, 0, :], pred))
# own pipeline on host arrays
fm = J.lwplsr(X, y, nlvdis=nlvdis, metric=metric, h=1.0, k=k, nlv=nlv, ctx=ctx)
res = J.predict(fm, Xq, nlv=range(0, nlv + 1), ctx=ctx)
print("host pipeline: neighbours equal", np.mean(res.listnn == ref["listnn"]), "pred err", O.rel_fro(ref["pred"][:, 0, :], np.stack([q_[:, 0] for q_ in res.pred], 1)))
