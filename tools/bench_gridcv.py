"""§8f rank 1 at cfg2 size: 5-fold gridcvlv(X, Y; fun = plskern, nlv = 0:25, score = rmsep) on device-resident data.
Each fold = one weighted plskern fit (weight 0 on the held-out rows, no rmrow copy) + an n x 25 score GEMM + one
statistics pass.  The reference would copy 3.2 GB per fold and run a CPU fit."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jchemo.jl_amd")]
import numpy as np, torch
import jchemo_hip as J
n, p, q, nlv, K = 1_000_000, 500, 10, 25, 5
ctx = J.Context(0, stream="torch"); lib = J.load()
X = J.colmajor_empty(n, p); Y = J.colmajor_empty(n, q)
ctx.check(lib.jch_fill_uniform(ctx._h, X.data_ptr(), n, p, n, 0, n, 20250112))
ctx.check(lib.jch_fill_uniform(ctx._h, Y.data_ptr(), n, q, n, 0, n, 20250113))
segm = J.segmkf(n, K, rep=1, seed=1)
J.gridcvlv(X, Y, segm=[segm[0][:1]], score=J.rmsep, fun=J.plskern, nlv=range(0, nlv + 1), ctx=ctx)   # warm-up (1 fold)
torch.cuda.synchronize(); t0 = time.perf_counter()
res = J.gridcvlv(X, Y, segm=segm, score=J.rmsep, fun=J.plskern, nlv=range(0, nlv + 1), ctx=ctx)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"workload": f"gridcvlv {K}-fold, plskern n={n} p={p} q={q} nlv=0..{nlv}, rmsep, device-resident", "seconds": dt,
                  "ms_per_fold": dt / K * 1e3, "folds_per_s": K / dt, "rmsep_nlv0_y1": float(res["res"][0, 0]), "rmsep_nlv25_y1": float(res["res"][-1, 0])}))
