"""Where one gridcvlv fold at cfg2 size spends its time: host-side fold setup, the weighted fit, the statistics call."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jchemo.jl_amd")]
import numpy as np, torch
import jchemo_hip as J
from jchemo_hip import plsr as PL
n, p, q, nlv, K = 1_000_000, 500, 10, 25, 5
ctx = J.Context(0, stream="torch"); lib = J.load()
X = J.colmajor_empty(n, p); Y = J.colmajor_empty(n, q)
ctx.check(lib.jch_fill_uniform(ctx._h, X.data_ptr(), n, p, n, 0, n, 20250112))
ctx.check(lib.jch_fill_uniform(ctx._h, Y.data_ptr(), n, q, n, 0, n, 20250113))
segm = J.segmkf(n, K, rep=1, seed=1)
s = np.asarray(segm[0][0])
def t(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
def setup():
    held = np.zeros(n); held[s] = 1.0
    w = 1.0 - held
    return torch.as_tensor(held, device=X.device), torch.as_tensor(w, device=X.device)
ms_setup, (held, w) = t(setup)
ms_fit_w, fm = t(lambda: J.plskern(X, Y, w, nlv=nlv, ctx=ctx))
ms_fit_u, _ = t(lambda: J.plskern(X, Y, nlv=nlv, ctx=ctx))
ms_sums, _ = t(lambda: PL._score_sums_lv(fm.T, fm, Y, held, list(range(0, nlv + 1)), ctx))
print(json.dumps({"fold_setup_ms": ms_setup, "weighted_fit_ms": ms_fit_w, "unweighted_fit_ms": ms_fit_u, "score_sums_lv_ms": ms_sums}))
