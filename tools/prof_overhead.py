"""What do the HIP events of jch_profile (two per sweep launch) cost a fit?  Same ctx, same buffers, profiling on / off alternating."""
import sys, os, time, json, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jchemo.jl_amd")); sys.path.insert(0, ROOT)
import jchemo_hip as J
from jchemo_hip import _lib
lib = J.load(); dev = torch.device("cuda", 0); ctx = J.Context(0, stream="torch")
out = {}
for n in (1_000_000, 125_000):
    p, q, nlv = 500, 10, 25
    X = J.colmajor_empty(n, p, dev); Y = J.colmajor_empty(n, q, dev)
    ctx.check(lib.jch_fill_uniform(ctx._h, X.data_ptr(), n, p, n, 0, n, 20250112)); ctx.check(lib.jch_fill_uniform(ctx._h, Y.data_ptr(), n, q, n, 0, n, 20250113))
    T = J.colmajor_empty(n, nlv, dev); wn = torch.empty(n, dtype=torch.float64, device=dev)
    P = np.zeros((p, nlv), order="F"); R = P.copy(order="F"); W = P.copy(order="F"); Cm = np.zeros((q, nlv), order="F"); TT = np.zeros(nlv)
    xm = np.empty(p); xs = np.empty(p); ym = np.empty(q); ys = np.empty(q); got = C.c_int32(0)
    desc = _lib.PlsDesc(n=n, p=p, q=q, nlv=nlv, scal=0, dtype=_lib.F64, loc=_lib.LOC_DEVICE, inplace=0, reserved=0)
    def step():
        ctx.check(lib.jch_plskern_fit(ctx._h, C.byref(desc), X.data_ptr(), n, Y.data_ptr(), n, None, T.data_ptr(), P.ctypes.data, R.ctypes.data, W.ctypes.data,
                                      Cm.ctypes.data, TT.ctypes.data, xm.ctypes.data, xs.ctypes.data, ym.ctypes.data, ys.ctypes.data, wn.data_ptr(), C.byref(got)))
    res = {"on": [], "off": []}
    for rep in range(4):
        for mode in ("on", "off"):
            ctx.set_profiling(mode == "on")
            for _ in range(3): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): step()
            torch.cuda.synchronize(); res[mode].append((time.perf_counter() - t0) / 20 * 1e3)
    out[f"n={n}"] = {k: [round(v, 4) for v in vs] for k, vs in res.items()}
    del X, Y, T, wn; torch.cuda.empty_cache()
print(json.dumps(out))
