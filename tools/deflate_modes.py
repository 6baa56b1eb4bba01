"""cfg4 (plsnipals n = 1e6, p = 2000, q = 1): per-LV device time under JCH_DEFLATE_* variants, alternating inside ONE
process on the same buffers (cf. tools/k2_modes.py).  DEFL_VARIANTS="A=1 B=2;C=3" """
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jchemo.jl_amd")]
import numpy as np, torch
import jchemo_hip as J
from jchemo_hip import _lib
n, p, q, nlv = 1_000_000, 2000, 1, int(os.environ.get("NLV", "10"))
lib = J.load()
ctx = J.Context(0, stream="torch"); ctx.set_profiling(True)
X = J.colmajor_empty(n, p); Y = J.colmajor_empty(n, q)
ctx.check(lib.jch_fill_uniform(ctx._h, X.data_ptr(), n, p, n, 0, n, 20250112))
ctx.check(lib.jch_fill_uniform(ctx._h, Y.data_ptr(), n, q, n, 0, n, 20250113))
T = J.colmajor_empty(n, nlv); wn = torch.empty(n, dtype=torch.float64, device="cuda")
P = np.zeros((p, nlv), order="F"); R = P.copy(); W = P.copy(); Cm = np.zeros((q, nlv), order="F"); TT = np.zeros(nlv)
xm = np.empty(p); xs = np.empty(p); ym = np.empty(q); ys = np.empty(q)
desc = _lib.PlsDesc(n=n, p=p, q=q, nlv=nlv, scal=0, dtype=_lib.F64, loc=_lib.LOC_DEVICE, inplace=0, reserved=0)
got = C.c_int32(0)
variants = [v for v in os.environ.get("DEFL_VARIANTS", "").split(";") if v] or [""]
res = {v: [] for v in variants}
for it in range(3 * len(variants) + 1):
    v = variants[it % len(variants)]
    for kv in [x for x in os.environ if x.startswith(("JCH_DEFLATE_", "JCH_SWEEP_", "JCH_NIPALS_"))]:
        del os.environ[kv]
    for kv in v.split():
        k_, v_ = kv.split("="); os.environ[k_] = v_
    ctx.check(lib.jch_plsnipals_fit(ctx._h, C.byref(desc), X.data_ptr(), n, Y.data_ptr(), n, None, T.data_ptr(), P.ctypes.data, R.ctypes.data,
                                    W.ctypes.data, Cm.ctypes.data, TT.ctypes.data, xm.ctypes.data, xs.ctypes.data, ym.ctypes.data, ys.ctypes.data,
                                    wn.data_ptr(), C.byref(got)))
    pr = ctx.profile()
    if it > 0:
        res[v].append((pr.fit_ms - pr.prologue_ms) / nlv)
for v in variants:
    print(f"[{v or 'default'}] ms per LV (sweep + deflate + small):", " ".join(f"{x:.3f}" for x in res[v]), flush=True)
