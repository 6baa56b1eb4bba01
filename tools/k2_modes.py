"""Is the K2 time of the cfg2 prologue stable inside one process, and does it depend on where the buffers live?
Prints prologue_ms of successive fits; between groups the ctx (workspace: row-major copy etc.) is re-created after an
extra allocation of varying size, so the copy lands at a different address."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jchemo.jl_amd")]
import numpy as np, torch
import jchemo_hip as J
from jchemo_hip import _lib
n, p, q, nlv = 1_000_000, 500, 10, 25
lib = J.load()
ctx0 = J.Context(0, stream="torch")
X = J.colmajor_empty(n, p); Y = J.colmajor_empty(n, q)
ctx0.check(lib.jch_fill_uniform(ctx0._h, X.data_ptr(), n, p, n, 0, n, 20250112))
ctx0.check(lib.jch_fill_uniform(ctx0._h, Y.data_ptr(), n, q, n, 0, n, 20250113))
T = J.colmajor_empty(n, nlv); wn = torch.empty(n, dtype=torch.float64, device="cuda")
P = np.zeros((p, nlv), order="F"); R = P.copy(); W = P.copy(); Cm = np.zeros((q, nlv), order="F"); TT = np.zeros(nlv)
xm = np.empty(p); xs = np.empty(p); ym = np.empty(q); ys = np.empty(q)
desc = _lib.PlsDesc(n=n, p=p, q=q, nlv=nlv, scal=0, dtype=_lib.F64, loc=_lib.LOC_DEVICE, inplace=0, reserved=0)
got = C.c_int32(0)
pads = []
for grp, pad_mb in enumerate([0, 3, 64, 1000, 7, 0]):
    if pad_mb:
        pads.append(torch.empty(pad_mb * 1024 * 1024 + 4096 * grp, dtype=torch.uint8, device="cuda"))
    ctx = J.Context(0, stream="torch"); ctx.set_profiling(True)
    out = []
    variants = [v for v in os.environ.get("K2_VARIANTS", "").split(";") if v] or [""]
    for it in range(4 * len(variants)):
        for kv in [x for x in os.environ if x.startswith("JCH_K2_")]:
            del os.environ[kv]
        for kv in variants[it % len(variants)].split():
            k_, v_ = kv.split("="); os.environ[k_] = v_
        ctx.check(lib.jch_plskern_fit(ctx._h, C.byref(desc), X.data_ptr(), n, Y.data_ptr(), n, None, T.data_ptr(), P.ctypes.data, R.ctypes.data,
                                      W.ctypes.data, Cm.ctypes.data, TT.ctypes.data, xm.ctypes.data, xs.ctypes.data, ym.ctypes.data, ys.ctypes.data,
                                      wn.data_ptr(), C.byref(got)))
        out.append(ctx.profile().prologue_ms)
    print(f"X at {X.data_ptr():#x}", file=sys.stderr)
    print(f"group {grp}: " + " | ".join(f"[{variants[i] or 'default'}] " + " ".join(f"{v:.3f}" for v in out[i::len(variants)]) for i in range(len(variants))), flush=True)
    ctx.close()
