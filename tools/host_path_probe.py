"""Where the host-arrays fit spends its time: H2D through the library (jch_col_stats on host X), whole fits with fresh vs
pre-touched output arrays."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jchemo.jl_amd")]
import numpy as np, torch
import jchemo_hip as J
from jchemo_hip import _lib
from oracle import c_oracle as CO
n, p, q, nlv = 1_000_000, 500, 10, 25
X = CO.fill_uniform(20250112, n, p); Y = CO.fill_uniform(20250113, n, q)
ctx = J.Context(0); lib = J.load()
m = np.empty(p)
for _ in range(3):
    t0 = time.perf_counter(); ctx.check(lib.jch_col_stats(ctx._h, 0, X.ctypes.data, n, p, n, None, m.ctypes.data, None)); dt = time.perf_counter() - t0
    print("jch_col_stats(host X): %.1f ms -> %.1f GB/s incl. one device pass" % (dt * 1e3, 4.0 / dt))
ctx.set_profiling(True)
for _ in range(3):
    t0 = time.perf_counter(); fm = J.plskern(X, Y, nlv=nlv, ctx=ctx); dt = time.perf_counter() - t0
    print("J.plskern(host): %.1f ms (device fit %.1f ms)" % (dt * 1e3, ctx.profile().fit_ms))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); fm = J.plskern(X, Y, nlv=nlv, ctx=ctx); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
