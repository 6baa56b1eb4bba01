"""PCIe-inclusive (host arrays in -> host Plsr out) rate of plskern at cfg2, for DESIGN.md §7.  Never bench.py's `value`."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jchemo.jl_amd")]
import numpy as np
import jchemo_hip as J
from oracle import c_oracle as CO
n, p, q, nlv = 1_000_000, 500, 10, 25
X = CO.fill_uniform(20250112, n, p); Y = CO.fill_uniform(20250113, n, q)
ctx = J.Context(0)
J.plskern(X, Y, nlv=nlv, ctx=ctx)     # warm-up (workspace allocation, kernel load)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); fm = J.plskern(X, Y, nlv=nlv, ctx=ctx); ts.append(time.perf_counter() - t0)
t = sorted(ts)[1]
print(json.dumps({"workload": "plskern n=1e6 p=500 q=10 nlv=25, pageable host arrays in, host Plsr out (4.08 GB H2D + 0.21 GB D2H)",
                  "seconds_per_fit": t, "LV_per_s_host_to_host": nlv / t, "effective_H2D_GBps_upper_bound": 4.08 / t}))
