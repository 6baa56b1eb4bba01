"""Fixed cost of the P2P inbox all-reduce kernel: a ONE-rank inbox (self store, flag, immediate wait, sum) inside the
cfg2-per-rank fit (n = 125 000 rows).  Run under `rocprofv3 --kernel-trace --stats` to read k_p2p_allreduce's duration;
prints ms per fit with and without the inbox in the loop."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jchemo.jl_amd"))
import torch  # noqa: E402
import jchemo_hip as J  # noqa: E402

n, p, q, nlv = 125_000, 500, 10, 25
ctx = J.Context(0)
X = J.colmajor_empty(n, p); Y = J.colmajor_empty(n, q)
ctx.check(J.load().jch_fill_uniform(ctx._h, X.data_ptr(), n, p, n, 0, n, 20250112))
ctx.check(J.load().jch_fill_uniform(ctx._h, Y.data_ptr(), n, q, n, 0, n, 20250113))
torch.cuda.synchronize()


def run(steps=30):
    for _ in range(3):
        J.plskern(X, Y, nlv=nlv, ctx=ctx)
    t0 = time.perf_counter()
    for _ in range(steps):
        J.plskern(X, Y, nlv=nlv, ctx=ctx)
    return (time.perf_counter() - t0) / steps * 1e3


base = run()
h = ctx.p2p_export(1)
assert ctx.p2p_import([h], 0, 1)
ctx.p2p_enable(True)
with_inbox = run()
print(json.dumps({"rows": n, "ms_per_fit_no_collective": base, "ms_per_fit_one_rank_inbox": with_inbox,
                  "inbox_cost_us_per_lv": (with_inbox - base) / nlv * 1e3}))
