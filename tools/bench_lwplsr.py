"""cfg5 (BASELINE.json configs[4]): lwplsr n=1e5 p=500, 1000 queries x k=200 neighbours, nlvdis=20, mahal, nlv=15.
Reports queries/s of predict(::Lwplsr) on the GPU(s) (device-resident data) and of the numpy oracle on a query sample.

    python tools/bench_lwplsr.py
    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 tools/bench_lwplsr.py
G > 1 = "replicas" (SURVEY §8e): every rank holds the whole training set on its own GPU and predicts the slice
query_shard(m, rank, G) of the queries; the only exchange is the gather of the predictions (JCH_BENCH_REHEARSAL=1: every
rank on GPU 0 with gloo, the way to run this on a one-GPU box)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jchemo.jl_amd")]
import numpy as np, torch
import jchemo_hip as J
from oracle import plsr_oracle as O, c_oracle as CO

world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
rehearsal = os.environ.get("JCH_BENCH_REHEARSAL", "0") == "1"
local = 0 if rehearsal else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)
n, p, m, k, nlvdis, nlv = int(os.environ.get("LW_N", "100000")), 500, 1000, 200, 20, int(os.environ.get("LW_NLV", "15"))   # (LW_NLV: probe of the per-LV cost)
ctx = J.Context(local, stream="torch")
lib = J.load()
dev = torch.device("cuda", local)
# spectra-like inputs: 30 latent sources + noise, so that the 20 global and 15 local LVs are numerically meaningful
# (on iid-uniform columns PLS1 exhausts its Krylov space after ~10 LVs: TT -> 1e-12 and every implementation,
# the reference included, returns rounding noise for the later LVs)
r = 30
def gen(rows, seed):
    S = J.colmajor_empty(rows, r, dev); E = J.colmajor_empty(rows, p, dev)
    ctx.check(lib.jch_fill_uniform(ctx._h, S.data_ptr(), rows, r, rows, 0, rows, seed))
    ctx.check(lib.jch_fill_uniform(ctx._h, E.data_ptr(), rows, p, rows, 0, rows, seed + 100))
    L = J.colmajor_empty(r, p, dev); ctx.check(lib.jch_fill_uniform(ctx._h, L.data_ptr(), r, p, r, 0, r, 777))
    out = J.colmajor_empty(rows, p, dev); out.copy_(S @ L + 0.1 * E)
    return out
X = gen(n, 20250112); Xq = gen(m, 20250115)
beta = torch.zeros(p, dtype=torch.float64, device=dev); beta[:5] = torch.tensor([1.0, -2.0, 0.5, 3.0, 1.5], dtype=torch.float64)
noise = J.colmajor_empty(n, 1, dev); ctx.check(lib.jch_fill_uniform(ctx._h, noise.data_ptr(), n, 1, n, 0, n, 20250113))
y = J.colmajor_empty(n, 1, dev); y.copy_((X @ beta + torch.sin(3 * X[:, 5])).reshape(-1, 1) + 0.05 * noise)
t0 = time.perf_counter(); fm = J.lwplsr(X, y, nlvdis=nlvdis, metric="mahal", h=1.0, k=k, nlv=nlv, ctx=ctx); torch.cuda.synchronize(); t_fit = time.perf_counter() - t0
kw = dict(rank=rank, world=world) if world > 1 else {}
J.predict(fm, Xq, nlv=range(0, nlv + 1), ctx=ctx, **kw)   # warm-up
reps = 5
if world > 1:
    dist.barrier()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    res = J.predict(fm, Xq, nlv=range(0, nlv + 1), ctx=ctx, **kw)     # (G > 1: includes the gather of the results)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
dt = (time.perf_counter() - t0) / reps
if rank == 0:
    # CPU oracle on a sample of the queries (same neighbours/weights as computed by the oracle itself)
    ms = 20
    Xh, yh, Xqh = X.cpu().numpy(), y.cpu().numpy(), Xq[:ms].cpu().numpy()
    t0 = time.perf_counter(); ref = O.lwplsr_predict(O.lwplsr(Xh, yh, nlvdis=nlvdis, metric="mahal", h=1.0, k=k, nlv=nlv), Xqh, nlv=range(0, nlv + 1)); dtc = time.perf_counter() - t0
    pred = np.stack([p_[:ms, 0] for p_ in res.pred], axis=1)
    print(json.dumps({"workload": f"lwplsr predict n={n} p={p} m={m} k={k} nlvdis={nlvdis} mahal nlv=0..{nlv}", "n_gpus": world,
                      "parallelism": "single GPU" if world == 1 else f"replicas: training set on every GPU, queries split {world} ways, results gathered",
                      "gpu_queries_per_s": m / dt, "gpu_ms_per_call": dt * 1e3, "global_fit_s": t_fit, "cpu_oracle_queries_per_s": ms / dtc,
                      "cpu_sample": f"{ms} queries, numpy oracle incl. global fit",
                      "parity_pred_rel_fro_on_sample": O.rel_fro(ref["pred"][:, 0, :], pred), "neighbours_equal": float(np.mean(res.listnn[:ms] == ref["listnn"]))}))
ctx.close()
if world > 1:
    dist.destroy_process_group()
