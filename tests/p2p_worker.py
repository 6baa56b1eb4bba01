"""Worker of tests/test_p2p_multiproc.py: one PROCESS per rank, all on GPU 0, joined only by the P2P inbox transport
(IPC handles exchanged over a gloo group — no RCCL: it refuses two ranks on one device).  Each rank fits its row shard
and stores what it got; the parent test compares with the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jchemo.jl_amd"))


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    outdir, alg = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import jchemo_hip as J
    ctx = J.Context(0)
    handle = ctx.p2p_export(world)
    handles = [None] * world
    dist.all_gather_object(handles, handle)
    ok = ctx.p2p_import(handles, rank, world)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if flag.item() != 1:
        print(f"rank {rank}: P2P self-test failed: {getattr(ctx, 'p2p_error', '')}", file=sys.stderr)
        sys.exit(3)
    ctx.p2p_enable(True)
    d = np.load(os.path.join(outdir, "inputs.npz"))
    X, Y, w, edges, nlv = d["X"], d["Y"], d["w"], d["edges"], int(d["nlv"])
    a, b = int(edges[rank]), int(edges[rank + 1])
    Xs, Ys, ws = np.asfortranarray(X[a:b]), np.asfortranarray(Y[a:b]), w[a:b].copy()
    kw = {}
    name = alg
    if alg == "plskern_v2":
        name, kw = "plskern", {"variant": 1}
    if alg == "plskern_bf16":
        name = "plskern"
        Xd = J.colmajor_empty(b - a, X.shape[1], dtype=torch.bfloat16); Xd.copy_(torch.from_numpy(Xs))
        Yd = J.colmajor_empty(b - a, Y.shape[1], dtype=torch.bfloat16); Yd.copy_(torch.from_numpy(Ys))
        Xs, Ys, ws = Xd, Yd, torch.from_numpy(ws).cuda()
    reps = int(os.environ.get("JCH_P2P_TEST_REPS", "3"))
    fms = [getattr(J, name)(Xs, Ys, ws, nlv=nlv, scal=True, ctx=ctx, **kw) for _ in range(reps)]   # repeated: epochs keep in step
    assert all(np.array_equal(fms[0].P, f.P) and np.array_equal(fms[0].TT, f.TT) for f in fms[1:]), "fits differ from call to call"
    fm = fms[-1]
    T = fm.T.cpu().numpy() if hasattr(fm.T, "cpu") else fm.T
    wn = fm.weights.cpu().numpy() if hasattr(fm.weights, "cpu") else fm.weights
    np.savez(os.path.join(outdir, f"out_{rank}.npz"), T=T, P=fm.P, R=fm.R, W=fm.W, C=fm.C, TT=fm.TT, xmeans=fm.xmeans,
             xscales=fm.xscales, weights=wn, P0=fms[0].P)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
