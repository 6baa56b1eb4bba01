"""Worker of tests/test_lwplsr_replicas.py: one process per replica, all on GPU 0, joined by gloo only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jchemo.jl_amd")]


def main():
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import jchemo_hip as J
    from oracle import plsr_oracle as O
    n, p, m = 3000, 80, 23
    X = O.rand_matrix(1, n, p)
    Y = np.stack([X[:, :4] @ np.array([1.0, -2.0, 0.5, 3.0]) + np.sin(3 * X[:, 5]), X[:, 2] * X[:, 3]], axis=1) + 0.05 * O.rand_matrix(2, n, 2)
    Xq = O.rand_matrix(3, m, p)
    kw = dict(nlvdis=6, metric="mahal", h=1.2, k=70, nlv=5)
    ctx = J.Context(0)
    fm = J.lwplsr(X, Y, ctx=ctx, **kw)                        # replicated training set + global scores on every rank
    res = J.predict(fm, Xq, nlv=range(0, 6), ctx=ctx, rank=rank, world=world)
    out = dict(pred=np.stack(res.pred, axis=2), listnn=res.listnn, listd=res.listd, listw=res.listw)
    if rank == 0:
        full = J.predict(fm, Xq, nlv=range(0, 6), ctx=ctx)
        ref = O.lwplsr_predict(O.lwplsr(X, Y, **kw), Xq, nlv=range(0, 6))
        out.update(pred_unsplit=np.stack(full.pred, axis=2), listnn_unsplit=full.listnn, err_vs_oracle=O.rel_fro(ref["pred"], out["pred"]))
    np.savez(os.path.join(sys.argv[1], f"out_{rank}.npz"), **out)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
