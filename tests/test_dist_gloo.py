"""N > 1 path on CPU: world_size-2 gloo processes run the row-sharded plskern with REAL collectives at
exactly the points where libjchemo_hip issues its RCCL all-reduces (fit.hip): [sum w, n], column moments,
(scal) second moments, K = X'DY, and ONE [zp, tt] vector per latent variable.  The oracle plays the role
of the per-rank kernels here (test infrastructure); what is under test is the collective schedule, the
row partition used by bench.py and that the replicated small state ends up identical on every rank."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_total, p, q, nlv, scal, ret):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import plsr_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ncalls = [0]

    def allreduce(vec):
        t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ncalls[0] += 1
        return t.numpy()

    # the same row partition bench.py uses: rank g holds rows [n*g/G, n*(g+1)/G)
    row0 = (n_total * rank) // world
    n = (n_total * (rank + 1)) // world - row0
    X = O.rand_matrix(20250112, n, p, row0, n_total)
    Y = O.rand_matrix(20250113, n, q, row0, n_total)
    w = 0.25 + O.splitmix64_uniform(7, row0, n)
    fm = O.plskern_sharded([X], [Y], [w], nlv=nlv, scal=scal, allreduce=allreduce)
    # replicated small state must be bit-identical on every rank (deterministic collectives)
    blob = np.concatenate([fm.P.ravel(), fm.R.ravel(), fm.W.ravel(), fm.C.ravel(), fm.TT])
    others = [torch.zeros(blob.size, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(others, torch.from_numpy(blob))
    same = all(torch.equal(others[0], o) for o in others)
    Tparts = [None] * world
    dist.all_gather_object(Tparts, fm.T[0])
    if rank == 0:
        ret["fm"] = fm
        ret["T"] = np.vstack(Tparts)
        ret["same"] = same
        ret["ncalls"] = ncalls[0]
    dist.destroy_process_group()


@pytest.mark.parametrize("scal", [False, True])
def test_sharded_plskern_gloo_world2(scal):
    import torch.multiprocessing as mp
    from oracle import plsr_oracle as O
    n_total, p, q, nlv, world = 203, 31, 3, 6, 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, p, q, nlv, scal, ret), nprocs=world, join=True)
    X = O.rand_matrix(20250112, n_total, p)
    Y = O.rand_matrix(20250113, n_total, q)
    w = 0.25 + O.splitmix64_uniform(7, 0, n_total)
    ref = O.plskern(X, Y, w, nlv=nlv, scal=scal)
    fm = ret["fm"]
    assert ret["same"], "replicated small state diverged between ranks"
    # collectives: header, moments, (scal: variances), K, then exactly one per latent variable
    assert ret["ncalls"] == 3 + int(scal) + nlv
    s = O.sign_align(ref.W, fm.W)
    assert O.rel_fro(ref.T, ret["T"] * s) < 1e-10
    for f in ("P", "R", "W", "C"):
        assert O.rel_fro(getattr(ref, f), getattr(fm, f) * s) < 1e-10, f
    assert O.rel_fro(ref.TT, fm.TT) < 1e-11
    assert O.rel_fro(ref.xmeans, fm.xmeans) < 1e-13 and O.rel_fro(ref.xscales, fm.xscales) < 1e-12


def test_bench_row_partition_covers_all_rows():
    for n_total in (1_000_000, 1_000_003, 7):
        for world in (1, 2, 4, 8):
            cuts = [(n_total * r) // world for r in range(world + 1)]
            sizes = np.diff(cuts)
            assert cuts[0] == 0 and cuts[-1] == n_total and sizes.min() >= 0 and sizes.max() - sizes.min() <= 1
