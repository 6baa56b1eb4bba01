"""cfg5 multi-GPU ("replicas"): the queries of predict(::Lwplsr) are split over the ranks, every rank
holds the whole training set, results are gathered in rank order.

CPU part (gloo, world size 2 and 3): the split / gather / merge logic of jchemo_hip.lwplsr_predict with the device call
replaced by the oracle's per-query arithmetic (what is under test is the host logic: slice boundaries incl. empty
slices, order, single-nlv vs range results).  GPU part (-m gpu): three processes on the one GPU of the box, each running
its slice through libjchemo_hip.so, compared with the unsplit call and the oracle."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_query_shard_covers_all_queries_once():
    sys.path.insert(0, os.path.join(ROOT, "jchemo.jl_amd"))
    from jchemo_hip.plsr import query_shard
    for m in (0, 1, 2, 7, 1000, 1001):
        for world in (1, 2, 3, 8):
            edges = [query_shard(m, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == m
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            assert max(hi - lo for lo, hi in edges) - min(hi - lo for lo, hi in edges) <= 1      # balanced
    with pytest.raises(ValueError):
        query_shard(10, 3, 3)


def _cpu_worker(rank, world, port, m, ret):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "jchemo.jl_amd")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import jchemo_hip.plsr as P
    from oracle import plsr_oracle as O
    n, p = 300, 12
    X = O.rand_matrix(1, n, p); y = X[:, :3] @ np.array([1.0, -1.0, 0.5]) + 0.1 * O.rand_matrix(2, n, 1)[:, 0]
    Xq = O.rand_matrix(3, m, p)
    kw = dict(nlvdis=4, metric="eucl", h=1.5, k=30, nlv=3)
    ofm = O.lwplsr(X, y, **kw)
    calls = []
    real = P.lwplsr_predict

    def fake_local(obj, Xs, *, nlv=None, ctx=None, rank=None, world=None, gather=None):
        if world is not None and world > 1:
            return real(obj, Xs, nlv=nlv, ctx=ctx, rank=rank, world=world, gather=gather)
        calls.append(Xs.shape[0])                       # the per-replica device call, played by the oracle
        r = O.lwplsr_predict(ofm, Xs, nlv=nlv)
        preds = [r["pred"][:, :, a].copy() for a in range(r["pred"].shape[2])]
        return P.LwplsrPred(preds[0] if len(preds) == 1 else preds, r["listnn"], r["listd"], r["listw"])

    P.lwplsr_predict = fake_local
    obj = P.Lwplsr(X, y.reshape(-1, 1), None, "eucl", 1.5, 30, 3, 1e-4, False)
    out = {}
    for name, nlv in (("range", range(0, 4)), ("single", 2)):
        res = real(obj, Xq, nlv=nlv, rank=rank, world=world)
        ref = O.lwplsr_predict(ofm, Xq, nlv=nlv)
        pred = np.stack(res.pred, axis=2) if isinstance(res.pred, list) else res.pred[:, :, None]
        # (the stand-in's BLAS calls see 1-5 rows per slice instead of m: equal up to rounding, neighbours exactly)
        close = lambda a, b: a.shape == b.shape and np.allclose(a, b, rtol=1e-11, atol=1e-13)
        out[name] = (close(pred, ref["pred"]) and np.array_equal(res.listnn, ref["listnn"]) and close(res.listd, ref["listd"])
                     and close(res.listw, ref["listw"]), isinstance(res.pred, list))
    ret[rank] = (out, list(calls))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,m", [(2, 9), (3, 2)])
def test_query_split_and_gather_gloo(world, m):
    import multiprocessing as mp
    ctxm = mp.get_context("spawn")
    port = _free_port()
    with ctxm.Manager() as man:
        ret = man.dict()
        procs = [ctxm.Process(target=_cpu_worker, args=(r, world, port, m, ret)) for r in range(world)]
        for pr in procs:
            pr.start()
        for pr in procs:
            pr.join(180)
        assert all(pr.exitcode == 0 for pr in procs)
        from jchemo_hip.plsr import query_shard
        for r in range(world):
            out, calls = ret[r]
            assert out["range"] == (True, True) and out["single"] == (True, False), (r, out)       # full result on EVERY rank
            lo, hi = query_shard(m, r, world)
            assert calls == ([hi - lo] * 2 if hi > lo else [])                                    # empty slice: no device call


@pytest.mark.gpu
def test_lwplsr_replicas_three_processes_one_gpu(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="3")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "lwplsr_replica_worker.py"), str(tmp_path)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(3)]
    outs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q_ in procs:
                q_.kill()
            pytest.fail("a replica worker did not finish")
        outs.append(o)
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)
    res = [np.load(tmp_path / f"out_{r}.npz") for r in range(3)]
    for r in res[1:]:
        for f in ("pred", "listnn", "listd", "listw"):
            assert np.array_equal(res[0][f], r[f]), f          # every rank ends up with the same, complete result
    assert np.array_equal(res[0]["pred"], res[0]["pred_unsplit"]) and np.array_equal(res[0]["listnn"], res[0]["listnn_unsplit"])
    assert float(res[0]["err_vs_oracle"]) < 1e-8
