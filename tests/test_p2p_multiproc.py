"""P2P inbox transport (jchemo.jl_amd/csrc/p2p.hip) with REAL inter-process IPC: 3 processes on the one GPU of the box,
each with its own ctx and row shard, joined only by the inbox all-reduce (handles exchanged over gloo).  What a
one-GPU box cannot show is the xGMI hop itself; everything else of the multi-GPU fast path runs here."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import plsr_oracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("alg,world,reps", [("plskern", 3, 3), ("plsnipals", 3, 3), ("plskern_v2", 3, 3), ("plskern_bf16", 3, 3),
                                            ("plskern", 4, 150)])   # last: stress of the epoch / parity protocol (~3000 all-reduces per rank)
def test_p2p_inbox_three_processes(alg, world, reps, tmp_path):
    n, p, q, nlv = 5000, 150, 3, 6
    rng = np.random.default_rng(21)
    Lt = rng.standard_normal((n, 2 * nlv))
    X = Lt @ rng.standard_normal((2 * nlv, p)) + 0.5 * rng.standard_normal((n, p))
    Y = Lt[:, :q] @ rng.standard_normal((q, q)) + 0.3 * rng.standard_normal((n, q))
    w = rng.uniform(0.5, 1.5, n)
    if alg == "plskern_bf16":   # the oracle sees the bf16-rounded values
        import torch
        X = torch.from_numpy(X).to(torch.bfloat16).to(torch.float64).numpy()
        Y = torch.from_numpy(Y).to(torch.bfloat16).to(torch.float64).numpy()
    edges = np.array([0, 1700, 1704, n]) if world == 3 else np.array([0, 1200, 1204, 3100, n])   # one shard smaller than nlv
    np.savez(tmp_path / "inputs.npz", X=X, Y=Y, w=w, edges=edges, nlv=nlv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world),
               JCH_P2P_TIMEOUT_MS="20000", JCH_P2P_TEST_REPS=str(reps))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "p2p_worker.py"), str(tmp_path), alg],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q_ in procs:
                q_.kill()
            pytest.fail("a P2P worker process did not finish")
        outs.append(o)
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)
    res = [np.load(tmp_path / f"out_{r}.npz") for r in range(world)]
    for f in ("P", "R", "W", "C", "TT", "xmeans", "xscales"):
        for r in res[1:]:
            assert np.array_equal(res[0][f], r[f]), f                 # replicated state: bit-identical on every rank
    assert np.array_equal(res[0]["P"], res[0]["P0"])                  # and reproducible from call to call
    name = "plsnipals" if alg == "plsnipals" else "plskern"
    ref = getattr(O, name)(X, Y, w, nlv=nlv, scal=True)
    T = np.concatenate([r["T"] for r in res], axis=0)
    s = O.sign_align(ref.W, res[0]["W"])
    tol = 1e-3 if alg == "plskern_bf16" else 1e-8
    assert O.rel_fro(ref.T, T * s) < tol
    for f in ("P", "R", "W", "C"):
        assert O.rel_fro(getattr(ref, f), res[0][f] * s) < tol, f
    assert O.rel_fro(ref.TT, res[0]["TT"]) < tol
    assert O.rel_fro(ref.weights, np.concatenate([r["weights"] for r in res])) < 1e-12
