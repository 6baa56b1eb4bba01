"""bench.py's multi-rank code path, executed: 3 ranks on the ONE GPU of the box (JCH_BENCH_REHEARSAL=1: gloo for the
host-side exchange, the P2P inbox as the only transport — RCCL refuses two ranks on one device), launched exactly as the
driver launches the N > 1 bench (`python -m torch.distributed.run ... bench.py --gpus N`).  Asserts the JSON schema of the
N > 1 line (collective block, per-rank min / max, rank-share efficiency) so that the first run on real
xGMI is diagnosable from its one line, and that cfg3's bench code path (bf16 storage, n = 8e6 rows sharded) has run.
What a one-GPU box cannot show — the xGMI hop and RCCL with > 1 rank — stays for the driver's scaling run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(extra, world=3, timeout=420):
    env = dict(os.environ, JCH_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", JCH_P2P_TIMEOUT_MS="30000")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1"] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-3000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def _check_schema(out, world, n_total):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data",
                "config", "roofline", "device_ms_per_step", "collective", "device_ms_per_step_ranks", "rank_share", "efficiency_vs_rank_share"):
        assert key in out, key
    assert out["n_gpus"] == world and out["value"] > 0 and out["scaling"] == "strong"
    assert out["config"]["n"] == n_total and abs(out["config"]["rows_per_gpu"] - n_total / world) <= 1
    dm = out["device_ms_per_step"]
    for key in ("fit", "prologue", "sweeps", "small_state_and_gaps", "collective", "small_state_kernels_and_gaps"):
        assert key in dm and dm[key] >= 0.0, key
    assert dm["collective"] > 0.0                       # the inbox exchange was timed inside the fits
    assert dm["collective"] <= dm["small_state_and_gaps"] + 1e-6
    c = out["collective"]
    assert c["ranks_seen"]["torch_distributed_world"] == world
    assert c["ranks_seen"]["jch_ctx_comm_info"]["nranks"] == world
    assert c["ranks_seen"]["inbox"] == world            # a vector of ones summed over the transport = ranks actually reached
    pl = c["per_lv_allreduce_us"]
    assert pl["in_timed_fits"] > 0.0 and pl["calls_per_fit"] == out["config"]["nlv"]
    assert pl["probe_inbox_kernel"] is not None and pl["probe_inbox_kernel"] > 0.0
    assert pl["probe_rccl"] is None                     # rehearsal: no RCCL on a shared device
    assert "fused" in c["transport_in_fit"] or "inbox" in c["transport_in_fit"]
    for what in ("per_lv_zp_tt", "prologue_moments", "prologue_xty"):
        assert c["probes"]["inbox"][what]["us"] > 0.0
    rs = out["device_ms_per_step_ranks"]
    for key in ("fit", "prologue", "sweeps", "small_state_and_gaps", "collective", "rows"):
        assert rs[key]["min"] <= rs[key]["max"], key
    assert rs["rows"]["min"] >= n_total // world and rs["rows"]["max"] <= n_total // world + 1
    assert out["rank_share"]["ms_per_fit"]["max"] > 0.0 and out["efficiency_vs_rank_share"] > 0.0
    r = out["roofline"]
    assert r["bound"] == "hbm" and 0.0 < r["frac"] < 1.0 and r["launches"] == 3 * out["config"]["nlv"]


def test_bench_three_ranks_on_one_gpu_f64(tmp_path):
    """cfg2 (n = 1e6, p = 500, q = 10, nlv = 25, Float64) sharded 3 ways."""
    out = _run([])
    _check_schema(out, 3, 1_000_000)
    assert out["dtype"] == "f64" and out["config"]["nlv"] == 25
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_rehearsal_3ranks_f64.json"), "w") as f:
        json.dump(out, f)


def test_bench_three_ranks_on_one_gpu_cfg3_bf16(tmp_path):
    """BASELINE.json configs[2] through bench.py: bf16 storage, ALL n = 8e6 rows, sharded 3 ways on the one GPU."""
    out = _run(["--dtype", "bf16", "--rows", "8000000"], timeout=600)
    _check_schema(out, 3, 8_000_000)
    assert out["dtype"].startswith("bf16")
    with open(os.path.join(ROOT, "gpurun_out", "bench_rehearsal_3ranks_cfg3_bf16_n8e6.json"), "w") as f:
        json.dump(out, f)
