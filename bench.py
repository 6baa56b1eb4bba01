#!/usr/bin/env python3
"""bench.py — latent-variables/sec of the plskern hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one complete `plskern(X, Y; nlv = 25)` fit (prologue + 25 latent variables) through the C ABI,
with X (n x p) and Y (n x q) already resident in HBM in Julia's column-major layout (README.md:79-94:
X = rand(n, p), Y = rand(n, q); here the portable splitmix64 generator, filled on the device).  For N > 1
the n = 1e6 rows are sharded over the ranks (STRONG scaling, as the metric is quoted at fixed n) and every
latent variable costs one RCCL all-reduce of [zp (p), tt].  value = nlv * steps / max-over-ranks wall time.

The JSON line also carries
  roofline     : the fused sweep kernel's algorithmic bytes per launch (n_local*ld*8 + 16 n_local)
                 / its average duration measured with HIP events on the ctx stream inside the timed region,
                 against the 8 TB/s HBM3E peak;
  cpu_baseline : both CPU stand-ins of the reference's `plskern!` — the numpy/OpenBLAS restatement (the same dgemv /
                 dgemm / dgesdd Julia calls) and the C + OpenMP port of its schedule — timed on this host's cores at the
                 FULL n, two runs each; value = the faster one (rank 0, N == 1 only).
  other_configs: (N == 1) the other BASELINE.json configs that fit one GPU, run AFTER the headline's timed region: cfg4
                 plsnipals n=1e6 p=2000 q=1 nlv=50, the one-GPU share of cfg3 (bf16 storage, n=1e6), cfg5 lwplsr
                 (1000 queries x k=200) — each with value, ms_per_step and its own roofline block.
  collective   : (N > 1) what the all-reduces cost — transport, ranks actually reached, per-LV all-reduce microseconds
                 through RCCL and through the P2P inbox (probe on the fit's own message size + HIP-event / device-clock
                 time inside profiled fits), prologue collectives; plus per-rank device times (min / max over ranks) and
                 rank_share = the same shard fitted WITHOUT any collective on the same GPU (efficiency_vs_rank_share).
                 DESIGN.md §8 says how to read each field.
"""
import argparse
import ctypes as C
import json
import os
import sys
import math, time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "jchemo.jl_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
README_PLSKERN_LVS = 25 / 8.100469   # README.md:90-91: plskern n=1e6 p=500 q=10 nlv=25 in 8.10 s (i9-10885H)


PMC_FILE = "profiles/r04_pmc_traffic.json"


def pmc_traffic(algo, n_local, p):
    """HBM bytes per sweep launch from the COMMITTED PMC pass (profiles/r04_pmc_traffic.json: FETCH_SIZE x2 gfx950 correction
    + WRITE_SIZE, separate rocprofv3 passes, tools/final_pmc_r04.sh), scaled by rows when this rank holds a different
    share.  Not measured in the bench run itself (counters need the profiler); None for shapes without a committed pass."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            pm = json.load(f)
        w = pm["workload"]
        if algo == w["algo"] and p == w["p"]:
            return pm["hbm_bytes_per_launch"] * (n_local / w["n"])
    except Exception:
        pass
    return None


def pmc_config_traffic(key, **shape):
    """HBM bytes per `launch` (in the unit of that configuration's roofline block) of one of the OTHER bench configurations, from
    the same committed PMC file (`configs`: cfg4_plsnipals per LV, bf16_share per sweep launch, cfg5_lwplsr per call of 1000
    queries); None when the shape differs from the one the counters were collected at."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            c = json.load(f)["configs"][key]
        if all(c.get(k_) == v_ for k_, v_ in shape.items()):
            return c["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def _blas_threads():
    """Threads numpy's BLAS (OpenBLAS) will use — what Julia's `BLAS.get_num_threads()` reports for the reference."""
    try:
        from threadpoolctl import threadpool_info
        info = [i for i in threadpool_info() if i.get("user_api") == "blas"]
        return max((int(i.get("num_threads", 0)) for i in info), default=0), ",".join(sorted({str(i.get("internal_api")) for i in info}))
    except Exception:
        return 0, "unknown"


def _cpu_share():
    """What the box actually grants this process: scheduler affinity and the cgroup CPU quota (cpu.max / cfs_quota), in cores.  A GPU
    box of this pool exposes all hardware threads of the host but may cap the CPU TIME of a job (the pool's notes speak of a share
    of 16 per GPU), which bounds any CPU baseline timed there."""
    out = {"affinity_cpus": None, "cgroup_quota_cores": None}
    try:
        out["affinity_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    out["cgroup_quota_cores"] = float(txt[0]) / float(txt[1])
            else:
                q_ = float(txt[0])
                if q_ > 0:
                    out["cgroup_quota_cores"] = q_ / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    return out


def cpu_baseline(n_total, p, q, nlv, sample_rows):
    """Both CPU stand-ins for the reference's `plskern!` (README.md:93), at `sample_rows` rows (default: the FULL n, no
    extrapolation), each run twice on all host cores; value = the faster stand-in's better run.
      openblas: oracle/plsr_oracle.py plskern_ — the restatement whose matrix-vector products (src/plskern.jl:162,167),
                X'DY (:132) and SVD (:154) go to the SAME routines Julia dispatches to (OpenBLAS dgemv / dgemm, LAPACK dgesdd)
      c_port  : oracle/plsr_oracle.c — the reference's schedule in plain C + OpenMP
    X is first-touched in parallel (OpenMP fill) so its pages are spread over the NUMA nodes."""
    from oracle import c_oracle as CO
    from oracle import plsr_oracle as O
    CO.build()
    ns = int(min(sample_rows, n_total))
    X = CO.fill_uniform(20250112, ns, p, 0, n_total)
    Y = CO.fill_uniform(20250113, ns, q, 0, n_total)
    scale = n_total / ns
    runs = {"openblas": [], "c_port": []}
    # a cgroup CPU quota below the thread count makes an oversubscribed OpenMP team stall on throttling: cap the C port's team at
    # twice the quota (the quota itself is reported next to the result)
    share = _cpu_share()
    if share["cgroup_quota_cores"]:
        CO.lib().orc_set_num_threads(int(max(1, min(CO.lib().orc_num_threads(), round(2 * share["cgroup_quota_cores"])))))
    for name, fn in (("c_port", lambda: CO.plskern_(X, Y, None, nlv=nlv, scal=False)),
                     ("openblas", lambda: O.plskern_(X, Y, None, nlv=nlv, scal=False))):
        for _ in range(2):     # `plskern!` is in place: the second run re-centres centred data — same passes, same cost
            t0 = time.perf_counter()
            fn()
            runs[name].append(time.perf_counter() - t0)
    best = {k_: min(v) for k_, v in runs.items()}
    winner = min(best, key=best.get)
    blas_threads, blas_name = _blas_threads()
    omp_threads = int(CO.lib().orc_num_threads())
    cores = blas_threads if winner == "openblas" else omp_threads
    # probe for the real reference (Julia + Jchemo) and record the outcome; never assume it
    import shutil, subprocess
    jl = shutil.which("julia")
    probe = "julia: not on PATH (reference itself cannot be timed here)"
    if jl:
        try:
            r = subprocess.run([jl, "-e", "using Jchemo"], capture_output=True, timeout=120)
            probe = "julia present, `using Jchemo` " + ("works (not timed by this harness yet)" if r.returncode == 0 else "fails")
        except Exception as e:
            probe = f"julia present, probe failed: {e}"
    lvs = {k_: nlv / (v * scale) for k_, v in best.items()}
    return {"value": lvs[winner], "unit": "LV/s", "cores": cores, "kind": "port", "stand_in": winner,
            "thread_cap_note": (f"the reported stand-in ran on {cores} threads of a host with {os.cpu_count()} hardware threads "
                                f"(OpenBLAS in this image is built for at most {blas_threads} threads; the C port used {omp_threads}); "
                                "vs_cpu_baseline is a ratio against THAT, context only — the roofline fraction is the quality measure"),
            "host_cpus": os.cpu_count(), "cpu_share": _cpu_share(), "threads": {"openblas": blas_threads, "blas": blas_name, "c_port_openmp": omp_threads},
            "effective_GBps": {k_: (2 * nlv + 3) * ns * p * 8 / v / 1e9 for k_, v in best.items()},   # the reference schedule: 2 passes over X per LV + 3 in the preamble
            "lv_per_s": lvs, "seconds": {k_: [round(t, 3) for t in v] for k_, v in runs.items()},
            "sample": (f"plskern! (in place, README.md:93) on {'all' if ns == n_total else 'the first'} {ns} of {n_total} rows "
                       f"(p={p}, q={q}, nlv={nlv}), two runs per stand-in, best run reported"
                       + ("" if ns == n_total else f", time scaled x{scale:.2f}") + f"; {probe}")}


def _roof(bytes_per_launch, kernel_ms_total, launches, kernel, note=None):
    avg_s = kernel_ms_total / max(launches, 1) * 1e-3
    ach = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
    r = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "traffic": None, "bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_s * 1e3, "launches": launches}
    if note:
        r["note"] = note
    return r


def secondary_fit(J, _lib, lib, ctx, dev, *, label, algo, n, p, q, nlv, bf16, steps, warmup, reserved=0):
    """One of BASELINE.json's OTHER configs on this GPU (device-resident synthetic inputs, same generator and the same C-ABI
    entry points as the headline; run after — never inside — the headline's timed region).  value = LV/s over `steps` fits."""
    X = J.colmajor_empty(n, p, dev); Y = J.colmajor_empty(n, q, dev)
    ctx.check(lib.jch_fill_uniform(ctx._h, X.data_ptr(), n, p, n, 0, n, 20250112))
    ctx.check(lib.jch_fill_uniform(ctx._h, Y.data_ptr(), n, q, n, 0, n, 20250113))
    if bf16:
        Xb = J.colmajor_empty(n, p, dev, dtype=torch.bfloat16); Xb.copy_(X); X = Xb
        Yb = J.colmajor_empty(n, q, dev, dtype=torch.bfloat16); Yb.copy_(Y); Y = Yb
        del Xb, Yb
        torch.cuda.empty_cache()
    kmax = min(nlv, p, n)
    T = J.colmajor_empty(n, kmax, dev); wn = torch.empty(n, dtype=torch.float64, device=dev)
    P = np.zeros((p, kmax), order="F"); R = np.zeros((p, kmax), order="F"); W = np.zeros((p, kmax), order="F")
    Cm = np.zeros((q, kmax), order="F"); TT = np.zeros(kmax)
    xm = np.empty(p); xs = np.empty(p); ym = np.empty(q); ys = np.empty(q)
    desc = _lib.PlsDesc(n=n, p=p, q=q, nlv=nlv, scal=0, dtype=_lib.BF16 if bf16 else _lib.F64, loc=_lib.LOC_DEVICE, inplace=0, reserved=reserved)
    got = C.c_int32(0)
    entry = lib.jch_plsnipals_fit if algo == "plsnipals" else lib.jch_plskern_fit

    def step():
        ctx.check(entry(ctx._h, C.byref(desc), X.data_ptr(), n, Y.data_ptr(), n, None, T.data_ptr(), P.ctypes.data, R.ctypes.data,
                        W.ctypes.data, Cm.ctypes.data, TT.ctypes.data, xm.ctypes.data, xs.ctypes.data, ym.ctypes.data, ys.ctypes.data,
                        wn.data_ptr(), C.byref(got)))
    for _ in range(warmup):
        step()
    sw = 0.0; nl = 0; fit = 0.0; pro = 0.0; sb = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        step()
        pr = ctx.profile()
        sw += pr.sweep_ms; nl += pr.sweep_launches; fit += pr.fit_ms; pro += pr.prologue_ms; sb = pr.sweep_bytes
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    k = got.value
    assert np.all(np.isfinite(TT[:k])) and np.all(TT[:k] > 0), "secondary fit produced non-finite / non-positive t't"
    kernel = ("k_sweep_lazy + a sixth of k_kpass_lazy (OPT-IN one-pass variant: ONE read of X per LV, rows rewritten every 6th LV; bytes_per_launch = bytes actually moved per LV)"
              if (algo == "plsnipals" and reserved == 4) else
              "k_sweep_lazy + k_kpass_lazy (per LV: two reads of X, rows rewritten every 6th LV; bytes_per_launch = bytes actually moved per LV)"
              if algo == "plsnipals" else "k_sweep_bf16_v2 (fused sweep over the bf16 row-major copy)" if bf16 else "k_sweep")
    roof = _roof(sb, sw, nl, kernel)
    roof["traffic"] = pmc_config_traffic("bf16_share", n=n, p=p) if bf16 else (pmc_config_traffic("cfg4_plsnipals", n=n, p=p) if (algo == "plsnipals" and not reserved) else None)
    roof["traffic_source"] = PMC_FILE + ": committed rocprofv3 --pmc passes of this configuration's kernels (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); NOT measured in this run"
    out = {"config": label, "metric": "latent-variables/sec", "value": k * steps / dt, "unit": "LV/s", "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "dtype": "bf16 storage / f32 rows / f64 state" if bf16 else "f64",
           "roofline": roof,
           "device_ms_per_step": {"fit": fit / steps, "prologue": pro / steps, "dominant_kernels": sw / steps,
                                  "small_state_and_gaps": (fit - pro - sw) / steps}}
    del X, Y, T, wn
    torch.cuda.empty_cache()
    return out


def secondary_lwplsr(J, lib, ctx, dev, calls):
    """cfg5 (BASELINE.json configs[4]): predict(::Lwplsr) for 1000 queries, k = 200, n = 1e5, p = 500, nlvdis = 20, mahal,
    nlv = 0..15 on device-resident spectra-like inputs (30 latent sources + noise: on iid-uniform columns a PLS1 fit has
    ~10 meaningful LVs and every fp64 implementation returns rounding noise afterwards; tools/bench_lwplsr.py).
    value = queries/s of the whole predict call (global transform + kNN + weights + local fits + results to the host)."""
    n, p, m, k, nlvdis, nlv, r = 100_000, 500, 1000, 200, 20, 15, 30

    def gen(rows, seed):
        S = J.colmajor_empty(rows, r, dev); E = J.colmajor_empty(rows, p, dev); L = J.colmajor_empty(r, p, dev)
        ctx.check(lib.jch_fill_uniform(ctx._h, S.data_ptr(), rows, r, rows, 0, rows, seed))
        ctx.check(lib.jch_fill_uniform(ctx._h, E.data_ptr(), rows, p, rows, 0, rows, seed + 100))
        ctx.check(lib.jch_fill_uniform(ctx._h, L.data_ptr(), r, p, r, 0, r, 777))
        out = J.colmajor_empty(rows, p, dev); out.copy_(S @ L + 0.1 * E)
        return out
    X = gen(n, 20250112); Xq = gen(m, 20250115)
    beta = torch.zeros(p, dtype=torch.float64, device=dev); beta[:5] = torch.tensor([1.0, -2.0, 0.5, 3.0, 1.5], dtype=torch.float64)
    noise = J.colmajor_empty(n, 1, dev); ctx.check(lib.jch_fill_uniform(ctx._h, noise.data_ptr(), n, 1, n, 0, n, 20250113))
    y = J.colmajor_empty(n, 1, dev); y.copy_((X @ beta + torch.sin(3 * X[:, 5])).reshape(-1, 1) + 0.05 * noise)
    fm = J.lwplsr(X, y, nlvdis=nlvdis, metric="mahal", h=1.0, k=k, nlv=nlv, ctx=ctx)
    J.predict(fm, Xq, nlv=range(0, nlv + 1), ctx=ctx)
    scr0, red0 = ctx.counter(2), ctx.counter(3)          # JCH_COUNTER_KNN_SCREENED / _SCREEN_REDONE
    dev_ms = {"query_scores_or_copies": 0.0, "knn_and_weights": 0.0, "local_fits": 0.0}
    # the timed calls run WITHOUT the stage events (four event records and three elapsed-time queries per call are ~4 % of a 1.2 ms
    # call); the stage times come from as many calls with them, outside the timed region
    ctx.set_profiling(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(calls):
        res = J.predict(fm, Xq, nlv=range(0, nlv + 1), ctx=ctx)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / calls
    ctx.set_profiling(True)
    for _ in range(calls):
        J.predict(fm, Xq, nlv=range(0, nlv + 1), ctx=ctx)
        pr = ctx.profile()
        dev_ms["query_scores_or_copies"] += pr.smallstate_ms; dev_ms["knn_and_weights"] += pr.prologue_ms - pr.smallstate_ms; dev_ms["local_fits"] += pr.sweep_ms
        gather_bytes = pr.sweep_bytes
    ctx.set_profiling(False)
    pred = np.stack([p_[:, 0] for p_ in res.pred], axis=1)
    assert pred.shape == (m, nlv + 1) and np.all(np.isfinite(pred)), "lwplsr predictions are not finite"
    # the same call WITHOUT the neighbour lists / distances / weights on the host (NULL outputs of the C ABI): predictions only
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(calls):
        res_p = J.lwplsr_predict(fm, Xq, nlv=range(0, nlv + 1), ctx=ctx, lists=False)
    torch.cuda.synchronize(); dt_p = (time.perf_counter() - t1) / calls
    assert res_p.listnn is None and all(np.array_equal(a_, b_) for a_, b_ in zip(res.pred, res_p.pred))
    out = {"config": f"lwplsr predict n={n} p={p}, {m} queries x k={k} neighbours, nlvdis={nlvdis} mahal, nlv=0..{nlv} (BASELINE.json configs[4])",
           "metric": "queries/sec", "value": m / dt, "unit": "queries/s", "steps": calls, "warmup": 1, "ms_per_step": dt * 1e3, "dtype": "f64",
           "local_lv_per_s": m * nlv / dt,
           "predictions_only": {"ms_per_step": dt_p * 1e3, "value": m / dt_p, "unit": "queries/s",
                                "note": "the same call with NULL ind / dist / w outputs (no 3.2 MB D2H of the neighbour lists; not the reference's result shape)"},
           "roofline": _roof(gather_bytes, dev_ms["local_fits"], calls, "k_locw_* (batched local weighted plskern, one workgroup per query)",
                             note="algorithmic bytes = the gathered neighbour rows m k p 8 (SURVEY §8d: the path is latency / occupancy bound, "
                                  "the HBM fraction is reported for completeness; profiles/ holds the counted traffic)"),
           "device_ms_per_step": {k_: v / calls for k_, v in dev_ms.items()}}
    ctx.set_profiling(True)
    out["knn_search"] = {"screened_queries_per_call": (ctx.counter(2) - scr0) / (3 * calls), "redone_by_the_exact_scan_per_call": (ctx.counter(3) - red0) / (3 * calls),
                         "note": "round 4: all (row, query) pairs on v_mfma_f32_32x32x16_bf16 (two-piece bf16 operands), error-bounded bar from group minima, exact f64 "
                                 "distances for the survivors; neighbours / distances / weights identical to the exact scan (JCH_KNN_SCREEN=0), tests/test_gpu_knn_screen.py"}
    out["roofline"]["traffic"] = pmc_config_traffic("cfg5_lwplsr", m=m, k=k, p=p)
    out["roofline"]["traffic_source"] = PMC_FILE + ": committed rocprofv3 --pmc passes of k_locw_kspace at this shape; NOT measured in this run"
    # the local fits are a matrix-pipe + vector kernel since round 3 (k_locw_kspace: the k x k Gram matrix of the gathered rows on
    # v_mfma_f64_16x16x4, then the LVs on it): its second bound, against the 78.6 TFLOP/s of the f64 matrix pipe
    lf_s = dev_ms["local_fits"] / calls * 1e-3
    syrk = float(m) * k * (k + 1) * p                       # algorithmic flop of the Gram matrices (one triangle)
    out["roofline_matrix"] = {"bound": "mfma", "kernel": "k_locw_kspace (whole kernel: Gram pass + latent variables)", "achieved": syrk / lf_s / 1e12,
                              "peak": 78.6, "unit": "TFLOP/s", "frac": syrk / lf_s / 1e12 / 78.6, "flop_per_launch": syrk,
                              "note": "Gram pass alone (JCH_LOCW_DBG=2 stamps, profiles/README.md): about half of the kernel's time"}
    del X, Xq, y, noise, fm
    torch.cuda.empty_cache()
    return out


def _minmax(vals):
    return {"min": float(min(vals)), "max": float(max(vals))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", "--rows", dest="n", type=int, default=1_000_000)  # (--rows: torchrun's own parser trips over --n)
    ap.add_argument("--p", type=int, default=500)
    ap.add_argument("--q", type=int, default=10)
    ap.add_argument("--nlv", type=int, default=25)
    ap.add_argument("--algo", choices=["plskern", "plsnipals", "plskern2", "plssimp", "plsrosa", "plswold"], default="plskern",
                    help="plskern2 = opt-in kernel algorithm #2 (Gram once; not the reference's algorithm, never the headline)")
    ap.add_argument("--dtype", choices=["f64", "bf16"], default="f64", help="bf16 = storage mode of BASELINE configs[2]")
    ap.add_argument("--one-pass", action="store_true", help="plsnipals / plswold: the OPT-IN one-pass variant (JCH_NIPALS_ONE_PASS; never the default)")
    ap.add_argument("--scal", action="store_true", help="scale the columns by their stds (scal = true; not the headline configuration)")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="rows of the CPU baseline run (0 = all n: no extrapolation)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the secondary host-arrays-in / host-Plsr-out timing")
    ap.add_argument("--no-other-configs", action="store_true", help="skip cfg4 / cfg3-share / cfg5 after the headline (N == 1)")
    ap.add_argument("--no-rank-share", action="store_true", help="N > 1: skip the collective-free fit of this rank's shard")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # JCH_BENCH_REHEARSAL=1: every rank on GPU 0, gloo for the host-side exchange, P2P inbox as the only transport —
    # the way to run the multi-rank code of this file on a one-GPU box (RCCL refuses two ranks on one device)
    rehearsal = os.environ.get("JCH_BENCH_REHEARSAL", "0") == "1" and world > 1
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import jchemo_hip as J
    from jchemo_hip import _lib
    lib = J.load()
    ctx = J.Context(local_rank, stream="torch")
    p2p_candidate = False
    rccl_ok = 1
    fdev = dev
    transport = "none (single GPU)"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            # the library's own RCCL communicator (dlopen'ed librccl: the copy torch loaded).  Should it fail to come up on some
            # rank, the run continues on the P2P inbox alone if THAT passes its self-test everywhere (every message of a
            # plskern-shaped fit fits the inbox), exactly like the one-GPU rehearsal; otherwise it stops with the error.
            rccl_ok = 1
            try:
                box = [J.unique_id() if rank == 0 else None]
            except Exception as e:  # noqa: BLE001
                box = [None]
                print(f"[bench] rank {rank}: RCCL unique id failed: {e}", file=sys.stderr)
            dist.broadcast_object_list(box, src=0)
            try:
                if box[0] is None:
                    raise RuntimeError("no RCCL unique id")
                ctx.comm_init(box[0], rank, world)
            except Exception as e:  # noqa: BLE001
                rccl_ok = 0
                print(f"[bench] rank {rank}: RCCL communicator failed: {e}", file=sys.stderr)
            flag = torch.tensor([rccl_ok], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            rccl_ok = int(flag.item())
        fdev = torch.device("cpu") if rehearsal else dev      # where the small agreement tensors live
        # P2P inbox transport for the latency-bound per-LV all-reduce (csrc/p2p.hip): IPC handles exchanged here, the
        # library runs a collective self-test; it is only ENABLED further down, after a whole fit through it has
        # reproduced the RCCL fit on every rank.  JCH_P2P=0 keeps RCCL for everything.
        if os.environ.get("JCH_P2P", "1") != "0" and world <= 16:
            try:
                handle = ctx.p2p_export(world)
            except Exception as e:  # noqa: BLE001
                handle = None
                print(f"[bench] rank {rank}: p2p export failed: {e}", file=sys.stderr)
            handles = [None] * world
            dist.all_gather_object(handles, handle)
            ok = False
            if all(h is not None for h in handles):
                ok = ctx.p2p_import(handles, rank, world)
                if not ok:
                    print(f"[bench] rank {rank}: p2p import/self-test failed: {getattr(ctx, 'p2p_error', '')}", file=sys.stderr)
            flag = torch.tensor([1 if ok else 0], device=fdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            p2p_candidate = bool(flag.item() == 1)
        if rehearsal or not rccl_ok:
            if not p2p_candidate:
                sys.exit("bench.py: neither the RCCL communicator nor the P2P transport came up" if not rehearsal else
                         "bench.py rehearsal: the P2P transport did not come up")
            ctx.p2p_enable(True)

    n_total, p, q, nlv = args.n, args.p, args.q, args.nlv
    row0 = (n_total * rank) // world
    n = (n_total * (rank + 1)) // world - row0
    # ---- synthetic inputs, generated in place on the device (column-major, Julia layout)
    X = J.colmajor_empty(n, p, dev)
    Y = J.colmajor_empty(n, q, dev)
    ctx.check(lib.jch_fill_uniform(ctx._h, X.data_ptr(), n, p, n, row0, n_total, 20250112))
    ctx.check(lib.jch_fill_uniform(ctx._h, Y.data_ptr(), n, q, n, row0, n_total, 20250113))
    bf16 = args.dtype == "bf16"
    if bf16:   # round-to-nearest-even to bf16 storage; the f64 originals are dropped
        Xb = J.colmajor_empty(n, p, dev, dtype=torch.bfloat16); Xb.copy_(X); X = Xb
        Yb = J.colmajor_empty(n, q, dev, dtype=torch.bfloat16); Yb.copy_(Y); Y = Yb
        del Xb, Yb
        torch.cuda.empty_cache()
    kmax = min(nlv, p, n_total)
    T = J.colmajor_empty(n, kmax, dev)
    wn = torch.empty(n, dtype=torch.float64, device=dev)
    P = np.zeros((p, kmax), order="F"); R = np.zeros((p, kmax), order="F"); W = np.zeros((p, kmax), order="F")
    Cm = np.zeros((q, kmax), order="F"); TT = np.zeros(kmax)
    xm = np.empty(p); xs = np.empty(p); ym = np.empty(q); ys = np.empty(q)
    desc = _lib.PlsDesc(n=n, p=p, q=q, nlv=nlv, scal=int(args.scal), dtype=_lib.BF16 if bf16 else _lib.F64, loc=_lib.LOC_DEVICE, inplace=0,
                        reserved=1 if args.algo == "plskern2" else (4 if (args.one_pass and args.algo in ("plsnipals", "plswold")) else 0))
    got = C.c_int32(0)
    entry = {"plsnipals": lib.jch_plsnipals_fit, "plssimp": lib.jch_plssimp_fit, "plsrosa": lib.jch_plsrosa_fit}.get(args.algo, lib.jch_plskern_fit)
    niter = np.zeros(kmax)

    def step(ctx=ctx):
        if args.algo == "plswold":   # sibling algorithm: reference defaults tol = sqrt(eps), maxit = 200
            ctx.check(lib.jch_plswold_fit(ctx._h, C.byref(desc), X.data_ptr(), n, Y.data_ptr(), n, None, float(np.sqrt(np.finfo(float).eps)), 200,
                                          T.data_ptr(), P.ctypes.data, R.ctypes.data, W.ctypes.data, Cm.ctypes.data, TT.ctypes.data,
                                          xm.ctypes.data, xs.ctypes.data, ym.ctypes.data, ys.ctypes.data, wn.data_ptr(), niter.ctypes.data,
                                          C.byref(got)))
            return
        ctx.check(entry(ctx._h, C.byref(desc), X.data_ptr(), n, Y.data_ptr(), n, None, T.data_ptr(), P.ctypes.data,
                        R.ctypes.data, W.ctypes.data, Cm.ctypes.data, TT.ctypes.data, xm.ctypes.data, xs.ctypes.data,
                        ym.ctypes.data, ys.ctypes.data, wn.data_ptr(), C.byref(got)))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    in_fit = {}

    def _coll_of(pr):   # per-LV all-reduce microseconds of one profiled fit (include/jchemo_hip.h: jch_profile collective fields)
        return {"transport": _lib.TRANSPORT_NAMES.get(pr.collective_transport, str(pr.collective_transport)),
                "us_per_lv": pr.collective_ms * 1e3 / max(pr.collective_calls, 1), "calls": pr.collective_calls,
                "polling_us_per_lv": pr.collective_wait_ms * 1e3 / max(pr.collective_calls, 1), "prologue_us": pr.prologue_collective_ms * 1e3}
    if rehearsal:
        transport = "p2p inbox only (one-GPU rehearsal)"
    elif world > 1 and not rccl_ok:
        transport = "p2p inbox over xGMI only (the library's RCCL communicator did not come up)"
    elif world > 1:
        transport = "rccl"
        step()                          # reference fit: every collective through RCCL (also RCCL's lazy channel set-up)
        ctx.set_profiling(True)
        step()
        in_fit["rccl"] = _coll_of(ctx.profile())
        ctx.set_profiling(False)
        if p2p_candidate:
            P_ref, TT_ref = P.copy(), TT.copy()
            good = 0
            try:
                ctx.p2p_enable(True)
                step()
                den = max(float(np.linalg.norm(P_ref)), 1e-300)
                good = int(np.linalg.norm(P - P_ref) <= 1e-9 * den and np.allclose(TT, TT_ref, rtol=1e-9, atol=0.0))
            except Exception as e:  # noqa: BLE001   (a bounded wait timed out: the library has switched the transport off)
                print(f"[bench] rank {rank}: fit through the p2p transport failed: {e}", file=sys.stderr)
            flag = torch.tensor([good], device=fdev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            if flag.item() == 1:
                transport = "p2p inbox over xGMI for messages <= 256 KB (per-LV [zp, tt], moments, XtY), rccl otherwise"
            else:
                try:
                    ctx.p2p_enable(False)
                except Exception:  # noqa: BLE001
                    pass
    # HIP events on the ctx stream around the sweep launches — SAMPLED: an event record costs the stream about 3 us (50 per fit were
    # 0.16 ms: 1 % of a cfg2 fit, 6 % of a 125 k-row share — tools/prof_overhead.py), so the timed fits bracket every PROF_STRIDE-th
    # launch only (the counter runs across fits: 6 is coprime to nlv = 25, every LV's sweep is timed in turn).  JCH_BENCH_PROF_STRIDE=1
    # brackets every launch as before.
    PROF_STRIDE = max(1, min(int(os.environ.get("JCH_BENCH_PROF_STRIDE", "6")), nlv))   # (<= nlv: every fit has a bracketed launch)
    if rehearsal and "JCH_BENCH_PROF_STRIDE" not in os.environ:
        PROF_STRIDE = 1   # ranks time-slicing ONE GPU: a sweep launch takes anything between 1 x and world x its own time, a sampled mean says nothing about the sum
    while PROF_STRIDE > 1 and math.gcd(PROF_STRIDE, nlv) != 1:   # coprime to nlv: over PROF_STRIDE consecutive fits every LV's sweep is bracketed exactly once
        PROF_STRIDE -= 1                                          # (a stride dividing nlv would time the SAME LVs in every fit - the first sweep after the prologue is 10 % slower than the rest)
    ctx.set_profiling(PROF_STRIDE if PROF_STRIDE > 1 else True)
    for _ in range(args.warmup):
        step()
    sweep_ms = 0.0; sweep_launches = 0; fit_ms = 0.0; prologue_ms = 0.0; small_ms = 0.0
    coll_ms = 0.0; coll_wait_ms = 0.0; coll_pro_ms = 0.0; coll_calls = 0; coll_tr = 0
    barrier()
    timed0 = timed_prev = ctx.counter(_lib.COUNTER_SWEEPS_TIMED); raw_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        pr = ctx.profile()     # host-side read of already-recorded events (the fit call is blocking)
        sweep_launches += pr.sweep_launches; fit_ms += pr.fit_ms; prologue_ms += pr.prologue_ms
        if PROF_STRIDE > 1 and pr.sweep_launches > 0:
            # sampled events: jch_profile.sweep_ms is THIS fit's estimate (mean of its bracketed launches x launches made); undo the scaling and keep the
            # bracketed launches' own sum, so that the mean below is over all bracketed launches of the timed region with equal weights
            timed1 = ctx.counter(_lib.COUNTER_SWEEPS_TIMED)
            raw_ms += pr.sweep_ms * (timed1 - timed_prev) / pr.sweep_launches; timed_prev = timed1
        else:
            sweep_ms += pr.sweep_ms; small_ms += pr.smallstate_ms
        coll_ms += pr.collective_ms; coll_wait_ms += pr.collective_wait_ms; coll_pro_ms += pr.prologue_collective_ms
        coll_calls += pr.collective_calls; coll_tr = pr.collective_transport
        sweep_bytes = pr.sweep_bytes
    barrier()
    dt = time.perf_counter() - t0
    sweeps_timed = ctx.counter(_lib.COUNTER_SWEEPS_TIMED) - timed0
    if PROF_STRIDE > 1 and sweeps_timed > 0:   # mean over the bracketed launches x the launches made; the rest of a fit is what the sweeps leave
        sweep_ms = raw_ms / sweeps_timed * sweep_launches
        small_ms = fit_ms - prologue_ms - sweep_ms
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=fdev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())

    collective = None; ranks_stats = None; rank_share = None
    if world > 1:
        import torch.distributed as dist
        mine = {"fit": fit_ms / args.steps, "prologue": prologue_ms / args.steps, "sweeps": sweep_ms / args.steps,
                "small_state_and_gaps": small_ms / args.steps, "collective": coll_ms / args.steps,
                "collective_polling": coll_wait_ms / args.steps, "prologue_collective": coll_pro_ms / args.steps, "rows": n}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        ranks_stats = {k_: _minmax([a_[k_] for a_ in allr]) for k_ in mine}
        # ---- what one all-reduce costs on this fit's own message sizes, per transport (collective probe: every rank calls it)
        ldr_b = (p + 1) & ~1
        sizes = {"per_lv_zp_tt": ldr_b + 2, "prologue_moments": p + q, "prologue_xty": p * 16}
        probes = {}; seen = {"torch_distributed_world": world}
        ri, ni = C.c_int32(-1), C.c_int32(-1)
        ctx.check(lib.jch_ctx_comm_info(ctx._h, C.byref(ri), C.byref(ni)))
        seen["jch_ctx_comm_info"] = {"rank": int(ri.value), "nranks": int(ni.value)}
        for name, code, ok_ in (("rccl", _lib.TRANSPORT_RCCL, (not rehearsal) and bool(rccl_ok)), ("inbox", _lib.TRANSPORT_INBOX, p2p_candidate)):
            if not ok_:
                probes[name] = None
                continue
            try:
                pr_ = {}
                for what, cnt in sizes.items():
                    vec, us = ctx.allreduce_probe(np.ones(cnt), code, iters=40)
                    pr_[what] = {"doubles": cnt, "us": us}
                    seen[name] = int(round(float(vec[0])))
                    if not np.all(vec == vec[0]):
                        seen[name + "_inconsistent"] = True
                probes[name] = pr_
            except Exception as e:  # noqa: BLE001
                probes[name] = {"failed": str(e)}
        collective = {"transport_timed": transport, "transport_in_fit": _lib.TRANSPORT_NAMES.get(coll_tr, str(coll_tr)), "ranks_seen": seen,
                      "per_lv_allreduce_us": {"in_timed_fits": coll_ms * 1e3 / max(coll_calls, 1), "polling_part": coll_wait_ms * 1e3 / max(coll_calls, 1),
                                              "calls_per_fit": coll_calls / max(args.steps, 1),
                                              "probe_rccl": (probes.get("rccl") or {}).get("per_lv_zp_tt", {}).get("us") if isinstance(probes.get("rccl"), dict) else None,
                                              "probe_inbox_kernel": (probes.get("inbox") or {}).get("per_lv_zp_tt", {}).get("us") if isinstance(probes.get("inbox"), dict) else None,
                                              "one_profiled_fit": in_fit},
                      "prologue_collectives_us_per_fit": coll_pro_ms * 1e3 / max(args.steps, 1), "probes": probes,
                      "how_measured": "in_timed_fits / one_profiled_fit: rank 0's jch_profile (RCCL, stand-alone inbox kernel: HIP events around each call on "
                                      "the ctx stream, so waiting for the slowest rank is included; fused inbox: wall_clock64 inside the small-state kernel); "
                                      "probe_*: 39 back-to-back all-reduces of the same message, HIP events, no compute in between"}
        # ---- this rank's shard fitted with NO communicator on the same GPU: what the fit would cost if collectives were free
        if not args.no_rank_share:
            solo = J.Context(local_rank, stream="torch")
            solo.set_profiling(PROF_STRIDE if PROF_STRIDE > 1 else True)   # (the same event sampling as the timed fits)
            step(solo)
            torch.cuda.synchronize(); ts = time.perf_counter()
            reps = max(2, min(args.steps, 5)); sfit = 0.0; ssmall = 0.0
            for _ in range(reps):
                step(solo)
                prs = solo.profile(); sfit += prs.fit_ms; ssmall += prs.smallstate_ms
            torch.cuda.synchronize(); share = (time.perf_counter() - ts) / reps
            solo.close()
            shares = [None] * world
            dist.all_gather_object(shares, {"ms": share * 1e3, "device_fit_ms": sfit / reps, "small_state_and_gaps_ms": ssmall / reps})
            smax = max(s_["ms"] for s_ in shares)
            rank_share = {"rows_per_rank": _minmax([a_["rows"] for a_ in allr]), "ms_per_fit": _minmax([s_["ms"] for s_ in shares]),
                          "device_fit_ms": _minmax([s_["device_fit_ms"] for s_ in shares]),
                          "small_state_and_gaps_ms": _minmax([s_["small_state_and_gaps_ms"] for s_ in shares]),
                          "efficiency_vs_rank_share": smax / (dt / args.steps * 1e3),
                          "note": "every rank fits its own shard alone (no communicator) on its GPU, all ranks at the same time; "
                                  "efficiency_vs_rank_share = slowest share / measured ms_per_step: 1.0 would mean the all-reduces and the skew between ranks cost nothing"
                                  + ("; REHEARSAL: the ranks share one GPU, so the shares contend and the ratio is not meaningful" if rehearsal else "")}
    if rank == 0:
        k = got.value
        value = k * args.steps / dt
        avg_sweep_s = (sweep_ms / max(sweep_launches, 1)) * 1e-3
        achieved = sweep_bytes / avg_sweep_s / 1e9 if avg_sweep_s > 0 else 0.0
        kernel = {"plskern": "k_sweep (fused t = X r, tt, zp = X'Dt, T column store)", "plsnipals": "k_sweep_lazy + k_kpass_lazy (per LV: two reads of X, rows rewritten every 6th LV; bytes_per_launch = bytes actually moved per LV)",
                  "plskern2": "k_syrk (X'DX on v_mfma_f64_16x16x4, once per fit)", "plssimp": "k_sweep (same fused sweep as plskern)",
                  "plsrosa": "k_sweep (same fused sweep as plskern)", "plswold": "k_sweep + k_deflate (per LV; q <= 4: postponed write-back as plsnipals)"}[args.algo]
        # whole-fit and prologue fractions of the HBM roof (plskern-shaped f64 fits): the one-pass prologue
        # moves 2 n p 8 bytes (read column-major X, write the row-major copy), every LV one more read of the copy
        fit_roofline = {}
        if args.algo in ("plskern", "plssimp", "plsrosa") and not bf16 and fit_ms > 0:
            xb = float(n) * p * 8.0
            fit_s, pro_s = fit_ms / args.steps * 1e-3, prologue_ms / args.steps * 1e-3
            fit_roofline = {"fit_bytes": (2 + k) * xb, "fit_frac": (2 + k) * xb / fit_s / 1e9 / HBM_PEAK_GBS,
                            "prologue_bytes": 2 * xb, "prologue_frac": 2 * xb / pro_s / 1e9 / HBM_PEAK_GBS if pro_s > 0 else None}
        out = {
            "metric": f"latent-variables/sec ({args.algo} n={n_total:.0e} p={p} q={q} nlv={nlv})".replace("e+0", "e"),
            "value": value, "unit": "LV/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline_source": "README.md:90-91 of the reference: plskern 8.10 s on an i9-10885H laptop = 3.09 LV/s (BASELINE.md §1, context only; the 10x target is vs_cpu_baseline)",
            "vs_baseline": value / README_PLSKERN_LVS if (args.algo == "plskern" and not bf16 and not args.scal and (n_total, p, q, nlv) == (1_000_000, 500, 10, 25)) else None,
            "dtype": "bf16 storage / f32 rows / f64 state" if bf16 else "f64", "data": "synthetic",
            "config": {"workload": f"{args.algo}{' (OPT-IN one-pass variant, not the reference algorithm)' if (args.one_pass and args.algo in ('plsnipals', 'plswold')) else ''} n={n_total} p={p} q={q} nlv={nlv} {'bf16-stored' if bf16 else 'Float64'} "
                                   f"({'BASELINE.json configs[1]' if (args.algo, n_total, p, q, nlv, bf16) == ('plskern', 1000000, 500, 10, 25, False) else 'variant'}), "
                                   f"X/Y device-resident column-major, rows sharded over {world} GPU(s)",
                       "n": n_total, "p": p, "q": q, "nlv": k, "rows_per_gpu": n, "timed": "prologue + LV loop, device-resident",
                       "collective_transport": transport},
            "roofline": ({"bound": "mfma", "kernel": kernel, "achieved": n * p * (p + 1) / avg_sweep_s / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                          "frac": n * p * (p + 1) / avg_sweep_s / 1e12 / 78.6, "traffic": None, "flop_per_launch": n * p * (p + 1),
                          "avg_launch_ms": avg_sweep_s * 1e3, "launches": sweep_launches} if args.algo == "plskern2" else
                         {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None if (bf16 or args.algo not in ("plskern", "plsnipals")) else pmc_traffic(args.algo, n, p),
                         "traffic_source": PMC_FILE + ": committed rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) of this kernel at this shape, scaled by rows; NOT measured in this run",
                         "bytes_per_launch": sweep_bytes, "avg_launch_ms": avg_sweep_s * 1e3, "launches": sweep_launches,
                         "launches_timed": sweeps_timed if (sweeps_timed and PROF_STRIDE > 1) else sweep_launches,
                         "timing": (f"HIP events on the launch stream around every {PROF_STRIDE}-th sweep launch of the timed fits (an event record costs the stream ~3 us; "
                                    "JCH_BENCH_PROF_STRIDE=1 brackets all of them); avg_launch_ms = mean over launches_timed") if (sweeps_timed and PROF_STRIDE > 1)
                                   else "HIP events on the launch stream around every launch of the dominant kernel in the timed fits",
                         **fit_roofline}),
            "device_ms_per_step": {"fit": fit_ms / args.steps, "prologue": prologue_ms / args.steps,
                                   "sweeps": sweep_ms / args.steps, "small_state_and_gaps": small_ms / args.steps,
                                   "collective": coll_ms / args.steps, "small_state_kernels_and_gaps": (small_ms - coll_ms) / args.steps},
        }
        if world > 1:
            out["collective"] = collective
            out["device_ms_per_step_ranks"] = ranks_stats
            if rank_share:
                out["rank_share"] = rank_share
                out["efficiency_vs_rank_share"] = rank_share["efficiency_vs_rank_share"]
        if world == 1 and not args.no_host_path and args.algo == "plskern" and not bf16:
            # SURVEY §8(d) "secondary, also reported": host arrays in -> host Plsr out (H2D of X, Y + D2H of T and the small
            # matrices inside the timed call).  PCIe-bound; never `value`.
            try:
                from oracle import c_oracle as CO
                Xh = CO.fill_uniform(20250112, n_total, p); Yh = CO.fill_uniform(20250113, n_total, q)
                J.plskern(Xh, Yh, nlv=nlv, scal=bool(args.scal), ctx=ctx)              # staging buffers, page faults
                ts = []; keep = []
                for _ in range(3):   # (the returned Plsr stays referenced: releasing the previous 0.2 GB of scores is the caller's business, not the fit's)
                    t0 = time.perf_counter(); keep.append(J.plskern(Xh, Yh, nlv=nlv, scal=bool(args.scal), ctx=ctx)); ts.append(time.perf_counter() - t0)
                th = min(ts)
                del keep
                gb = (n_total * (p + q) * 8 + n_total * (k + 1) * 8) / 1e9
                out["host_arrays"] = {"ms_per_fit": th * 1e3, "value": k / th, "unit": "LV/s", "bytes_over_pcie_gb": gb,
                                      "effective_gb_per_s": gb / th,
                                      "note": "pageable numpy arrays in, host Plsr out (score columns copied back while the LV loop runs); floor on this box = 4.16 GB H2D at the measured "
                                              "56-57 GB/s PCIe rate (73 ms) + the 15.5 ms LV loop, which needs all of X; the generator of the host inputs is oracle/ (test data only)"}
                del Xh, Yh
            except Exception as e:  # noqa: BLE001
                out["host_arrays"] = {"value": None, "note": f"failed: {e}"}
        if world == 1 and not args.no_other_configs and args.algo == "plskern" and not bf16 and (n_total, p, q, nlv) == (1_000_000, 500, 10, 25):
            # the other BASELINE.json configs that fit one GPU — outside the headline's timed region, same library, same generator
            del X, Y, T, wn
            torch.cuda.empty_cache()
            others = []
            for fn_, kw_ in ((secondary_fit, dict(label="plsnipals n=1000000 p=2000 q=1 nlv=50 Float64 (BASELINE.json configs[3])", algo="plsnipals",
                                                  n=1_000_000, p=2000, q=1, nlv=50, bf16=False, steps=2, warmup=1)),
                             (secondary_fit, dict(label="plskern n=1000000 p=500 q=10 nlv=25 bf16-stored: the one-GPU share of BASELINE.json configs[2] (n=8e6 over 8 GPUs)",
                                                  algo="plskern", n=1_000_000, p=500, q=10, nlv=25, bf16=True, steps=5, warmup=2)),
                             (secondary_lwplsr, dict(calls=20)),
                             # last, so that the indices of the three BASELINE configs above stay what earlier rounds' records use
                             (secondary_fit, dict(label="OPT-IN (JCH_NIPALS_ONE_PASS; never the default, not the reference's schedule: K updated by the exact identity "
                                                        "K_{a+1} = K_a - zp_raw c_raw'/tt instead of a second pass over X) plsnipals n=1000000 p=2000 q=1 nlv=50 Float64",
                                                  algo="plsnipals", n=1_000_000, p=2000, q=1, nlv=50, bf16=False, steps=2, warmup=1, reserved=4))):
                try:
                    others.append(fn_(J, _lib, lib, ctx, dev, **kw_) if fn_ is secondary_fit else fn_(J, lib, ctx, dev, **kw_))
                except Exception as e:  # noqa: BLE001   (never take the headline down)
                    others.append({"config": kw_.get("label", "lwplsr cfg5"), "value": None, "note": f"failed: {e}"})
            out["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(n_total, p, q, nlv, args.cpu_sample_rows or n_total)
                if out["cpu_baseline"].get("value"):
                    out["vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]   # the ratio the north star's 10x target is about
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "LV/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
