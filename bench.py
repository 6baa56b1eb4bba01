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
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "jchemo.jl_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
README_PLSKERN_LVS = 25 / 8.100469   # README.md:90-91: plskern n=1e6 p=500 q=10 nlv=25 in 8.10 s (i9-10885H)


PMC_FILE = "profiles/r02_pmc_sweep.json"


def pmc_traffic(algo, n_local, p):
    """HBM bytes per sweep launch from the COMMITTED PMC pass (profiles/r02_pmc_sweep.json: FETCH_SIZE x2 gfx950 correction
    + WRITE_SIZE, separate rocprofv3 passes, tools/final_pmc_r02.sh), scaled by rows when this rank holds a different
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


def _blas_threads():
    """Threads numpy's BLAS (OpenBLAS) will use — what Julia's `BLAS.get_num_threads()` reports for the reference."""
    try:
        from threadpoolctl import threadpool_info
        info = [i for i in threadpool_info() if i.get("user_api") == "blas"]
        return max((int(i.get("num_threads", 0)) for i in info), default=0), ",".join(sorted({str(i.get("internal_api")) for i in info}))
    except Exception:
        return 0, "unknown"


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
            "host_cpus": os.cpu_count(), "threads": {"openblas": blas_threads, "blas": blas_name, "c_port_openmp": omp_threads},
            "lv_per_s": lvs, "seconds": {k_: [round(t, 3) for t in v] for k_, v in runs.items()},
            "sample": (f"plskern! (in place, README.md:93) on {'all' if ns == n_total else 'the first'} {ns} of {n_total} rows "
                       f"(p={p}, q={q}, nlv={nlv}), two runs per stand-in, best run reported"
                       + ("" if ns == n_total else f", time scaled x{scale:.2f}") + f"; {probe}")}


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
    ap.add_argument("--scal", action="store_true", help="scale the columns by their stds (scal = true; not the headline configuration)")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="rows of the CPU baseline run (0 = all n: no extrapolation)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the secondary host-arrays-in / host-Plsr-out timing")
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
                        reserved=1 if args.algo == "plskern2" else 0)
    got = C.c_int32(0)
    entry = {"plsnipals": lib.jch_plsnipals_fit, "plssimp": lib.jch_plssimp_fit, "plsrosa": lib.jch_plsrosa_fit}.get(args.algo, lib.jch_plskern_fit)
    niter = np.zeros(kmax)

    def step():
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

    if rehearsal:
        transport = "p2p inbox only (one-GPU rehearsal)"
    elif world > 1 and not rccl_ok:
        transport = "p2p inbox over xGMI only (the library's RCCL communicator did not come up)"
    elif world > 1:
        transport = "rccl"
        step()                          # reference fit: every collective through RCCL
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
    ctx.set_profiling(True)        # HIP events on the ctx stream around every sweep launch
    for _ in range(args.warmup):
        step()
    sweep_ms = 0.0; sweep_launches = 0; fit_ms = 0.0; prologue_ms = 0.0; small_ms = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        pr = ctx.profile()     # host-side read of already-recorded events (the fit call is blocking)
        sweep_ms += pr.sweep_ms; sweep_launches += pr.sweep_launches; fit_ms += pr.fit_ms
        prologue_ms += pr.prologue_ms; small_ms += pr.smallstate_ms
        sweep_bytes = pr.sweep_bytes
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=fdev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())

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
            "config": {"workload": f"{args.algo} n={n_total} p={p} q={q} nlv={nlv} {'bf16-stored' if bf16 else 'Float64'} "
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
                         **fit_roofline}),
            "device_ms_per_step": {"fit": fit_ms / args.steps, "prologue": prologue_ms / args.steps,
                                   "sweeps": sweep_ms / args.steps, "small_state_and_gaps": small_ms / args.steps},
        }
        if world == 1 and not args.no_host_path and args.algo == "plskern" and not bf16:
            # SURVEY §8(d) "secondary, also reported": host arrays in -> host Plsr out (H2D of X, Y + D2H of T and the small
            # matrices inside the timed call).  PCIe-bound; never `value`.
            try:
                from oracle import c_oracle as CO
                Xh = CO.fill_uniform(20250112, n_total, p); Yh = CO.fill_uniform(20250113, n_total, q)
                J.plskern(Xh, Yh, nlv=nlv, scal=bool(args.scal), ctx=ctx)              # staging buffers, page faults
                ts = []
                for _ in range(2):
                    t0 = time.perf_counter(); J.plskern(Xh, Yh, nlv=nlv, scal=bool(args.scal), ctx=ctx); ts.append(time.perf_counter() - t0)
                th = min(ts)
                gb = (n_total * (p + q) * 8 + n_total * (k + 1) * 8) / 1e9
                out["host_arrays"] = {"ms_per_fit": th * 1e3, "value": k / th, "unit": "LV/s", "bytes_over_pcie_gb": gb,
                                      "effective_gb_per_s": gb / th, "note": "pageable numpy arrays in, host Plsr out; the generator of the host inputs is oracle/ (test data only)"}
                del Xh, Yh
            except Exception as e:  # noqa: BLE001
                out["host_arrays"] = {"value": None, "note": f"failed: {e}"}
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
