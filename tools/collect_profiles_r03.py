"""Copy the round-3 evidence from gpurun_out/final3 (tools/final_profiles_r03.sh) into profiles/r03_* and derive
profiles/r03_pmc_sweep.json (HBM bytes per launch of the dominant kernels, read by bench.py's roofline.traffic)."""
import csv, glob, json, os, shutil, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final3"); DST = os.path.join(ROOT, "profiles")
def newest(pattern):
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]
def cp(a, b):
    pa = os.path.join(SRC, a)
    if os.path.exists(pa) and os.path.getsize(pa) > 0:
        shutil.copy(pa, os.path.join(DST, b)); return True
    print("missing", a); return False
for a, b in [("bench.json", "r03_final_bench.json"), ("bench_cfg4.json", "r03_final_bench_cfg4_plsnipals.json"), ("bench_bf16.json", "r03_final_bench_bf16.json"),
             ("bench_bf16_n8e6_one_gpu.json", "r03_final_bench_bf16_n8e6_one_gpu.json"), ("bench_rank_share_125k.json", "r03_rank_share_125k_rows.json"),
             ("bench_scal.json", "r03_final_bench_scal_true.json"), ("bench_plsnipals_q10.json", "r03_bench_plsnipals_q10_cfg2_shape.json"),
             ("lwplsr_cfg5.json", "r03_lwplsr_cfg5.json"), ("lwplsr_cfg5_3replicas_one_gpu.json", "r03_lwplsr_cfg5_3replicas_one_gpu_rehearsal.json"),
             ("gridcv.json", "r03_gridcvlv_cfg2.json"), ("accessors.json", "r03_accessors_cfg2.json"), ("pmc_summary.txt", "r03_pmc_headline_all_kernels.txt"),
             ("pmc_bf16_summary.txt", "r03_pmc_bf16_traffic.txt"), ("pmc_lwplsr_summary.txt", "r03_pmc_lwplsr_cfg5.txt")]:
    cp(a, b)
for a in ("plssimp", "plsrosa", "plswold", "plskern2"):
    cp(f"bench_{a}.json", f"r03_sibling_bench_{a}.json" if a != "plskern2" else "r03_final_bench_plskern2_optin.json")
with open(os.path.join(DST, "r03_rank_share_table.jsonl"), "w") as f:
    for nm in ("bench.json", "bench_rank_share_500000.json", "bench_rank_share_250000.json", "bench_rank_share_125k.json"):
        p_ = os.path.join(SRC, nm)
        if os.path.exists(p_) and os.path.getsize(p_) > 0:
            d = json.load(open(p_)); f.write(json.dumps({"rows_per_gpu": d["config"]["rows_per_gpu"], "LV_per_s": d["value"], "device_ms_per_step": d["device_ms_per_step"], "sweep_GBps": d["roofline"]["achieved"]}) + "\n")
for tag, out in (("stats", "r03_final_kernel_stats.csv"), ("stats_cfg4", "r03_final_kernel_stats_cfg4_plsnipals.csv"), ("stats_bf16", "r03_final_kernel_stats_bf16.csv"),
                 ("stats_lwplsr", "r03_final_kernel_stats_lwplsr_cfg5.csv")):
    fs = newest(os.path.join(SRC, tag, "*", "*kernel_stats.csv"))
    if fs: shutil.copy(fs[0], os.path.join(DST, out))
def pmc(tag):
    res = collections.defaultdict(dict)
    for g in sorted(glob.glob(os.path.join(SRC, tag, "g*"))):
        if not os.path.isdir(g): continue
        cc = newest(os.path.join(g, "*", "*_counter_collection.csv")); kt = newest(os.path.join(g, "*", "*_kernel_trace.csv"))
        if not cc: continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
        for r in csv.DictReader(open(cc[0])): acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if kt:
            for r in csv.DictReader(open(kt[0])): dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
        for k, cs in acc.items():
            for c, v in cs.items(): res[k][c] = sum(v) / len(v)
            if dur.get(k): res[k]["_us"] = sum(dur[k]) / len(dur[k]); res[k]["_n"] = len(dur[k])
    return res
def traffic(r):  # FETCH_SIZE / WRITE_SIZE are KiB; gfx950 correction: wide coalesced reads are tallied at half their bytes
    return 2.0 * r.get("FETCH_SIZE", 0.0) * 1024 + r.get("WRITE_SIZE", 0.0) * 1024
n, p = 1_000_000, 500
head = pmc("pmc"); b16 = pmc("pmc_bf16")
out = {"command": "rocprofv3 --pmc <group> --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs (one counter group per pass, tools/final_profiles_r03.sh)",
       "workload": {"algo": "plskern", "n": n, "p": p, "q": 10, "nlv": 25},
       "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads (MI355X_MICROARCH.md HBM section) -> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE exact",
       "kernels": {}}
alg = {"void k_sweep_v2<4, 8, 2>": n * p * 8 + 16 * n, "void k_center_xty_panel<64, 64, false, false>": 2 * n * p * 8 + n * 10 * 8 + n * 16 * 8 + 8 * n}
algb = {"void k_center_xty_bf16_panel<8, 2, false, 0>": n * p * 2 + n * 504 * 2 + n * 10 * 2 + n * 16 * 8 + 8 * n, "void k_sweep_bf16_v2<1, 8>": n * 504 * 2 + 16 * n}
for src_, table in ((head, alg), (b16, algb)):
    for k, r in src_.items():
        if k in table:
            t = traffic(r)
            out["kernels"][k] = {"FETCH_SIZE_avg_KiB": r.get("FETCH_SIZE"), "WRITE_SIZE_avg_KiB": r.get("WRITE_SIZE"), "hbm_bytes_per_launch": t, "algorithmic_bytes_per_launch": table[k],
                                 "ratio": t / table[k], "avg_us_under_pmc": r.get("_us"), "mfma_mops_f64": r.get("SQ_INSTS_VALU_MFMA_MOPS_F64"),
                                 "mfma_busy_cycles": r.get("SQ_VALU_MFMA_BUSY_CYCLES"), "sq_busy_cycles": r.get("SQ_BUSY_CYCLES")}
sw = out["kernels"].get("void k_sweep_v2<4, 8, 2>")
if sw:
    out["kernel"] = "void k_sweep_v2<4, 8, 2>"; out["hbm_bytes_per_launch"] = sw["hbm_bytes_per_launch"]; out["algorithmic_bytes_per_launch"] = sw["algorithmic_bytes_per_launch"]
    json.dump(out, open(os.path.join(DST, "r03_pmc_sweep.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1)[:2500])
