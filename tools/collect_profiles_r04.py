"""Copy the round-4 evidence from gpurun_out/final4 (tools/final_profiles_r04.sh, tools/final_pmc_r04.sh) into profiles/r04_* and
derive profiles/r04_pmc_traffic.json (HBM bytes per launch of the dominant kernels of EVERY bench configuration, read by
bench.py's roofline.traffic fields)."""
import csv, glob, json, os, shutil, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final4"); DST = os.path.join(ROOT, "profiles")
def newest(pattern):
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]
def cp(a, b):
    pa = os.path.join(SRC, a)
    if os.path.exists(pa) and os.path.getsize(pa) > 0:
        shutil.copy(pa, os.path.join(DST, b)); return True
    print("missing", a); return False
for a, b in [("bench.json", "r04_final_bench.json"), ("bench_cfg4.json", "r04_final_bench_cfg4_plsnipals.json"), ("bench_bf16.json", "r04_final_bench_bf16.json"),
             ("bench_bf16_n8e6_one_gpu.json", "r04_final_bench_bf16_n8e6_one_gpu.json"), ("bench_rank_share_125k.json", "r04_rank_share_125k_rows.json"),
             ("bench_rank_share_125k_one_kernel_small_state.json", "r04_rank_share_125k_rows_one_kernel_small_state.json"),
             ("bench_one_kernel_small_state.json", "r04_bench_one_kernel_small_state.json"),
             ("bench_scal.json", "r04_final_bench_scal_true.json"), ("bench_plsnipals_q10.json", "r04_bench_plsnipals_q10_cfg2_shape.json"),
             ("bench_plsnipals_q10_four_waves.json", "r04_bench_plsnipals_q10_cfg2_shape_four_waves.json"),
             ("lwplsr_cfg5.json", "r04_lwplsr_cfg5.json"), ("gridcv.json", "r04_gridcvlv_cfg2.json"), ("accessors.json", "r04_accessors_cfg2.json"),
             ("pmc_summary.txt", "r04_pmc_headline_all_kernels.txt"), ("pmc_bf16_summary.txt", "r04_pmc_bf16_traffic.txt"),
             ("pmc_cfg4_summary.txt", "r04_pmc_cfg4_traffic.txt"), ("pmc_nipals_q10_summary.txt", "r04_pmc_plsnipals_q10.txt"),
             ("pmc_lwplsr_summary.txt", "r04_pmc_lwplsr_cfg5.txt"), ("lv_debug_split.err", "r04_lv_debug_stamps_split.txt"),
             ("lv_debug_one_kernel.err", "r04_lv_debug_stamps_one_kernel.txt"),
             ("bench_plsnipals_q10_one_pass_optin.json", "r04_bench_plsnipals_q10_one_pass_optin.json"),
             ("bench_plswold_one_pass_optin.json", "r04_bench_plswold_one_pass_optin.json"),
             ("bench_cfg4_one_pass_optin.json", "r04_bench_cfg4_one_pass_optin.json")]:
    cp(a, b)
for a in ("plssimp", "plsrosa", "plswold", "plskern2"):
    cp(f"bench_{a}.json", f"r04_sibling_bench_{a}.json" if a != "plskern2" else "r04_final_bench_plskern2_optin.json")
with open(os.path.join(DST, "r04_rank_share_table.jsonl"), "w") as f:
    for nm in ("bench.json", "bench_rank_share_500000.json", "bench_rank_share_250000.json", "bench_rank_share_125k.json"):
        p_ = os.path.join(SRC, nm)
        if os.path.exists(p_) and os.path.getsize(p_) > 0:
            d = json.loads(open(p_).read().strip().splitlines()[-1])
            f.write(json.dumps({"rows_per_gpu": d["config"]["rows_per_gpu"], "LV_per_s": d["value"], "device_ms_per_step": d["device_ms_per_step"], "sweep_GBps": d["roofline"]["achieved"]}) + "\n")
for tag, out in (("stats", "r04_final_kernel_stats.csv"), ("stats_cfg4", "r04_final_kernel_stats_cfg4_plsnipals.csv"), ("stats_bf16", "r04_final_kernel_stats_bf16.csv"),
                 ("stats_lwplsr", "r04_final_kernel_stats_lwplsr_cfg5.csv"), ("stats_nipals_q10", "r04_final_kernel_stats_plsnipals_q10.csv")):
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
def entry(r, alg):
    t = traffic(r)
    return {"FETCH_SIZE_avg_KiB": r.get("FETCH_SIZE"), "WRITE_SIZE_avg_KiB": r.get("WRITE_SIZE"), "hbm_bytes_per_launch": t, "algorithmic_bytes_per_launch": alg,
            "ratio": (t / alg) if alg else None, "avg_us_under_pmc": r.get("_us"), "launches": r.get("_n"), "mfma_mops_f64": r.get("SQ_INSTS_VALU_MFMA_MOPS_F64"),
            "mfma_busy_cycles": r.get("SQ_VALU_MFMA_BUSY_CYCLES"), "sq_busy_cycles": r.get("SQ_BUSY_CYCLES"), "sq_wave_cycles": r.get("SQ_WAVE_CYCLES"),
            "sq_wait_inst_any": r.get("SQ_WAIT_INST_ANY"), "sq_wait_any": r.get("SQ_WAIT_ANY")}
n, p = 1_000_000, 500
out = {"command": "rocprofv3 --pmc <group> --kernel-trace -- python bench.py <config> (ONE counter group per pass: tools/final_pmc_r04.sh, tools/pmc_pass.sh)",
       "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads (MI355X_MICROARCH.md HBM section) -> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE exact",
       "workload": {"algo": "plskern", "n": n, "p": p, "q": 10, "nlv": 25}, "kernels": {}, "configs": {}}
head, b16, c4, nq, lw = pmc("pmc"), pmc("pmc_bf16"), pmc("pmc_cfg4"), pmc("pmc_nipals_q10"), pmc("pmc_lwplsr")
alg_head = {"void k_sweep_v2<4, 8, 2>": n * p * 8 + 16 * n, "void k_center_xty_panel<64, 64, false, false>": 2 * n * p * 8 + n * 10 * 8 + n * 16 * 8 + 8 * n}
for k, r in head.items():
    if k in alg_head or k.startswith("void k_lv_") or k.startswith("k_lv_"): out["kernels"][k] = entry(r, alg_head.get(k))
alg_b = {"void k_xty_bf16_panel_m32<8, 2>": n * p * 2 + n * 504 * 2 + n * 10 * 2, "void k_sweep_bf16_v2<1, 8>": n * 504 * 2 + 16 * n}
for k, r in b16.items():
    if k in alg_b: out["kernels"][k] = entry(r, alg_b[k])
n4, p4 = 1_000_000, 2000
for k, r in c4.items():
    if "k_sweep_lazy" in k: out["kernels"][k + " [cfg4]"] = entry(r, n4 * p4 * 8 + 24 * n4)
    if "k_kpass_lazy" in k: out["kernels"][k + " [cfg4: read-only and flushing launches averaged]"] = entry(r, None)
for k, r in nq.items():
    if "k_sweep_lazy" in k: out["kernels"][k + " [plsnipals q=10, cfg2 shape]"] = entry(r, n * p * 8 + 24 * n + n * 16 * 8)
    if "k_kpass_mfma_lazy" in k: out["kernels"][k + " [plsnipals q=10, cfg2 shape: read-only and flushing launches averaged]"] = entry(r, None)
for k, r in lw.items():
    if "k_locw" in k or "k_knn" in k: out["kernels"][k + " [cfg5, per 1000 queries]"] = entry(r, 1000 * 200 * 500 * 8 if "k_locw" in k else None)
# per bench configuration: bytes per "launch" in the unit bench.py's roofline blocks use
sw = out["kernels"].get("void k_sweep_v2<4, 8, 2>")
if sw:
    out["kernel"] = "void k_sweep_v2<4, 8, 2>"; out["hbm_bytes_per_launch"] = sw["hbm_bytes_per_launch"]; out["algorithmic_bytes_per_launch"] = sw["algorithmic_bytes_per_launch"]
def first(d, pat):
    for k, v in d.items():
        if pat in k: return v
    return None
s4, k4 = first(c4, "k_sweep_lazy"), first(c4, "k_kpass_lazy")
if s4 and k4:   # per LV: one sweep launch + one pass launch (the pass average already contains its share of flushing launches)
    out["configs"]["cfg4_plsnipals"] = {"unit": "bytes per LV (k_sweep_lazy + k_kpass_lazy, flushes averaged in)", "hbm_bytes_per_launch": traffic(s4) + traffic(k4), "n": n4, "p": p4}
sb = first(b16, "k_sweep_bf16_v2")
if sb: out["configs"]["bf16_share"] = {"unit": "bytes per sweep launch (k_sweep_bf16_v2)", "hbm_bytes_per_launch": traffic(sb), "n": n, "p": p}
lk = first(lw, "k_locw")
if lk: out["configs"]["cfg5_lwplsr"] = {"unit": "bytes per call of 1000 queries (k_locw_kspace)", "hbm_bytes_per_launch": traffic(lk), "m": 1000, "k": 200, "p": 500}
json.dump(out, open(os.path.join(DST, "r04_pmc_traffic.json"), "w"), indent=1)
for k, v in out["kernels"].items(): print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("hbm_bytes_per_launch", "ratio", "avg_us_under_pmc")})
print(out["configs"])
