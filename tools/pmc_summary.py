"""Summarise rocprofv3 --pmc passes (tools/pmc_pass.sh): per kernel, mean counter value per dispatch and mean duration.
usage: python tools/pmc_summary.py gpurun_out/<tag> [name-substring ...]"""
import csv, glob, os, sys, collections, json
root = sys.argv[1]; pats = sys.argv[2:]
out = collections.defaultdict(dict)
for g in sorted(glob.glob(os.path.join(root, "g*"))):
    if not os.path.isdir(g):
        continue
    cc = sorted(glob.glob(os.path.join(g, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]
    kt = sorted(glob.glob(os.path.join(g, "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1:]
    if not cc:
        continue
    dur = collections.defaultdict(list)
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        short = k.split("(")[0]
        if pats and not any(p in short for p in pats):
            continue
        for c, v in cs.items():
            out[short][c] = sum(v) / len(v)
        if dur.get(k):
            out[short]["_us"] = sum(dur[k]) / len(dur[k]); out[short]["_n"] = len(dur[k])
for k, v in out.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"    {c:48s} {x:16.1f}")
