#!/bin/bash
# kernel-trace stats of cfg5 runs with the screened kNN, under the knobs given as arguments ("A=1 B=2" per argument)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for kv in "$@"; do
  i=$((i+1))
  export $kv
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/screen_prof$i -- python3 $R/tools/bench_lwplsr.py > $R/gpurun_out/screen_prof$i.log 2>&1 || { tail -5 $R/gpurun_out/screen_prof$i.log; exit 1; }
  for v in $kv; do unset ${v%%=*}; done
  echo "== $kv"; grep "screened kNN" $R/gpurun_out/screen_prof$i.log | head -1
  python3 - $R/gpurun_out/screen_prof$i <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for row in csv.DictReader(open(f[0])):
    if any(t in row["Name"] for t in ("k_knn", "k_ks_", "k_locw")):
        print("  ", row["Name"][:60].ljust(60), row["Calls"], round(float(row["AverageNs"]) / 1e3, 1), "us")
PY
done
