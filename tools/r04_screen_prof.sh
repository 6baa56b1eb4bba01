#!/bin/bash
# kernel-trace stats of one cfg5 run with the screened kNN
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/screen_prof -- python3 $R/tools/bench_lwplsr.py > $R/gpurun_out/screen_prof.log 2>&1 || exit 1
cd $R && python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/screen_prof/**/*kernel_stats.csv", recursive=True)
for row in list(csv.DictReader(open(f[0])))[:16]:
    print(row["Name"][:80].ljust(80), row["Calls"], row["AverageNs"], row["Percentage"])
PY
