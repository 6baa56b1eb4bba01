#!/bin/bash
# first run of the screened kNN: its tests, the existing lwplsr tests, then cfg5 per call with and without it + kernel stats
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_knn_screen.py -x -q -m gpu 2>&1 | tail -15 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_envelope.py tests/test_gpu_fullsize.py -x -q -m gpu -k "lwplsr or knn" 2>&1 | tail -5 || exit 1
for kv in "JCH_X=1" "JCH_KNN_SCREEN=0"; do
  echo "== $kv"; env $kv timeout -k 10 300 python tools/bench_lwplsr.py 2>&1 | tail -2
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/screen_prof -o screen -- python3 $GRAFT_REPO_ROOT/tools/bench_lwplsr.py > $GRAFT_REPO_ROOT/gpurun_out/screen_prof.log 2>&1
cd $GRAFT_REPO_ROOT && python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/screen_prof/**/*kernel_stats.csv", recursive=True)
for row in list(csv.DictReader(open(f[0])))[:14]:
    print(row["Name"][:70].ljust(70), row["Calls"], row["AverageNs"], row["Percentage"])
PY
