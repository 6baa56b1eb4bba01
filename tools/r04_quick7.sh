#!/bin/bash
O=gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q -k "lwplsr or envelope or knn or cfg5" > $O/r04_gpu_tests_i.log 2>&1; tail -4 $O/r04_gpu_tests_i.log | cut -c1-300
for mode in 0 1 2 3; do echo "== JCH_KNN_SCAN8=$mode"; JCH_KNN_SCAN8=$mode python tools/bench_lwplsr.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['gpu_ms_per_call'],3), 'ms', d['neighbours_equal'], d['parity_pred_rel_fro_on_sample'])"; done
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/scan8_stats -- python $R/tools/bench_lwplsr.py > $R/$O/scan8_stats.log 2>&1
cd $R; python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/scan8_stats/*/*kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:50].ljust(50), r['Calls'].rjust(4), '%.1f us'%(float(r['AverageNs'])/1e3))
PY
