#!/bin/bash
# rocprofv3 kernel stats of tools/bench_lwplsr.py (cfg5) -> gpurun_out/lwplsr_stats
export TMPDIR=/tmp; R=$PWD; out=$R/gpurun_out/lwplsr_stats; mkdir -p $out; cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $R/tools/bench_lwplsr.py > $out/run.log 2>&1
echo "exit $?"; tail -1 $out/run.log | cut -c1-300
f=$(ls -t $out/stats/*/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.3f} {r['Percentage']:>6s}%")
PY
