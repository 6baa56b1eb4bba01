#!/bin/bash
# rocprofv3 kernel-trace stats of one bench run: tools/kstats.sh <tag> <bench.py args...>  -> gpurun_out/<tag>/ + top kernels on stdout
export TMPDIR=/tmp; R=$PWD; tag=$1; shift
mkdir -p $R/gpurun_out/$tag; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/stats -- python $R/bench.py "$@" > $R/gpurun_out/$tag/stats.log 2>&1
echo "exit $?"
f=$(ls $R/gpurun_out/$tag/stats/*/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.3f} {r['Percentage']:>6s}%")
PY
