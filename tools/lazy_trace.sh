#!/bin/bash
# per-launch durations of the lazy NIPALS kernels at cfg4 (kernel trace): tools/lazy_trace.sh <m> [NLV]
export TMPDIR=/tmp; R=$PWD; m=${1:-9}; export NLV=${2:-19}
out=$R/gpurun_out/lazy_trace_m$m; mkdir -p $out; cd /tmp
export DEFL_VARIANTS="JCH_NIPALS_DEFER=$m"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python $R/tools/deflate_modes.py > $out/run.log 2>&1
echo "exit $?"; tail -2 $out/run.log
f=$(ls -t $out/tr/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'lazy' in r['Kernel_Name'] or 'k_deflate_stream' in r['Kernel_Name'] or r['Kernel_Name'].startswith('void k_sweep<')]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
nl=len(rows)
last=rows[-(nl//4):]    # the last of the four fits
for r in last:
    print(f"{r['Kernel_Name'][:40]:40s} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:8.3f} ms")
PY
