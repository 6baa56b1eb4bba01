#!/bin/bash
# device timeline of one cfg5 predict call: kernels + memory copies (no counters)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lw_timeline; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python3 $R/tools/bench_lwplsr.py > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
cd $R && python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob('gpurun_out/lw_timeline/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)): ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:44]))
for f in glob.glob('gpurun_out/lw_timeline/*/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)): ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', r.get('Name', ''))[:30]))
ev.sort()
idx = [i for i, e in enumerate(ev) if 'k_locw_kspace' in e[2]]
i = idx[-2]
t0 = ev[i - 9][0]
for s, e, nme in ev[i - 9:i + 8]:
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  {nme}")
PY
