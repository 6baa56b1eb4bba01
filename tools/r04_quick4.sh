#!/bin/bash
O=gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q > $O/r04_gpu_tests_d.log 2>&1; tail -5 $O/r04_gpu_tests_d.log | cut -c1-300
python tools/bench_accessors.py 2>/dev/null | tail -1
JCH_GEMM_WIDEOUT=0 python tools/bench_accessors.py 2>/dev/null | tail -1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/r04_bench_q4.json 2>/dev/null; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_q4.json').read().strip().splitlines()[-1])
print("headline", round(d['value'],1), d['device_ms_per_step'], d['roofline']['traffic'])
for o in d.get('other_configs',[]): print(str(o.get('config'))[:50], round(o['value'],1), o['roofline'].get('traffic'), o.get('predictions_only'))
PY
