#!/bin/bash
# round-4 quick check on the GPU box: full -m gpu suite, headline bench, 125 k-row share, LV debug stamps
O=gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q > $O/r04_gpu_tests.log 2>&1; tail -6 $O/r04_gpu_tests.log | cut -c1-300
python bench.py --no-cpu-baseline --no-host-path > $O/r04_bench_q.json 2> $O/r04_bench_q.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_q.json').read().strip().splitlines()[-1])
print("headline", round(d['value'],1), d['device_ms_per_step'])
for o in d.get('other_configs',[]): print(o['config']['workload'][:40] if 'workload' in o.get('config',{}) else o.get('config'), round(o['value'],1), o.get('device_ms_per_step'))
PY
python bench.py --rows 125000 --steps 20 --warmup 3 --no-cpu-baseline --no-host-path --no-other-configs > $O/r04_share125.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r04_share125.json').read().strip().splitlines()[-1]); print('share125', d['ms_per_step'], d['device_ms_per_step'])"
JCH_LV_DEBUG=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs > /dev/null 2> $O/r04_lvdebug.err; grep -m5 "jch" $O/r04_lvdebug.err
