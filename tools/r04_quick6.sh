#!/bin/bash
O=gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q -k "one_pass or accessor or long_input or transform or predict or golden or xfit or gridcv or scores or postponed" > $O/r04_gpu_tests_f.log 2>&1; tail -4 $O/r04_gpu_tests_f.log | cut -c1-300
python tools/bench_accessors.py 2>/dev/null | tail -1
JCH_GEMM_WIDEOUT_RT=2 python tools/bench_accessors.py 2>/dev/null | tail -1
for a in plsnipals plswold; do python bench.py --algo $a --one-pass --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-other-configs > $O/r04_bench_${a}_onepass.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/r04_bench_${a}_onepass.json').read().strip().splitlines()[-1]); print('$a one-pass', round(d['value'],1), d['device_ms_per_step'])"
done
python bench.py --algo plsnipals --one-pass --p 2000 --q 1 --nlv 50 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs > $O/r04_bench_cfg4_onepass.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/r04_bench_cfg4_onepass.json').read().strip().splitlines()[-1]); print('cfg4 one-pass', round(d['value'],1), d['device_ms_per_step'])"
