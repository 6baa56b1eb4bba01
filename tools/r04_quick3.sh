#!/bin/bash
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -k "split_small or postponed or golden or sibling or fuzz or nipals or wold or row_sharded" > $O/r04_gpu_tests_c.log 2>&1; tail -4 $O/r04_gpu_tests_c.log | cut -c1-300
for nw in 4 8; do for a in plsnipals plswold; do
JCH_KPASS_NW=$nw python bench.py --algo $a --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-other-configs > $O/r04_bench_${a}_nw$nw.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/r04_bench_${a}_nw$nw.json').read().strip().splitlines()[-1]); print('$a NW=$nw', round(d['value'],1), d['device_ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done
python bench.py --no-cpu-baseline --no-host-path --no-other-configs > $O/r04_bench_q3.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/r04_bench_q3.json').read().strip().splitlines()[-1]); print('headline', round(d['value'],1), d['device_ms_per_step'])"
