#!/bin/bash
# A/B: the first sweep of a fit walked backwards (JCH_SWEEP_FIRST_REV = 0 / 1 / 2); nlv = 3 makes the first sweep a third of the sweeps
O=gpurun_out/first_rev; mkdir -p $O
F="--no-cpu-baseline --no-host-path --no-other-configs"
export JCH_BENCH_PROF_STRIDE=1
for rep in 1 2; do
for rows in 1000000 125000; do
for nlv in 3 25; do
for m in 0 1 2; do
  JCH_SWEEP_FIRST_REV=$m python bench.py --rows $rows --nlv $nlv --steps 30 --warmup 5 $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); dm=d['device_ms_per_step']
print('rows $rows nlv $nlv mode $m: ms/step %.4f fit %.4f prologue %.4f sweeps %.4f small %.4f' % (d['ms_per_step'], dm['fit'], dm['prologue'], dm['sweeps'], dm['small_state_and_gaps']))" | tee -a $O/ab.log
done; done; done; done
