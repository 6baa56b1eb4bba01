#!/bin/bash
# K2 variants: prologue ms of the cfg2 bench under different JCH_K2_* settings; usage: k2_variants.sh <rounds> "ENV=.." ...
rounds=$1; shift
for r in $(seq $rounds); do
for v in "$@"; do
  env $v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('round $r', '$v', 'LV/s %.1f'%d['value'], 'prologue ms %.3f'%d['device_ms_per_step']['prologue'], 'sweeps %.3f'%d['device_ms_per_step']['sweeps'], 'fit ms %.3f'%d['device_ms_per_step']['fit'])"
done
done
