#!/bin/bash
# sweep variants: device ms of the 25 sweeps of the cfg2 bench under different JCH_SWEEP_* settings; usage: sweep_variants.sh <rounds> [bench args --] "ENV=.." ...
rounds=$1; shift
extra=""
if [ "$1" = "--args" ]; then extra="$2"; shift; shift; fi
for r in $(seq $rounds); do
for v in "$@"; do
  env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-path $extra 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('round $r', '$v', 'LV/s %.1f'%d['value'], 'sweeps %.3f'%d['device_ms_per_step']['sweeps'], 'GB/s %.0f'%d['roofline']['achieved'], 'small %.3f'%d['device_ms_per_step']['small_state_and_gaps'], 'fit ms %.3f'%d['device_ms_per_step']['fit'])"
done
done
