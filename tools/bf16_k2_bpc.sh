#!/bin/bash
# bf16 prologue (K2) under different blocks-per-CU settings
for b in 1 2 3 4 6 8; do
  JCH_BF16_K2_BPC=$b python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('K2_BPC=$b', 'LV/s %.1f'%d['value'], d['device_ms_per_step'])"
done
