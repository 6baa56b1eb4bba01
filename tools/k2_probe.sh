#!/bin/bash
for m in 0 1 2 3; do
  JCH_K2_SKIP=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('K2_SKIP=$m prologue ms %.3f'%d['device_ms_per_step']['prologue'])"
done
JCH_SWEEP_BLOCKS_PER_CU=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --n 2000000 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('n=2e6', d['value'], d['roofline']['achieved'], d['device_ms_per_step'])"
