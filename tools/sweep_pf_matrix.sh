#!/bin/bash
# prefetched vs plain sweep kernels over the row widths (KC = 1, 2, 4, 8, 16); prints LV/s and the sweep's GB/s
for shape in "100 4 10" "200 4 10" "500 10 25" "1000 4 12" "2000 1 10"; do
  set -- $shape
  for pf in 0 1; do
    JCH_SWEEP_PF=$pf timeout -k 10 200 python bench.py --rows 1000000 --p $1 --q $2 --nlv $3 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('p=$1 PF=$pf', round(d['value'],1), 'LV/s  sweep', round(d['roofline']['avg_launch_ms'],4), 'ms', round(d['roofline']['achieved']), 'GB/s')" || exit 1
  done
done
JCH_SWEEP_PF=0 timeout -k 10 200 python bench.py --algo plsnipals --p 2000 --q 1 --nlv 10 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg4-shaped plsnipals PF=0', round(d['value'],1), d['device_ms_per_step'])"
JCH_SWEEP_PF=1 timeout -k 10 200 python bench.py --algo plsnipals --p 2000 --q 1 --nlv 10 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg4-shaped plsnipals PF=1', round(d['value'],1), d['device_ms_per_step'])"
