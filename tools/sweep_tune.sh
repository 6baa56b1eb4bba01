#!/bin/bash
# diagnostic: HBM read ceiling + sweep variants (rows per iteration, non-temporal loads, blocks per CU)
tools/hbm_read_bw
for cfg in "4 0 0" "4 1 0" "2 0 0" "2 1 0" "8 0 0" "8 1 0" "4 0 2" "4 0 3" "8 0 2" "2 0 4"; do
  set -- $cfg
  JCH_SWEEP_R=$1 JCH_SWEEP_NT=$2 JCH_SWEEP_BLOCKS_PER_CU=$3 timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('R=$1 NT=$2 BPC=$3', 'LV/s %.0f'%d['value'], 'sweep ms %.4f'%d['roofline']['avg_launch_ms'], 'GB/s %.0f'%d['roofline']['achieved'])"
done
