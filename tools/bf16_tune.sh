#!/bin/bash
for cfg in "4 0" "2 0" "4 4" "2 6"; do
  set -- $cfg
  JCH_BF16_R=$1 JCH_BF16_BPC=$2 python bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('R=$1 BPC=$2', '%.0f LV/s'%d['value'], 'sweep ms %.4f'%d['roofline']['avg_launch_ms'], 'GB/s %.0f'%d['roofline']['achieved'], d['device_ms_per_step']['small_state_and_gaps'])"
done
