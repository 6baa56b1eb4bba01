#!/bin/bash
# short-shard sweep knobs at the 125 k-row share (device ms per fit, sweep us per launch)
for kv in "" "JCH_SWEEP_V2=8" "JCH_SWEEP_V2=4" "JCH_SWEEP_NBUF=3" "JCH_SWEEP_BLOCKS_PER_CU=2" "JCH_SWEEP_V2=0"; do
  env $kv python bench.py --rows 125000 --steps 20 --warmup 3 --no-cpu-baseline --no-host-path --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv'.ljust(28), 'fit', round(d['device_ms_per_step']['fit'],3), 'sweep', round(d['roofline']['avg_launch_ms']*1e3,1), 'us')"
done
