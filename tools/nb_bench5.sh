#!/bin/bash
# kernel shape x grid size at n = 1e6: <4,8,2> (default), <4,4,3> (JCH_SWEEP_NBUF=3), <4,4,2> (JCH_SWEEP_V2=4)
F="--no-cpu-baseline --no-host-path --no-other-configs"
one() {  # $1 = env assignments, $2 = nb, $3 = rows
  env $1 JCH_SWEEP_NB=$2 python bench.py --rows $3 --steps 30 --warmup 4 $F 2>/dev/null > gpurun_out/nb_b.json
  python -c "
import json
d=json.loads(open('gpurun_out/nb_b.json').read().strip().splitlines()[-1]); s=d['device_ms_per_step']
print('$1 nb=$2 rows=$3'.ljust(44), 'dev fit', round(s['fit'],4), 'sweep us', round(1e3*d['roofline']['avg_launch_ms'],2), 'LV/s', round(d['value'],1))"
}
for rep in 1 2; do for nb in 208 192 224 176; do for e in "JCH_X=0" "JCH_SWEEP_NBUF=3" "JCH_SWEEP_V2=4"; do one $e $nb 1000000; done; done; done
for rep in 1 2; do for nb in 224 208 240; do for e in "JCH_X=0" "JCH_SWEEP_NBUF=2" "JCH_SWEEP_V2=8"; do one $e $nb 125000; done; done; done
