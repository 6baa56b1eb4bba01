#!/bin/bash
# sweep grid size (JCH_SWEEP_NB) against the balance of the last round of row groups
F="--no-cpu-baseline --no-host-path --no-other-configs"
one() {
  nbv=$1; rowsv=$2; shift 2
  JCH_SWEEP_NB=$nbv python bench.py --rows $rowsv --steps 30 --warmup 4 $F "$@" 2>/dev/null > gpurun_out/nb_b.json
  python -c "
import json
d=json.loads(open('gpurun_out/nb_b.json').read().strip().splitlines()[-1]); s=d['device_ms_per_step']
print('nb=$nbv rows=$rowsv $*'.ljust(40), 'dev fit', round(s['fit'],4), 'sweep us', round(1e3*d['roofline']['avg_launch_ms'],2), 'small+gaps', round(s['small_state_and_gaps'],3), 'LV/s', round(d['value'],1))"
}
for rep in 1 2 3; do for nb in 208 176 192 200 216 184 256; do one $nb 1000000; done; done
for rep in 1 2 3; do for nb in 224 208 216 232 240 200 256; do one $nb 125000; done; done
echo p=1000; for rep in 1 2; do for nb in 256 224 208 192; do one $nb 500000 --p 1000; done; done
echo p=250; for rep in 1 2; do for nb in 256 224 208 192; do one $nb 2000000 --p 250; done; done
echo p=120; for rep in 1 2; do for nb in 256 224 208 192; do one $nb 4000000 --p 120; done; done
echo q=1; for rep in 1 2; do for nb in 256 224 208; do one $nb 1000000 --q 1; done; done
