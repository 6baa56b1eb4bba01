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
for rep in 1 2 3; do for nb in 256 224 208 256 232 216 200; do one $nb 1000000; done; done
for rep in 1 2 3; do for nb in 256 224 208 256 232 216 200; do one $nb 125000; done; done
for rep in 1 2; do for nb in 256 224 208 232 216; do one $nb 250000; done; done
for rep in 1 2; do for nb in 256 224 208 232 216; do one $nb 500000; done; done
echo bf16 bpc; JCH_SWEEP_NB=100000 python bench.py --rows 1000000 --dtype bf16 --steps 5 --warmup 2 $F 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 default', d['roofline']['launches'], d['roofline']['avg_launch_ms'])"
for rep in 1 2 3; do for nb in 100000 768 512 448 256 224; do one $nb 1000000 --dtype bf16; done; done
