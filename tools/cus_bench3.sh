#!/bin/bash
# JCH_CUS scan of the NIPALS-shaped fits at cfg2 shape (their own sweep / X'DY passes) and of the siblings on the plskern sweep
F="--no-cpu-baseline --no-host-path --no-other-configs"
for algo in "plsnipals --q 10" "plswold --q 10" "plsnipals --q 1" "plsnipals --q 10 --one-pass" "plssimp" "plsrosa"; do
  for cus in 256 224 208 256 232; do
    JCH_CUS=$cus python bench.py --algo $algo --steps 6 --warmup 2 $F 2>/dev/null > gpurun_out/cus_b.json
    python -c "
import json
d=json.loads(open('gpurun_out/cus_b.json').read().strip().splitlines()[-1]); s=d['device_ms_per_step']
print('$algo cus=$cus'.ljust(44), 'LV/s', round(d['value'],1), 'dev fit', round(s['fit'],3), 'dominant', round(s['sweeps'],3))"
  done
done
