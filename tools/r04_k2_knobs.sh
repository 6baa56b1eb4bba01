#!/bin/bash
# prologue knobs once more on the round-4 tree (bench prologue ms, 5 steps)
for kv in "" "JCH_K2_TH=128" "JCH_K2_TW=128" "JCH_K2_TH=128 JCH_K2_TW=128" "JCH_K2_BPC=2"; do
  env $kv python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv'.ljust(34), 'prologue', round(d['device_ms_per_step']['prologue'],3), 'fit', round(d['device_ms_per_step']['fit'],3))"
done
