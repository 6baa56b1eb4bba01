#!/bin/bash
# JCH_CUS=<count>: every persistent grid sized for fewer CUs — which kernels besides the sweep run faster with fewer streams?
for cus in 256 224 208 256 232 192; do
  JCH_CUS=$cus python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-path 2>/dev/null > gpurun_out/cus_b.json
  python -c "
import json
d=json.loads(open('gpurun_out/cus_b.json').read().strip().splitlines()[-1]); s=d['device_ms_per_step']
print('cus=$cus headline dev fit', round(s['fit'],4), 'prologue', round(s['prologue'],4), 'sweeps', round(s['sweeps'],4), 'small', round(s['small_state_and_gaps'],3), 'LV/s', round(d['value'],1))
for o in d['other_configs']:
    ds=o.get('device_ms_per_step',{})
    print('      ', str(o['config'])[:52].ljust(52), round(o['value'],1), o['unit'], {k:(round(v,4) if isinstance(v,float) else v) for k,v in ds.items()})"
  JCH_CUS=$cus python tools/bench_accessors.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('       accessors', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,(dict,list,str))})"
done
