# JCH_SWEEP_ALT: default-policy loads + alternating walk direction of the sweep, against the streaming default
F="--no-cpu-baseline --no-host-path --no-other-configs"
for rows in 125000 250000 500000 1000000; do for m in 0 1 2 0 1; do
  JCH_SWEEP_ALT=$m python bench.py --rows $rows --steps 30 --warmup 4 $F 2>/dev/null > gpurun_out/alt_b.json
  python -c "
import json
d=json.loads(open('gpurun_out/alt_b.json').read().strip().splitlines()[-1]); print('alt=$m rows=$rows ms/fit', round(d['ms_per_step'],4), 'sweep avg us', round(1e3*d['roofline']['avg_launch_ms'],2), 'small', round(d['device_ms_per_step']['small_state_and_gaps'],3))"
done; done
