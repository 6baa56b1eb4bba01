F="--no-cpu-baseline --no-host-path --no-other-configs"
JCH_LV_DEBUG=1 python bench.py --steps 1 --warmup 1 $F 2>&1 >/dev/null | grep "call 12\|call 24" | head -2
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 3 $F 2>/dev/null > gpurun_out/sp_b.json
  python -c "
import json
d=json.loads(open('gpurun_out/sp_b.json').read().strip().splitlines()[-1]); print('n=1e6', round(d['value'],1), d['device_ms_per_step']['small_state_and_gaps'])"
  python bench.py --rows 125000 --steps 40 --warmup 5 $F 2>/dev/null > gpurun_out/sp_b.json
  python -c "
import json
d=json.loads(open('gpurun_out/sp_b.json').read().strip().splitlines()[-1]); print('n=125k', round(d['ms_per_step'],4), d['device_ms_per_step']['fit'], d['device_ms_per_step']['small_state_and_gaps'])"
done
