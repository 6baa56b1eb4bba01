F="--dtype bf16 --no-cpu-baseline --no-host-path --no-other-configs"
for nb in 2 3 4 5 2 3 4 5; do
  JCH_BF16_NBUF=$nb python bench.py --steps 20 --warmup 4 $F 2>/dev/null > gpurun_out/bf_b.json
  python -c "
import json
d=json.loads(open('gpurun_out/bf_b.json').read().strip().splitlines()[-1]); print('nbuf=$nb n=1e6 LV/s', round(d['value'],1), 'sweep us', round(1e3*d['roofline']['avg_launch_ms'],2), d['roofline']['frac'])"
done
for nb in 2 4 5; do
  JCH_BF16_NBUF=$nb python bench.py --rows 8000000 --steps 3 --warmup 1 $F 2>/dev/null > gpurun_out/bf_b.json
  python -c "
import json
d=json.loads(open('gpurun_out/bf_b.json').read().strip().splitlines()[-1]); print('nbuf=$nb n=8e6 LV/s', round(d['value'],1), 'sweep us', round(1e3*d['roofline']['avg_launch_ms'],2), d['roofline']['frac'])"
done
