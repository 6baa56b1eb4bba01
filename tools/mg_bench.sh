# merged small-state kernel vs two launches: parity tests, then the headline + the 125 k-row share with JCH_LV_MERGED=1 / 0
set -e
python -m pytest tests/test_gpu_parity.py -x -q -k "merged_small_state or split_small_state" > gpurun_out/mg_test.log 2>&1 || { tail -30 gpurun_out/mg_test.log; exit 1; }
tail -3 gpurun_out/mg_test.log
F="--no-cpu-baseline --no-host-path --no-other-configs"
for m in 1 0 1 0; do
  JCH_LV_MERGED=$m python bench.py --steps 20 --warmup 3 $F 2>/dev/null > gpurun_out/mg_b.json
  python -c "
import json
d=json.loads(open('gpurun_out/mg_b.json').read().strip().splitlines()[-1]); print('merged=$m n=1e6', round(d['value'],1), d.get('device_ms_per_step'))"
  JCH_LV_MERGED=$m python bench.py --rows 125000 --steps 40 --warmup 5 $F 2>/dev/null > gpurun_out/mg_b.json
  python -c "
import json
d=json.loads(open('gpurun_out/mg_b.json').read().strip().splitlines()[-1]); print('merged=$m n=125k', round(d['ms_per_step'],4), d.get('device_ms_per_step'))"
done
