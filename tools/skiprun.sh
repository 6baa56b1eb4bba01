#!/bin/bash
# diagnostic: time k_lv_update_fast with sections disabled (JCH_LV_SKIP bitmask; results are wrong, timing only)
export TMPDIR=/tmp; R=$PWD; cd /tmp
for m in ${MASKS:-0 8 63}; do
  JCH_LV_SKIP=$m timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/skip$m -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/skip$m.log 2>&1
  echo "skip=$m: $(grep k_lv_update_fast $R/gpurun_out/skip$m/*/*kernel_stats.csv | cut -d, -f2-4,6,7)"
done
