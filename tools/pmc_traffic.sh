#!/bin/bash
# HBM traffic of the sweep kernel from PMC counters, one counter set per pass (MI355X_MICROARCH.md: TCC has 4 slots,
# FETCH_SIZE costs 3, WRITE_SIZE 2 -> separate passes).  Output: gpurun_out/pmc_{fetch,write}/
export TMPDIR=/tmp; R=$PWD; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$c.log 2>&1
  echo "$c exit $?"; ls $R/gpurun_out/pmc_$c/*/ | head
done
