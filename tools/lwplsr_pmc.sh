#!/bin/bash
# HBM traffic counters of the cfg5 kernels (one counter per pass): tools/lwplsr_pmc.sh -> gpurun_out/final4/pmc_lwplsr_summary.txt
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/final4/pmc_lwplsr; mkdir -p $O; cd /tmp
i=0
for g in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $O/g$i -- python $R/tools/bench_lwplsr.py > $O/g$i.log 2>&1 || exit 1
  i=$((i+1))
done
cd $R
python tools/pmc_summary.py $O k_locw k_knn k_to_rowmajor > gpurun_out/final4/pmc_lwplsr_summary.txt
cat gpurun_out/final4/pmc_lwplsr_summary.txt
