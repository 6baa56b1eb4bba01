#!/bin/bash
# round-4 evidence of the screened kNN at cfg5: kernel-trace stats, HBM traffic (FETCH_SIZE / WRITE_SIZE) and matrix-pipe / vector busy
# counters of its kernels, one counter group per pass -> gpurun_out/screen_ev/
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/screen_ev; mkdir -p $O; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/bench_lwplsr.py > $O/stats.log 2>&1 || exit 1
i=0
for g in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $O/g$i -- python3 $R/tools/bench_lwplsr.py > $O/g$i.log 2>&1 || { echo "pass $i ($g) failed"; tail -3 $O/g$i.log; }
  i=$((i+1))
done
cd $R
python3 tools/pmc_summary.py $O k_knn k_ks_ k_locw > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt | head -60
python3 tools/bench_lwplsr.py 2>&1 | tail -1 > $O/bench_lwplsr.json; cat $O/bench_lwplsr.json
