#!/bin/bash
# Does the K2 time of a process correlate with its TLB misses?  4 separate processes, UTCL1 counters + kernel trace each.
export JCH_K2_TH=64
for i in 1 2 3 4; do
  tools/pmc_pass.sh r02/tlb$i "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" -- --steps 3 --warmup 1 --no-cpu-baseline > /dev/null
  python tools/pmc_summary.py gpurun_out/r02/tlb$i k_center_xty
done
