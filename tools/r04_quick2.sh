#!/bin/bash
# round-4: failing tests again + P2P multi-process tests + stamps + accessor timings
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -k "sibling or kspace_matches or split_small or p2p or rehearsal or row_sharded or envelope or long_input or accessor" > $O/r04_gpu_tests_b.log 2>&1; tail -6 $O/r04_gpu_tests_b.log | cut -c1-300
JCH_LV_DEBUG=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs > /dev/null 2> $O/r04_lvdebug.err; grep -m5 "jch" $O/r04_lvdebug.err
for rt in 1 2 4; do echo "== JCH_GEMM_RT=$rt"; JCH_GEMM_RT=$rt python tools/bench_accessors.py 2>&1 | tail -4; done
