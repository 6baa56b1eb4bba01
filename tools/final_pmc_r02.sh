#!/bin/bash
# PMC part of tools/final_profiles_r02.sh (one counter group per pass; the program itself follows `--`)
O=gpurun_out/final2
tools/pmc_pass.sh final2/pmc "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" -- --steps 2 --warmup 1 --no-cpu-baseline --no-host-path
tools/pmc_pass.sh final2/pmc_wold "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" -- --algo plswold --steps 1 --warmup 1 --no-cpu-baseline --no-host-path
tools/pmc_pass.sh final2/pmc_bf16 "FETCH_SIZE" "WRITE_SIZE" -- --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path
tools/pmc_pass.sh final2/pmc_cfg4 "FETCH_SIZE" "WRITE_SIZE" -- --algo plsnipals --p 2000 --q 1 --nlv 10 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt
python tools/pmc_summary.py $O/pmc_wold k_deflate k_center k_affine > $O/pmc_wold_summary.txt
python tools/pmc_summary.py $O/pmc_bf16 k_sweep k_center > $O/pmc_bf16_summary.txt
python tools/pmc_summary.py $O/pmc_cfg4 k_sweep k_deflate k_center > $O/pmc_cfg4_summary.txt
python bench.py --algo plsnipals --p 2000 --q 1 --nlv 50 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $O/bench_cfg4_b.json 2>/dev/null
cat $O/pmc_summary.txt | head -60
