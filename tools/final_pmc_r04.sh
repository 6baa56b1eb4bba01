#!/bin/bash
# round-4 counter passes (separate rocprofv3 --pmc runs, one counter group each): headline, bf16 share, cfg4 (k_sweep_lazy /
# k_kpass_lazy), plsnipals q = 10 (k_kpass_mfma_lazy), cfg5 SQ counters
set -x
O=gpurun_out/final4
tools/pmc_pass.sh final4/pmc "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" -- --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs
tools/pmc_pass.sh final4/pmc_bf16 "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" -- --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path
tools/pmc_pass.sh final4/pmc_cfg4 "FETCH_SIZE" "WRITE_SIZE" -- --algo plsnipals --p 2000 --q 1 --nlv 13 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path
tools/pmc_pass.sh final4/pmc_nipals_q10 "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" -- --algo plsnipals --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt
python tools/pmc_summary.py $O/pmc_bf16 k_sweep k_center k_xty > $O/pmc_bf16_summary.txt
python tools/pmc_summary.py $O/pmc_cfg4 k_sweep_lazy k_kpass > $O/pmc_cfg4_summary.txt
python tools/pmc_summary.py $O/pmc_nipals_q10 k_kpass k_sweep > $O/pmc_nipals_q10_summary.txt
tools/lwplsr_pmc.sh > /dev/null
ls $O
