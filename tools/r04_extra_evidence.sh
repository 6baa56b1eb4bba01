#!/bin/bash
# round-4 extra evidence: fuzz sweeps with fresh seeds (every fit entry point incl. the new split / generic paths), SQ counters of the
# two small-state kernels and of the accessor kernels, the one-pass fuzz
O=gpurun_out/final4; mkdir -p $O
for seed in 4242 777 31337; do
  JCH_FUZZ_SEED=$seed JCH_FUZZ_COUNT=64 timeout -k 10 500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $O/fuzz_$seed.log 2>&1; echo "fuzz seed $seed: $(tail -1 $O/fuzz_$seed.log)"
done
export TMPDIR=/tmp; R=$PWD
tools/pmc_pass.sh final4/pmc_smallstate "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" -- --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs
python tools/pmc_summary.py $O/pmc_smallstate k_lv_ > $O/pmc_smallstate_summary.txt; cat $O/pmc_smallstate_summary.txt | head -40
cd /tmp
for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  d=$R/$O/pmc_accessors/g$(echo $g | cut -c1-5); mkdir -p $d
  timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $d -- python $R/tools/bench_accessors.py > $d.log 2>&1; echo "accessors [$g] exit $?"
done
cd $R; python tools/pmc_summary.py $O/pmc_accessors k_affine > $O/pmc_accessors_summary.txt; cat $O/pmc_accessors_summary.txt | head -40
