#!/bin/bash
# regenerates the round-3 evidence under gpurun_out/final3 (copied to profiles/r03_* afterwards by tools/collect_profiles_r03.py)
set -x
O=gpurun_out/final3; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cut -c1-400 $O/bench.json
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-other-configs > $R/$O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_cfg4 -- python $R/bench.py --algo plsnipals --p 2000 --q 1 --nlv 50 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > $R/$O/stats_cfg4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_bf16 -- python $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $R/$O/stats_bf16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_lwplsr -- python $R/tools/bench_lwplsr.py > $R/$O/stats_lwplsr.log 2>&1
cd $R
# PMC: one counter group per pass (FETCH_SIZE / WRITE_SIZE: TCC slots; MFMA / busy cycles: SQ), headline config and bf16
tools/pmc_pass.sh final3/pmc "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" -- --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs
tools/pmc_pass.sh final3/pmc_bf16 "FETCH_SIZE" "WRITE_SIZE" -- --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt
python tools/pmc_summary.py $O/pmc_bf16 k_sweep k_center > $O/pmc_bf16_summary.txt
tools/lwplsr_pmc.sh > /dev/null
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-host-path > $O/bench_bf16.json 2>/dev/null
python bench.py --dtype bf16 --rows 8000000 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $O/bench_bf16_n8e6_one_gpu.json 2>/dev/null
python bench.py --algo plsnipals --p 2000 --q 1 --nlv 50 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $O/bench_cfg4.json 2>/dev/null
python bench.py --rows 125000 --steps 20 --warmup 3 --no-cpu-baseline --no-host-path --no-other-configs > $O/bench_rank_share_125k.json 2>/dev/null
for r in 500000 250000; do python bench.py --rows $r --steps 10 --warmup 3 --no-cpu-baseline --no-host-path --no-other-configs > $O/bench_rank_share_$r.json 2>/dev/null; done
python bench.py --scal --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/bench_scal.json 2>/dev/null
for a in plssimp plsrosa plswold plskern2; do python bench.py --algo $a --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/bench_$a.json 2>/dev/null; done
python bench.py --algo plsnipals --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/bench_plsnipals_q10.json 2>/dev/null
python tools/bench_lwplsr.py 2>/dev/null | tail -1 > $O/lwplsr_cfg5.json
JCH_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29517 tools/bench_lwplsr.py 2>/dev/null | tail -1 > $O/lwplsr_cfg5_3replicas_one_gpu.json
python tools/bench_gridcv.py 2>/dev/null | tail -1 > $O/gridcv.json
python tools/bench_accessors.py 2>/dev/null | tail -1 > $O/accessors.json
ls -la $O
