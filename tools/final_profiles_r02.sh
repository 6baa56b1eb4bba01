#!/bin/bash
# regenerates the round-2 evidence under gpurun_out/final2 (copied to profiles/r02_* afterwards by tools/collect_profiles_r02.py)
set -x
O=gpurun_out/final2; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
cat $O/bench.json
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $R/$O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_cfg4 -- python $R/bench.py --algo plsnipals --p 2000 --q 1 --nlv 50 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > $R/$O/stats_cfg4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_bf16 -- python $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $R/$O/stats_bf16.log 2>&1
cd $R
# PMC: one counter group per pass (FETCH_SIZE / WRITE_SIZE: TCC slots; MFMA / busy cycles: SQ), headline config
tools/final_pmc_r02.sh
# MFMA users off the headline path: K6 tile deflation (plswold q = 10) and K7 (transform / predict)
python bench.py --algo plsnipals --p 2000 --q 1 --nlv 50 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $O/bench_cfg4.json 2>/dev/null
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-host-path > $O/bench_bf16.json 2>/dev/null
python bench.py --dtype bf16 --rows 8000000 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $O/bench_bf16_n8e6_one_gpu.json 2>/dev/null
python bench.py --rows 125000 --steps 20 --warmup 3 --no-cpu-baseline --no-host-path > $O/bench_rank_share_125k.json 2>/dev/null
for r in 500000 250000; do python bench.py --rows $r --steps 10 --warmup 3 --no-cpu-baseline --no-host-path > $O/bench_rank_share_$r.json 2>/dev/null; done
python bench.py --scal --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/bench_scal.json 2>/dev/null
for a in plssimp plsrosa plswold plskern2; do python bench.py --algo $a --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/bench_$a.json 2>/dev/null; done
python tools/bench_lwplsr.py 2>/dev/null | tail -1 > $O/lwplsr_cfg5.json
JCH_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29517 tools/bench_lwplsr.py 2>/dev/null | tail -1 > $O/lwplsr_cfg5_3replicas_one_gpu.json
python tools/bench_gridcv.py 2>/dev/null | tail -1 > $O/gridcv.json
python tools/bench_accessors.py 2>/dev/null | tail -1 > $O/accessors.json
K2_VARIANTS="JCH_K2_PANEL=0;JCH_K2_PANEL=1" python tools/k2_modes.py 2>/dev/null > $O/k2_tile_vs_panel.log
ls -la $O
