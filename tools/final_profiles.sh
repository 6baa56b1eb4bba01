#!/bin/bash
# regenerates the round's committed evidence under gpurun_out/final (copied to profiles/ afterwards)
set -x
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
cat gpurun_out/final/bench.json
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/final/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/final/pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/final/pmc_$c.log 2>&1
done
cd $R
python bench.py --algo plsnipals --p 2000 --q 1 --nlv 50 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_cfg4.json 2>/dev/null
python bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_bf16.json 2>/dev/null
python tools/host_rate.py > gpurun_out/final/host_rate.json 2>/dev/null
python tools/bench_lwplsr.py 2>/dev/null | tail -1 > gpurun_out/final/lwplsr_cfg5.json
python tools/bench_gridcv.py 2>/dev/null | tail -1 > gpurun_out/final/gridcv.json
python tools/bench_accessors.py 2>/dev/null | tail -1 > gpurun_out/final/accessors.json
python bench.py --algo plskern2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_plskern2.json 2>/dev/null
ls -la gpurun_out/final
for a in plssimp plsrosa plswold; do python bench.py --algo $a --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_$a.json 2>/dev/null; done
python tools/p2p_cost.py 2>/dev/null | tail -1 > gpurun_out/final/p2p_cost.json
python bench.py --rows 125000 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/final/bench_rank_share_125k.json 2>/dev/null
ls -la gpurun_out/final
python bench.py --scal --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_scal.json 2>/dev/null
python tools/raw_mode_error.py > gpurun_out/final/raw_mode_error.log 2>/dev/null
python tools/small_fit_latency.py 2>/dev/null | tail -1 > gpurun_out/final/small_fit_latency.json
