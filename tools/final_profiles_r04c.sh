#!/bin/bash
# last session of round 4: evidence for the smaller sweep grid (208 / 224 blocks); output under gpurun_out/final4c (copied to profiles/r04c_* by hand)
O=gpurun_out/final4c; mkdir -p $O
F="--no-cpu-baseline --no-host-path --no-other-configs"
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -2 $O/gputests.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
JCH_SWEEP_NB=256 python bench.py --steps 20 --warmup 5 $F > $O/bench_former_grid_256.json 2>/dev/null
for rows in 500000 250000 125000; do python bench.py --rows $rows --steps 40 --warmup 5 $F > $O/bench_rank_share_$rows.json 2>/dev/null; done
JCH_SWEEP_NB=256 python bench.py --rows 125000 --steps 40 --warmup 5 $F > $O/bench_rank_share_125000_former_grid_256.json 2>/dev/null
python bench.py --p 1000 --rows 500000 --steps 10 --warmup 3 $F > $O/bench_p1000.json 2>/dev/null
JCH_SWEEP_NB=256 python bench.py --p 1000 --rows 500000 --steps 10 --warmup 3 $F > $O/bench_p1000_former_grid_256.json 2>/dev/null
python tools/bench_gridcv.py 2>/dev/null | tail -1 > $O/gridcv.json
bash tools/kstats.sh final4c_stats --steps 5 --warmup 2 $F | tail -12
bash tools/pmc_pass.sh final4c_pmc "FETCH_SIZE" "WRITE_SIZE" -- --steps 3 --warmup 1 $F
python tools/pmc_summary.py gpurun_out/final4c_pmc k_sweep_v2 > $O/pmc_sweep.txt; cat $O/pmc_sweep.txt
