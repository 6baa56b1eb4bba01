#!/bin/bash
# round 4, last re-entry: sampled sweep events (jch_ctx_set_profiling(ctx, N > 1)); output under gpurun_out/final4d (copied to profiles/r04d_* by hand)
O=gpurun_out/final4d; mkdir -p $O
F="--no-cpu-baseline --no-host-path --no-other-configs"
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sampled_profiling" > $O/sampled_test.log 2>&1; tail -2 $O/sampled_test.log
python tools/prof_overhead.py > $O/prof_overhead.json 2> $O/prof_overhead.err; cat $O/prof_overhead.json
python bench.py --steps 20 --warmup 5 $F > $O/bench_stride6.json 2>/dev/null; cut -c1-160 $O/bench_stride6.json
JCH_BENCH_PROF_STRIDE=1 python bench.py --steps 20 --warmup 5 $F > $O/bench_stride1.json 2>/dev/null; cut -c1-160 $O/bench_stride1.json
python bench.py --rows 125000 --steps 40 --warmup 5 $F > $O/bench_rank_share_125000.json 2>/dev/null; cut -c1-160 $O/bench_rank_share_125000.json
JCH_BENCH_PROF_STRIDE=1 python bench.py --rows 125000 --steps 40 --warmup 5 $F > $O/bench_rank_share_125000_stride1.json 2>/dev/null; cut -c1-160 $O/bench_rank_share_125000_stride1.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
bash tools/kstats.sh final4d_stats --steps 5 --warmup 2 $F | tail -12
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -2 $O/gputests.log
