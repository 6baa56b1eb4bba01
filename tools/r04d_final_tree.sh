#!/bin/bash
# final tree of round 4 (sampled events, reversed first sweep): GPU suite, default bench, rocprofv3 kernel stats of the bench, smoke, rank share
O=gpurun_out/final4g; mkdir -p $O
F="--no-cpu-baseline --no-host-path --no-other-configs"
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -2 $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
python bench.py --rows 125000 --steps 40 --warmup 5 $F > $O/bench_rank_share_125000.json 2>/dev/null; cut -c1-160 $O/bench_rank_share_125000.json
bash tools/kstats.sh final4g_stats --steps 5 --warmup 2 $F | tail -8
python tools/r04d_invariant_probe.py 2>&1 | grep FIRST_REV | tee $O/invariant_probe.txt
