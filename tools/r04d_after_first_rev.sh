#!/bin/bash
O=gpurun_out/final4d; mkdir -p $O
F="--no-cpu-baseline --no-host-path --no-other-configs"
python -m pytest tests -m gpu -x -q > $O/gputests_first_rev.log 2>&1; tail -3 $O/gputests_first_rev.log
python bench.py --steps 20 --warmup 5 $F > $O/bench_first_rev.json 2>/dev/null; cut -c1-160 $O/bench_first_rev.json
JCH_SWEEP_FIRST_REV=0 python bench.py --steps 20 --warmup 5 $F > $O/bench_first_fwd.json 2>/dev/null; cut -c1-160 $O/bench_first_fwd.json
python bench.py > $O/bench_default_first_rev.json 2>$O/bench_default_first_rev.err; cut -c1-200 $O/bench_default_first_rev.json
