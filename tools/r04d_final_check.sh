#!/bin/bash
# final tree of round 4: smoke, the bench with nlv = 3 / 6 (stride made coprime), the 3-rank rehearsal test, the default bench
O=gpurun_out/final4d; mkdir -p $O
F="--no-cpu-baseline --no-host-path --no-other-configs"
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --nlv 3 --steps 6 --warmup 2 $F > $O/bench_nlv3.json 2>$O/bench_nlv3.err; cut -c1-200 $O/bench_nlv3.json
python bench.py --nlv 6 --steps 5 --warmup 2 $F > $O/bench_nlv6.json 2>$O/bench_nlv6.err; cut -c1-200 $O/bench_nlv6.json
python bench.py --steps 20 --warmup 5 $F > $O/bench_stride6_b.json 2>/dev/null; cut -c1-160 $O/bench_stride6_b.json
JCH_BENCH_PROF_STRIDE=1 python bench.py --steps 20 --warmup 5 $F > $O/bench_stride1_b.json 2>/dev/null; cut -c1-160 $O/bench_stride1_b.json
python -m pytest tests/test_bench_rehearsal.py -m gpu -x -q > $O/rehearsal.log 2>&1; tail -2 $O/rehearsal.log
python bench.py > $O/bench_default_final.json 2> $O/bench_default_final.err; cut -c1-220 $O/bench_default_final.json
