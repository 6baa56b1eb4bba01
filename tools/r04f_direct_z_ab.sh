#!/bin/bash
# same-box A/B of the direct form of Z = P'K in the split small-state path: the tree of b5428fc (incremental Z) built under scratch_z/old
# (git archive b5428fc | tar -x -C scratch_z/old; build() there) against this tree, interleaved, cfg2 default steps and the 125 k-row share
O=gpurun_out/r04f_ab; mkdir -p $O
F="--no-cpu-baseline --no-host-path --no-other-configs"
pick='import json,sys; d=json.loads(sys.stdin.readline()); p=d["device_ms_per_step"]; print(sys.argv[1], "LV/s %.1f ms %.3f prologue %.3f sweeps %.3f small+gaps %.3f sweep_us %.1f" % (d["value"], d["ms_per_step"], p["prologue"], p["sweeps"], p["small_state_and_gaps"], 1e3*d["roofline"]["avg_launch_ms"]))'
for r in 1 2 3; do
  for t in old new; do
    if [ $t = old ]; then B=scratch_z/old/bench.py; else B=bench.py; fi
    timeout -k 10 120 python $B --steps 20 --warmup 3 $F 2>/dev/null | python -c "$pick" "cfg2_$t" | tee -a $O/ab.txt || exit 1
    timeout -k 10 120 python $B --rows 125000 --steps 40 --warmup 5 $F 2>/dev/null | python -c "$pick" "125k_$t" | tee -a $O/ab.txt || exit 1
  done
done
