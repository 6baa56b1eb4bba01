#!/bin/bash
# regenerates the round-4 evidence under gpurun_out/final4 (copied to profiles/r04_* afterwards by tools/collect_profiles_r04.py)
# part 1 (this script): bench lines + rocprofv3 kernel-trace stats; part 2: tools/final_pmc_r04.sh (counter passes)
set -x
O=gpurun_out/final4; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cut -c1-300 $O/bench.json
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-other-configs > $R/$O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_cfg4 -- python $R/bench.py --algo plsnipals --p 2000 --q 1 --nlv 50 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > $R/$O/stats_cfg4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_bf16 -- python $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $R/$O/stats_bf16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_lwplsr -- python $R/tools/bench_lwplsr.py > $R/$O/stats_lwplsr.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_nipals_q10 -- python $R/bench.py --algo plsnipals --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs > $R/$O/stats_nipals_q10.log 2>&1
cd $R
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-host-path > $O/bench_bf16.json 2>/dev/null
python bench.py --dtype bf16 --rows 8000000 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $O/bench_bf16_n8e6_one_gpu.json 2>/dev/null
python bench.py --algo plsnipals --p 2000 --q 1 --nlv 50 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $O/bench_cfg4.json 2>/dev/null
python bench.py --rows 125000 --steps 20 --warmup 3 --no-cpu-baseline --no-host-path --no-other-configs > $O/bench_rank_share_125k.json 2>/dev/null
for r in 500000 250000; do python bench.py --rows $r --steps 10 --warmup 3 --no-cpu-baseline --no-host-path --no-other-configs > $O/bench_rank_share_$r.json 2>/dev/null; done
JCH_LV_SPLIT=0 python bench.py --rows 125000 --steps 20 --warmup 3 --no-cpu-baseline --no-host-path --no-other-configs > $O/bench_rank_share_125k_one_kernel_small_state.json 2>/dev/null
JCH_LV_SPLIT=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-path --no-other-configs > $O/bench_one_kernel_small_state.json 2>/dev/null
python bench.py --scal --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/bench_scal.json 2>/dev/null
for a in plssimp plsrosa plswold plskern2; do python bench.py --algo $a --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/bench_$a.json 2>/dev/null; done
python bench.py --algo plsnipals --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/bench_plsnipals_q10.json 2>/dev/null
JCH_KPASS_NW=4 python bench.py --algo plsnipals --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/bench_plsnipals_q10_four_waves.json 2>/dev/null
python tools/bench_lwplsr.py 2>/dev/null | tail -1 > $O/lwplsr_cfg5.json
python tools/bench_gridcv.py 2>/dev/null | tail -1 > $O/gridcv.json
python tools/bench_accessors.py 2>/dev/null | tail -1 > $O/accessors.json
JCH_LV_DEBUG=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs > /dev/null 2> $O/lv_debug_split.err
JCH_LV_DEBUG=1 JCH_LV_SPLIT=0 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs > /dev/null 2> $O/lv_debug_one_kernel.err
ls -la $O
# opt-in one-pass NIPALS (never the default): its bench lines
python bench.py --algo plsnipals --one-pass --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-other-configs > gpurun_out/final4/bench_plsnipals_q10_one_pass_optin.json 2>/dev/null
python bench.py --algo plswold --one-pass --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-other-configs > gpurun_out/final4/bench_plswold_one_pass_optin.json 2>/dev/null
python bench.py --algo plsnipals --one-pass --p 2000 --q 1 --nlv 50 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs > gpurun_out/final4/bench_cfg4_one_pass_optin.json 2>/dev/null
