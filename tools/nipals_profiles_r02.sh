#!/bin/bash
# plsnipals / plswold evidence after the postponed write-back (round 2): benches (default and eager), kernel stats and the
# HBM-traffic PMC passes of the lazy kernels -> gpurun_out/final2, copied to profiles/r02_* by tools/collect_profiles_r02.py
O=gpurun_out/final2; mkdir -p $O
export TMPDIR=/tmp; R=$PWD
B="--steps 3 --warmup 1 --no-cpu-baseline --no-host-path"
python bench.py --algo plsnipals --p 2000 --q 1 --nlv 50 $B > $O/bench_cfg4.json 2>/dev/null &&
JCH_NIPALS_DEFER=1 python bench.py --algo plsnipals --p 2000 --q 1 --nlv 50 $B > $O/bench_cfg4_eager.json 2>/dev/null &&
python bench.py --algo plsnipals $B > $O/bench_plsnipals_q10.json 2>/dev/null &&
JCH_NIPALS_DEFER=1 python bench.py --algo plsnipals $B > $O/bench_plsnipals_q10_eager.json 2>/dev/null &&
python bench.py --algo plswold $B > $O/bench_plswold.json 2>/dev/null &&
JCH_NIPALS_DEFER=1 python bench.py --algo plswold $B > $O/bench_plswold_eager.json 2>/dev/null || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_cfg4 -- python $R/bench.py --algo plsnipals --p 2000 --q 1 --nlv 50 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > $R/$O/stats_cfg4.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_nipals_q10 -- python $R/bench.py --algo plsnipals --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > $R/$O/stats_nipals_q10.log 2>&1 || exit 1
cd $R
tools/pmc_pass.sh final2/pmc_cfg4 "FETCH_SIZE" "WRITE_SIZE" -- --algo plsnipals --p 2000 --q 1 --nlv 13 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path &&
tools/pmc_pass.sh final2/pmc_nipals_q10 "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" -- --algo plsnipals --nlv 13 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path || exit 1
python tools/pmc_summary.py $O/pmc_cfg4 k_sweep k_kpass k_deflate k_center > $O/pmc_cfg4_summary.txt
python tools/pmc_summary.py $O/pmc_nipals_q10 k_sweep k_kpass k_deflate k_center > $O/pmc_nipals_q10_summary.txt
cat $O/pmc_cfg4_summary.txt $O/pmc_nipals_q10_summary.txt
for f in bench_cfg4 bench_cfg4_eager bench_plsnipals_q10 bench_plsnipals_q10_eager bench_plswold bench_plswold_eager; do python - $O/$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d["value"],1), "LV/s", d["device_ms_per_step"])
PY
done
