#!/bin/bash
# second half of round 4: evidence for the prediction-range kernel (k_predict_prefix), the per-level score statistics
# (k_score_sums_lv), the refreshed headline; output under gpurun_out/final4b (copied to profiles/r04b_* by hand)
set -x
O=gpurun_out/final4b; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cut -c1-300 $O/bench.json
python tools/bench_accessors.py 2>/dev/null | tail -1 > $O/accessors.json; cat $O/accessors.json
python tools/bench_gridcv.py 2>/dev/null | tail -1 > $O/gridcv.json; cat $O/gridcv.json
python tools/cv_fold_split.py 2>/dev/null | tail -1 > $O/gridcv_fold_split.json; cat $O/gridcv_fold_split.json
python bench.py --rows 125000 --steps 40 --warmup 5 --no-cpu-baseline --no-host-path --no-other-configs > $O/bench_rank_share_125k.json 2>/dev/null
JCH_LV_MERGED=1 python bench.py --rows 125000 --steps 40 --warmup 5 --no-cpu-baseline --no-host-path --no-other-configs > $O/bench_rank_share_125k_merged_optin.json 2>/dev/null
JCH_LV_MERGED=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-other-configs > $O/bench_merged_optin.json 2>/dev/null
JCH_SWEEP_ALT=1 python bench.py --rows 125000 --steps 40 --warmup 5 --no-cpu-baseline --no-host-path --no-other-configs > $O/bench_rank_share_125k_sweep_alt_optin.json 2>/dev/null
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-other-configs > $R/$O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_accessors -- python $R/tools/bench_accessors.py > $R/$O/stats_accessors.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_gridcv -- python $R/tools/bench_gridcv.py > $R/$O/stats_gridcv.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_acc_fetch -- python $R/tools/bench_accessors.py > $R/$O/pmc_acc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_acc_write -- python $R/tools/bench_accessors.py > $R/$O/pmc_acc_write.log 2>&1
cd $R
find $O -name "*kernel_stats.csv" | head
ls $O
