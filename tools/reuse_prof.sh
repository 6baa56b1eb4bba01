export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ru_stats -- python $R/tools/bench_gridcv.py > $R/gpurun_out/ru_stats.log 2>&1
cd $R
f=$(find gpurun_out/ru_stats -name "*kernel_stats.csv" | head -1)
grep -E "k_xty_rows|k_center_xty_panel|k_sweep_v2|k_score_sums_lv" $f | cut -c1-60,200-330
