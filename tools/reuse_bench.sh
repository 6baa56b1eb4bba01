set -e
python -m pytest tests/test_gpu_parity.py -x -q -k "previous_fits_row_major or many_responses or grid or pars or scores" > gpurun_out/ru_test.log 2>&1 || { tail -40 gpurun_out/ru_test.log; exit 1; }
tail -2 gpurun_out/ru_test.log
python tools/bench_gridcv.py 2>/dev/null | tail -1
python tools/bench_gridcv.py 2>/dev/null | tail -1
JCH_NO_REUSE_XCOPY=1 python tools/bench_gridcv.py 2>/dev/null | tail -1
