set -e
python -m pytest tests/test_gpu_parity.py -x -q -k "score or grid or pars or sweep_with_cached or vip or plslda" > gpurun_out/cv_test.log 2>&1 || { tail -40 gpurun_out/cv_test.log; exit 1; }
tail -3 gpurun_out/cv_test.log
python tools/bench_gridcv.py 2>/dev/null | tail -1
python tools/bench_gridcv.py 2>/dev/null | tail -1
