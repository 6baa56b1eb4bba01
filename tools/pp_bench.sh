set -e
python -m pytest tests/test_gpu_parity.py -x -q -k "long_input or predict or xfit" > gpurun_out/pp_test.log 2>&1 || { tail -40 gpurun_out/pp_test.log; exit 1; }
tail -3 gpurun_out/pp_test.log
python tools/bench_accessors.py 2>/dev/null | tail -1
JCH_PREDICT_NT=0 python tools/bench_accessors.py 2>/dev/null | tail -1
JCH_PREDICT_PREFIX=0 python tools/bench_accessors.py 2>/dev/null | tail -1
