#!/bin/bash
O=gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -q -k "lwplsr or envelope or accessor or long_input or transform or predict" > $O/r04_gpu_tests_e.log 2>&1; tail -4 $O/r04_gpu_tests_e.log | cut -c1-300
python tools/bench_accessors.py 2>/dev/null | tail -1
JCH_GEMM_WIDEOUT=1 python tools/bench_accessors.py 2>/dev/null | tail -1
python tools/bench_lwplsr.py 2>/dev/null | tail -1 | cut -c1-400
