#!/bin/bash
# JCH_CUS scan of the accessors (transform / predict / predict range / summary) and of gridcvlv (k_xty_rows, k_score_sums_lv)
for cus in 256 224 208 256 232 192; do
  JCH_CUS=$cus python tools/bench_accessors.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cus=$cus accessors', {k:round(v['ms'],4) for k,v in d.items() if isinstance(v,dict) and 'ms' in v})"
  JCH_CUS=$cus python tools/bench_gridcv.py 2>/dev/null | tail -1 | cut -c1-400
done
