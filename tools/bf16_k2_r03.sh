#!/bin/bash
# bf16 prologue variants at the cfg3 share (n = 1e6): tile kernel vs row panel (NH = 2 / 4, blocks per CU)
cd "$(dirname "$0")/.."
for v in "JCH_BF16_K2_PANEL=0" "JCH_BF16_K2_PANEL=1 JCH_BF16_K2_NH=2" "JCH_BF16_K2_PANEL=1 JCH_BF16_K2_NH=2 JCH_BF16_K2_PBPC=1" "JCH_BF16_K2_PANEL=1 JCH_BF16_K2_NH=2 JCH_BF16_K2_PBPC=3" "JCH_BF16_K2_PANEL=1 JCH_BF16_K2_NH=4" "JCH_BF16_K2_PANEL=1 JCH_BF16_K2_NH=2 JCH_K2_SKIP=1" "JCH_BF16_K2_PANEL=1 JCH_BF16_K2_NH=2 JCH_K2_SKIP=2" "JCH_BF16_K2_PANEL=1 JCH_BF16_K2_NH=2 JCH_K2_SKIP=3"; do
  echo "== $v"
  env $v python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-host-path --no-other-configs | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['device_ms_per_step'])"
done
