#!/bin/bash
# One rocprofv3 PMC pass per counter group over a short bench run; output CSVs under gpurun_out/<tag>/<group>/.
# usage: tools/pmc_pass.sh <tag> "<counters group 1>" "<counters group 2>" ... -- <bench.py args>
export TMPDIR=/tmp; R=$PWD; tag=$1; shift
groups=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do groups+=("$1"); shift; done
shift
mkdir -p $R/gpurun_out/$tag; cd /tmp
i=0
for g in "${groups[@]}"; do
  timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $R/gpurun_out/$tag/g$i -- python $R/bench.py "$@" > $R/gpurun_out/$tag/g$i.log 2>&1
  echo "group $i [$g] exit $?"
  i=$((i+1))
done
