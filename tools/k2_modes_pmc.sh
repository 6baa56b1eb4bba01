#!/bin/bash
# per-dispatch counters of the K2 kernel across ctx re-creations (tools/k2_modes.py) — which counter tracks the slow/fast mode?
export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out/r02/modes; cd /tmp
export JCH_K2_TH=64
i=0
for g in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum"; do
  timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $R/gpurun_out/r02/modes/g$i -- python $R/tools/k2_modes.py > $R/gpurun_out/r02/modes/g$i.log 2>&1
  echo "group $i exit $?"; grep "^group" $R/gpurun_out/r02/modes/g$i.log | cut -c1-60
  python - $R/gpurun_out/r02/modes/g$i <<'PY'
import csv,glob,sys,collections
g=sys.argv[1]
cc=glob.glob(g+"/*/*_counter_collection.csv")[0]; kt=glob.glob(g+"/*/*_kernel_trace.csv")[0]
dur={}
for r in csv.DictReader(open(kt)):
    if "k_center_xty" in r["Kernel_Name"]: dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
vals=collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if "k_center_xty" in r["Kernel_Name"]: vals[r["Dispatch_Id"]][r["Counter_Name"]]=float(r["Counter_Value"])
ids=sorted(vals,key=int)
for k in ids[::6]:
    print(k, "%.0f us"%dur.get(k,-1), {c:"%.3g"%v for c,v in vals[k].items()})
PY
  i=$((i+1))
done
