export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/nip_sq; mkdir -p $O; cd /tmp
i=0
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVES"; do
  timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $O/g$i -- python $R/bench.py --algo plsnipals --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs > $O/g$i.log 2>&1 || exit 1
  i=$((i+1))
done
cd $R
python - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/nip_sq/g*/')):
    for f in glob.glob(d+'**/*counter_collection.csv',recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0]
            if 'kpass' in k or 'sweep_lazy' in k:
                acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
        for k in acc:
            for c,v in acc[k].items(): print(f"{k[:44]:44s} {c:28s} per launch {v/n[(k,c)]:16.0f}  (n={n[(k,c)]})")
PY
