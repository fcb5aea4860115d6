#!/bin/bash
# SQ counter breakdown of the halo weight-gradient kernel on one big layer (tools/wgradprobe.py)
mkdir -p gpurun_out/r2_wgpmc
O=$PWD/gpurun_out/r2_wgpmc
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
wc -l $O/sq_counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_EXP_GDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python $R/tools/wgradprobe.py > $O/p$i.log 2>&1
done
cd $R
python - <<'PY'
import csv,collections,glob
for f in sorted(glob.glob('gpurun_out/r2_wgpmc/p*/p_counter_collection.csv')):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'wgrad_kernel' in r['Kernel_Name']:
            d[r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
    for g,c in d.items():
        print(f.split('/')[-2],"grid",g,{k:"%.3g"%(sum(v)/len(v)) for k,v in c.items()}, "n=",len(next(iter(c.values()))))
PY
