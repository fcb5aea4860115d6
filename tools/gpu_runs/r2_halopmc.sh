#!/bin/bash
# SQ counter breakdown of the halo conv kernel (tools/halotime.py, plain + xf3)
mkdir -p gpurun_out/r2_halopmc
O=$PWD/gpurun_out/r2_halopmc
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python $R/tools/halotime.py 0,3 > $O/p$i.log 2>&1
done
cd $R
python - <<'PY'
import csv,collections,glob
for f in sorted(glob.glob('gpurun_out/r2_halopmc/p*/p_counter_collection.csv')):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'halo_kernel' in r['Kernel_Name']:
            k=r['Kernel_Name'].split('halo_kernel')[1][:28]+' g'+r['Grid_Size']
            d[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for g,c in sorted(d.items()):
        print(f.split('/')[-2],g,{k.replace('SQ_',''):"%.3g"%(sum(v)/len(v)) for k,v in c.items()}, "n=",len(next(iter(c.values()))))
PY
