#!/bin/bash
# round 3, bring-up of conv3x3_v3: correctness vs the first halo kernel, cycle stamps, SQ counters (v3 and old, 256 ch @ 64x64)
O=$PWD/gpurun_out/r3_v3a
mkdir -p $O
R=$PWD
timeout 300 python tools/v3check.py quick > $O/check.log 2>&1
timeout 200 python tools/v3dbg.py > $O/dbg.log 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  for which in v3 old; do
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i$which -o p -- python $R/tools/v3pmc.py $which 256 64 > $O/p$i$which.log 2>&1
  done
done
cd $R
python - <<'PY' > $O/pmc.txt 2>&1
import csv,collections,glob
for f in sorted(glob.glob('gpurun_out/r3_v3a/p*/p_counter_collection.csv')):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'conv3x3' in r['Kernel_Name']:
            k=r['Kernel_Name'].split('conv3x3')[1][:40]+' g'+r['Grid_Size']
            d[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for g,c in sorted(d.items()):
        print(f.split('/')[-2],g,{k.replace('SQ_',''):"%.4g"%(sum(v)/len(v)) for k,v in c.items()}, "n=",len(next(iter(c.values()))))
PY
rm -rf $O/p1v3 $O/p1old $O/p2v3 $O/p2old
cat $O/check.log | tail -40; cat $O/dbg.log; cat $O/pmc.txt
