#!/bin/bash
# one PMC pass over a short bench run: per-kernel wave-state / LDS counters for every kernel of the step
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_step -o p -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/pmc_step.log 2>&1
cd $R
python - <<'PY'
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/pmc_step/p_counter_collection.csv')))
d=collections.defaultdict(lambda: collections.defaultdict(float))
cnt=collections.Counter()
for r in rows:
    k=r['Kernel_Name'][:70]
    d[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVE_CYCLES': cnt[k]+=1
out=[]
for k,c in d.items():
    w=c.get('SQ_WAVE_CYCLES',0)
    if w<=0: continue
    out.append((w,k,cnt[k],c))
out.sort(reverse=True)
for w,k,n,c in out[:22]:
    print("%-70s n=%4d wave=%8.1fM act=%4.0f%% waitinst=%4.0f%% wait=%4.0f%% ldsconf/ldsact=%5.1f%% mfma/busy=%5.2f"%(k,n,w/1e6,100*c['SQ_ACTIVE_INST_ANY']/w,100*c['SQ_WAIT_INST_ANY']/w,100*c['SQ_WAIT_ANY']/w,100*c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1), c['SQ_VALU_MFMA_BUSY_CYCLES']/max(c['SQ_BUSY_CYCLES'],1)))
PY
