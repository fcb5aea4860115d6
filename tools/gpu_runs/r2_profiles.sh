#!/bin/bash
# round-2 profile set of the current build: driver-sized bench + per-launch table, kernel stats, 5-bucket timeline, SQ PMC pass, HBM traffic passes
mkdir -p gpurun_out/r2_prof
O=gpurun_out/r2_prof
export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/$O/stats.log 2>&1
GDRN_BUCKETS=5 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/b5 -o p -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/$O/b5.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/pmc_sq -o p -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $R/$O/pmc_sq.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc_$c -o p -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $R/$O/pmc_$c.log 2>&1
done
cd $R
python tools/summarize_stats.py $O/stats/p_kernel_stats.csv 13 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras (round 2; bs=64 bf16 train step, 13 profiled steps)" > $O/kernel_stats.txt
python tools/bucket_timeline.py $O/b5/p_kernel_trace.csv > $O/bucket_timeline.txt 2>&1
cat $O/bucket_timeline.txt
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv $O/r02_hbm_traffic_bs64_bf16 > /dev/null 2>&1; head -12 $O/r02_hbm_traffic_bs64_bf16.txt
python - <<'PY'
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/r2_prof/pmc_sq/p_counter_collection.csv')))
d=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').replace('unsigned short','bf16')[:64]
    d[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVE_CYCLES': cnt[k]+=1
out=[(c.get('SQ_WAVE_CYCLES',0),k,cnt[k],c) for k,c in d.items() if c.get('SQ_WAVE_CYCLES',0)>0]
out.sort(reverse=True)
with open('gpurun_out/r2_prof/pmc_sq_summary.txt','w') as f:
    f.write("# rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -- python bench.py --steps 2 --warmup 2 (whole train step, round-2 build; sums over 4 profiled steps)\n")
    f.write("# act / waitinst / wait: share of SQ_WAVE_CYCLES (quad-cycles); mfma/busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES\n")
    for w,k,n,c in out[:30]:
        f.write("%-64s n=%4d wave=%9.1fM act=%4.0f%% waitinst=%4.0f%% wait=%4.0f%% ldsconf/ldsact=%5.1f%% mfma/busy=%5.2f\n"%(k,n,w/1e6,100*c['SQ_ACTIVE_INST_ANY']/w,100*c['SQ_WAIT_INST_ANY']/w,100*c['SQ_WAIT_ANY']/w,100*c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1), c['SQ_VALU_MFMA_BUSY_CYCLES']/max(c['SQ_BUSY_CYCLES'],1)))
print(open('gpurun_out/r2_prof/pmc_sq_summary.txt').read()[:3500])
PY
rm -rf $O/pmc_sq/*.csv.bak
# the driver-sized bench line last: it reports roofline.traffic from the PMC summary above only if that summary is committed under
# profiles/ with this build's source hash (copy it there first when regenerating the set)
cp $O/r02_hbm_traffic_bs64_bf16.json $O/r02_hbm_traffic_bs64_bf16.txt profiles/
GDRN_LAYER_TABLE=$O/layers.txt timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cut -c1-400 $O/bench.json
