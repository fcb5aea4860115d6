cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ksweep -o p -- python $R/tools/halo_ksweep.py > $R/gpurun_out/ksweep.log 2>&1
cd $R
python - <<'PY'
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/ksweep/p_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in rows:
    if 'conv3x3_halo' in r['Kernel_Name']:
        d[r['Grid_Size_X']+"/"+r['LDS_Block_Size']].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
seq=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if 'conv3x3_halo' in r['Kernel_Name']]
print(["%.1f"%x for x in seq])
PY
