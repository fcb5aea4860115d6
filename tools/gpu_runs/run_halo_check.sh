#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "halo or conv" 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_hc -o p -- python $R/tools/halobench.py > $R/gpurun_out/pmc_hc.log 2>&1
cd $R
python - <<'PY'
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/pmc_hc/p_counter_collection.csv')))
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'conv3x3_halo' not in r['Kernel_Name']: continue
    key=(r['Kernel_Name'].split('<')[1].split('>')[0], r['Grid_Size'])
    d[key][r['Counter_Name']].append(float(r['Counter_Value']))
    d[key]['dur_us'].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3*1e6)
for key,c in d.items():
    print(key, "  ".join(f"{k}={sum(v)/len(v)/1e6:.3f}" for k,v in sorted(c.items())))
PY
