#!/bin/bash
# round-6 profile set of the current build (copied into profiles/ by hand afterwards):
#   bench line (driver-sized) + per-launch table | steady-state kernel tables (default = side stream, and GDRN_WGRAD_STREAM=0 serial: per-kernel
#   durations that match bench.py's roofline brackets) | side-stream overlap | MFMA utilisation / HBM bandwidth per kernel | inference |
#   v3 vs first halo kernel, v3 cycle stamps, MFMA issue-rate micro-benchmark | CPU thread sweep
O=$PWD/gpurun_out/r6_prof
mkdir -p $O
R=$PWD
export PYTHONUNBUFFERED=1
f() { if ls $O/$1/*/p_$2.csv >/dev/null 2>&1; then ls $O/$1/*/p_$2.csv | head -1; else echo $O/$1/p_$2.csv; fi; }
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --no-extras"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- $B --steps 8 --warmup 3 > $O/trace.log 2>&1
GDRN_WGRAD_STREAM=serial timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o p -- $B --steps 8 --warmup 3 > $O/serial.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/infer -o p -- $B --fwd-only --steps 12 --warmup 3 > $O/infer.log 2>&1
B2="$B --steps 2 --warmup 2"
GDRN_WGRAD_STREAM=serial timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -o p -- $B2 > $O/sq.log 2>&1
GDRN_WGRAD_STREAM=serial timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $B2 > $O/fetch.log 2>&1
GDRN_WGRAD_STREAM=serial timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $B2 > $O/write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/cal -o p -- python $R/tools/ubench/mfma_rate.py > $O/cal.log 2>&1
cd $R
python tools/trace_steps.py $(f trace kernel_trace) 5 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-extras (round 6; bs=64 bf16 train step, default = bucket-end work on the side stream: durations of overlapped kernels include the sharing)" > $O/r06_kernel_steps_bs64_bf16.txt 2>&1
python tools/trace_steps.py $(f serial kernel_trace) 5 "GDRN_WGRAD_STREAM=serial rocprofv3 --kernel-trace --stats -- python bench.py --steps 8 --warmup 3 ... (round 6; the launches of the default step on ONE stream: stand-alone kernel durations, the ones bench.py's roofline brackets measure)" > $O/r06_kernel_steps_serial_bs64_bf16.txt 2>&1
python tools/overlap_trace.py $(f trace kernel_trace) > $O/r06_side_stream_overlap_bs64_bf16.txt 2>&1
python tools/step_timeline.py $(f trace kernel_trace) 2 > $O/r06_step_timeline_bs64_bf16.txt 2>&1
python tools/step_timeline.py $(f infer kernel_trace) 2 > $O/r06_inference_timeline_bs64_bf16.txt 2>&1
python tools/pmc_util.py $(f sq counter_collection) $(f fetch counter_collection) $(f write counter_collection) $(f serial kernel_trace) $(f cal counter_collection) $O/r06_mfma_util_hbm_bs64_bf16 > /dev/null 2>&1
python - <<PY > $O/r06_inference_steps_bs64_bf16.txt 2>&1
import csv, collections
f = "$(f infer kernel_trace)"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "pack_image" in r["Kernel_Name"]]
n = 8
seg = rows[marks[-n - 1]:marks[-1]]
per = collections.OrderedDict()
for r in seg:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("unsigned short", "bf16")[:90]
    d = per.setdefault(k, [0, 0.0]); d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
tot = sum(v[1] for v in per.values()) / n
span = (int(rows[marks[-1]]["Start_Timestamp"]) - int(rows[marks[-n-1]]["Start_Timestamp"])) * 1e-6 / n
print("# rocprofv3 --kernel-trace -- python bench.py --fwd-only --steps 12 --warmup 3 (round 6): eval-mode inference bs=64, last %d forwards:" % n)
print("# %.1f launches, %.3f ms kernel time, %.3f ms wall per forward = %.0f TFLOP/s (22.823 GFLOP per RoI)" % (len(seg) / n, tot, span, 64 * 22.823e9 / (span * 1e-3) / 1e12))
print("%-92s %6s %8s %8s" % ("kernel", "n/fwd", "ms/fwd", "avg us"))
for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("%-92s %6.1f %8.3f %8.1f" % (k, c / n, t / n, t / c * 1e3))
PY
cp $(f trace kernel_stats) $O/r06_kernel_stats_bs64_bf16.csv 2>/dev/null
cp $(f serial kernel_stats) $O/r06_kernel_stats_serial_bs64_bf16.csv 2>/dev/null
cd /tmp
GDRN_BUCKETS=5 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/dp -o p -- python $R/bench.py --no-cpu-baseline --no-roofline --no-extras --steps 8 --warmup 3 --dist-force > $O/dp.log 2>&1
cd $R
python tools/bucket_timeline.py $(f dp kernel_trace) > $O/r06_bucket_timeline_bs64_bf16.txt 2>&1
rm -rf $O/trace $O/serial $O/infer $O/sq $O/fetch $O/write $O/cal $O/dp
timeout 100 python tools/ubench/mfma_rate.py 2>&1 | grep -v "amdgpu.ids" > $O/r06_mfma_rate_ubench.txt
[ "$SWEEP" = "1" ] && timeout 500 python bench.py --cpu-threads-sweep > $O/r06_cpu_threads_sweep.txt 2>&1   # ~4 minutes of host time: only on request
# MFMA utilisation / HBM traffic per kernel of the eval-mode forward (the eight-wave form of the small-map tile runs there)
cd /tmp
BI="python $R/bench.py --no-cpu-baseline --no-roofline --no-extras --fwd-only"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/iplain -o p -- $BI --steps 12 --warmup 3 > $O/iplain.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/isq -o p -- $BI --steps 3 --warmup 2 > $O/isq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/ifetch -o p -- $BI --steps 3 --warmup 2 > $O/ifetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/iwrite -o p -- $BI --steps 3 --warmup 2 > $O/iwrite.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/ical -o p -- python $R/tools/ubench/mfma_rate.py > $O/ical.log 2>&1
cd $R
python tools/pmc_util.py $(f isq counter_collection) $(f ifetch counter_collection) $(f iwrite counter_collection) $(f iplain kernel_trace) $(f ical counter_collection) $O/r06_mfma_util_hbm_inference_bs64_bf16 "bs=64 bf16 eval-mode forward (python bench.py --fwd-only)" > /dev/null 2>&1
rm -rf $O/iplain $O/isq $O/ifetch $O/iwrite $O/ical
# the driver-sized bench line last (it reports roofline.traffic / mfma_util from the PMC summary above only if that summary sits in profiles/)
cp $O/r06_mfma_util_hbm_bs64_bf16.json $O/r06_mfma_util_hbm_bs64_bf16.txt profiles/
GDRN_LAYER_TABLE=$O/r06_layer_table_bs64_bf16.txt timeout 900 python bench.py > $O/r06_bench_bs64_bf16.json 2> $O/bench.err
cut -c1-600 $O/r06_bench_bs64_bf16.json
