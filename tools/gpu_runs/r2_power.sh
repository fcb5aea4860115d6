#!/bin/bash
# is the step power-limited?  sample socket power / clocks while the train step loops
mkdir -p gpurun_out/r2_power
O=gpurun_out/r2_power
python bench.py --steps 5000 --warmup 20 --no-cpu-baseline --no-roofline --no-extras > $O/bench_long.json 2> $O/bench_long.err &
BP=$!
sleep 28
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk\|Temperature (Sensor junction)\|edge" | tr '\n' ';' >> $O/smi.log; echo >> $O/smi.log
  sleep 1.0
done
wait $BP
cat $O/bench_long.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('long run', d['ms_per_step'])"
cat $O/smi.log | cut -c1-400
rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | tr '\n' ';'; echo " (idle)"
rocm-smi --showmaxpower 2>/dev/null | grep -i power
