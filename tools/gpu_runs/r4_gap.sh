#!/bin/bash
# round 4: is the main stream's idle gap in front of head_tail_bwd (profiles/r04_side_stream_overlap: 83 us on one box, 6-11 us before) systematic?
O=gpurun_out/r4_gap
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 0; do
  rm -rf /tmp/gp$v
  GDRN_FC2_SPLITK=$v timeout 300 rocprofv3 --kernel-trace -d /tmp/gp$v -o t --output-format csv -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/$O/bench$v.log 2>&1
  f=$(find /tmp/gp$v -name "*kernel_trace.csv" | head -1)
  echo "== GDRN_FC2_SPLITK=$v"; python $R/tools/gap_trace.py $f 5 15
done 2>&1 | tee $R/$O/gaps.txt
