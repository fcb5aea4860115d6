#!/bin/bash
# round 3, conv3x3_v3: 8-wave (12) against 4-wave (34) tile configurations -- correctness, launch times, cycle stamps
O=$PWD/gpurun_out/r3_v3b
mkdir -p $O
timeout 400 python tools/v3check.py full 12,34 > $O/check.log 2>&1
timeout 200 python tools/v3dbg.py 12,34 > $O/dbg.log 2>&1
grep -v "^ok" $O/check.log; cat $O/dbg.log
