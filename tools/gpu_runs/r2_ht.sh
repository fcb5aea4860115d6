#!/bin/bash
mkdir -p gpurun_out/r2_ht
timeout 600 python tools/halotime.py > gpurun_out/r2_ht/halotime.log 2>&1
cat gpurun_out/r2_ht/halotime.log
