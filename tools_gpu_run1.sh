#!/bin/bash
# first GPU bring-up: every test group in its own process (a faulting kernel must not poison the rest)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8 > gpurun_out/dev.log 2>&1
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" >> gpurun_out/dev.log 2>&1
for k in "test_conv_forward" "test_conv_epilogue or test_stem_conv or test_fc_as_conv" "test_conv_dgrad" "test_conv_transpose" "test_conv_wgrad" \
         "test_batchnorm or test_bn_relu_maxpool" "test_upsample2x or test_groupnorm or test_leaky" "test_head_tail"; do
  echo "=== -k $k" >> gpurun_out/t1.log
  timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "$k" --timeout=300 -p no:cacheprovider 2>&1 | tail -60 >> gpurun_out/t1.log
done
for k in "test_pose_decode or test_ranger" "test_fp32_train_step" "test_fp32_maps" "test_bf16_train_step" "test_vs_oracle" "test_full_size"; do
  echo "=== e2e -k $k" >> gpurun_out/t1.log
  timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -s -k "$k" --timeout=600 -p no:cacheprovider 2>&1 | tail -80 >> gpurun_out/t1.log
done
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke1.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench1.log 2>&1
tail -5 gpurun_out/bench1.log
grep -E "passed|failed|error" gpurun_out/t1.log | tail -30
