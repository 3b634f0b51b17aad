#!/bin/bash
TAG="${1:-s2}"
mkdir -p gpurun_out
timeout 300 ./tools/probe_fused > gpurun_out/probe_$TAG.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_stages.py -m gpu -q -s > gpurun_out/t_stages_$TAG.log 2>&1
echo "stages rc=$?" >> gpurun_out/t_stages_$TAG.log
cat gpurun_out/probe_$TAG.log
grep -E "G3 |blocks 1|stage taps|config |visualize|passed|failed|NCCL|Error|error" gpurun_out/t_stages_$TAG.log | head -60
