#!/bin/bash
# Run ON THE GPU BOX via gpurun: probe + GPU test suite + default bench, each with its own log under gpurun_out/.
TAG="${1:-s1}"
mkdir -p gpurun_out
timeout 300 ./tools/probe_fused > gpurun_out/probe_$TAG.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_stages.py -m gpu -q -s > gpurun_out/t_stages_$TAG.log 2>&1
echo "stages rc=$?" >> gpurun_out/t_stages_$TAG.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s > gpurun_out/t_parity_$TAG.log 2>&1
echo "parity rc=$?" >> gpurun_out/t_parity_$TAG.log
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench rc=$?" >> gpurun_out/bench_$TAG.err
tail -3 gpurun_out/t_stages_$TAG.log gpurun_out/t_parity_$TAG.log
cat gpurun_out/probe_$TAG.log
cat gpurun_out/bench_$TAG.json
