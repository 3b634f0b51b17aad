#!/bin/bash
TAG="${1:-s5}"
mkdir -p gpurun_out
timeout 600 python bench.py --bs 1 --no-cpu-baseline --no-extras > gpurun_out/bench_bs1_$TAG.json 2> gpurun_out/bench_bs1_$TAG.err
grep "profile pass\|timed region" gpurun_out/bench_bs1_$TAG.err
timeout 600 python bench.py --bs 4 --no-cpu-baseline --no-extras > gpurun_out/bench_bs4_$TAG.json 2> gpurun_out/bench_bs4_$TAG.err
grep "profile pass\|timed region" gpurun_out/bench_bs4_$TAG.err
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/t_all_$TAG.log 2>&1
tail -3 gpurun_out/t_all_$TAG.log
