#!/bin/bash
# round 4, GPU session 2: the whole GPU suite on the new ABI; tracked-prefetch A/B (4 pairs); default bench with the extras
R="$(pwd)"; O="$R/gpurun_out/r04_s2"; mkdir -p "$O"
python -m pytest tests -m gpu -x -q -s > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?" | tee -a "$O/pytest_gpu.txt"
tail -5 "$O/pytest_gpu.txt"
tools/ab.sh run S > "$O/ab_S_1.txt" 2>&1; tools/ab.sh run S > "$O/ab_S_2.txt" 2>&1
cat "$O"/ab_S_*.txt
python bench.py > "$O/bench_default.json" 2> "$O/bench_default.log"; tail -3 "$O/bench_default.log"; cat "$O/bench_default.json"
