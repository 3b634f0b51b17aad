#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s19"; mkdir -p "$O"
for i in 1 2; do python bench.py --no-cpu-baseline 2> "$O/bench_$i.err" | tail -1 > "$O/bench_$i.json"; grep "end to end" "$O/bench_$i.err"; done
timeout 300 python tools/time_e2e.py "" > "$O/e2e.txt" 2>&1; cat "$O/e2e.txt"
