#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s37"; mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -q -k "no_eff or noeff or full" > "$O/pytest.txt" 2>&1; tail -2 "$O/pytest.txt"
python bench.py --no-eff --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > "$O/bench_noeff.json"; grep -o "ms_per_step\": [0-9.]*" "$O/bench_noeff.json"
