#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s21"; mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_robust.py -x -q -k "pinned or harness or rccl or music or encode or visualize or evaluation" > "$O/pytest.txt" 2>&1; tail -3 "$O/pytest.txt"
DC_DIAG_ALONE=1 DC_DIAG_N=8 python tools/diag_e2e.py > "$O/diag_alone.txt" 2>&1; tail -9 "$O/diag_alone.txt"
for i in 1 2 3; do python bench.py --no-cpu-baseline 2> "$O/bench_$i.err" | tail -1 > "$O/bench_$i.json"; grep "end to end" "$O/bench_$i.err"; done
