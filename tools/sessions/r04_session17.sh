#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s17"; mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "rides_in_the_film" > "$O/pytest.txt" 2>&1; tail -3 "$O/pytest.txt"
timeout 600 python tools/time_e2e.py "" "4,28" "2,6,24" "16,16" "4,12,16" > "$O/e2e_chunks.txt" 2>&1; cat "$O/e2e_chunks.txt"
