#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s27"; mkdir -p "$O"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_robust.py -x -q -k "rides_in_the_film or layer16 or narrow or invarian or clip_layouts or status or graph" > "$O/pytest.txt" 2>&1; tail -3 "$O/pytest.txt"
