#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s24"; mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -q --durations=40 > "$O/pytest_durations.txt" 2>&1; tail -50 "$O/pytest_durations.txt"
