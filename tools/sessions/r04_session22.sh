#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s22"; mkdir -p "$O"
DC_DIAG_ALONE=1 DC_DIAG_N=40 python tools/diag_e2e.py > "$O/diag_alone.txt" 2>&1; grep -B1 -A1 "slow" "$O/diag_alone.txt" | head -40
