#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s20"; mkdir -p "$O"
python tools/diag_e2e.py > "$O/diag_plain.txt" 2>&1; cat "$O/diag_plain.txt" | tail -32
DC_DIAG_ALONE=1 DC_DIAG_N=12 python tools/diag_e2e.py > "$O/diag_alone.txt" 2>&1; tail -13 "$O/diag_alone.txt"
