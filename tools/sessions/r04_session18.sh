#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s18"; mkdir -p "$O"
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$O/kt_e2e" -- /usr/bin/python3 "$R/tools/time_e2e.py" "" > "$O/kt_e2e.log" 2>&1 )
tail -3 "$O/kt_e2e.log"; ls -la "$O"/kt_e2e/*/
