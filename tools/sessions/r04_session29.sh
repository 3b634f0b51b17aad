#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s29"; mkdir -p "$O"
export TMPDIR=/tmp
for b in 1 4; do
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_bs$b" -- /usr/bin/python3 "$R/tools/time_small_batch.py" $b > "$O/kt_bs$b.log" 2>&1 )
python3 tools/pmc_summary.py "$O/kt_bs$b" > "$O/summary_bs$b.txt" 2>&1; head -12 "$O/summary_bs$b.txt"
done
find "$O" -name "*kernel_trace.csv" -delete
