#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s23"; mkdir -p "$O"
for q in "" 8 2; do
  echo "== GPU_MAX_HW_QUEUES=${q:-default}"
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  DC_DIAG_ALONE=1 DC_DIAG_N=60 python tools/diag_e2e.py > "$O/diag_q$q.txt" 2>&1; grep -c "slow" "$O/diag_q$q.txt"; grep "slow" "$O/diag_q$q.txt" | head -8
  awk '/^call/ {print $4}' "$O/diag_q$q.txt" | sort -n | awk '{a[NR]=$1} END {print "calls", NR, "median", a[int(NR/2)+1], "p90", a[int(NR*0.9)], "max", a[NR]}'
done
