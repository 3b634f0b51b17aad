#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (status=transient: nothing charged).  usage: tools/gpurun_retry.sh <timeout> '<command>'
for i in 1 2 3 4 5 6 7 8; do
  out=$(/usr/local/graft/bin/gpurun --timeout "$1" -- "$2" 2>&1)
  echo "$out" | tail -60
  echo "$out" | grep -q "status=transient" || exit 0
  sleep 120
done
