#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (status=transient: nothing charged).  usage: tools/gpurun_retry.sh <timeout> '<command>'
# Exit status: gpurun's own (0 = the command ran and succeeded); 75 when every attempt was turned away.
for i in 1 2 3 4 5 6 7 8; do
  out=$(/usr/local/graft/bin/gpurun --timeout "$1" -- "$2" 2>&1)
  rc=$?
  echo "$out" | tail -60
  echo "$out" | grep -q "status=transient" || exit $rc
  sleep 120
done
echo "gpurun_retry: no GPU slot after 8 attempts" >&2
exit 75
