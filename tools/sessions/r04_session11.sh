#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s11"; mkdir -p "$O"
tools/ab_env.sh "" HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0 > "$O/ab_dev_kernarg.txt" 2>&1; cat "$O/ab_dev_kernarg.txt"
for v in "" HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0; do echo "== ${v:-default}"; env $v timeout 300 python tools/time_small_batch.py 1 2>&1 | grep "bs="; done > "$O/small_dev_kernarg.txt" 2>&1; cat "$O/small_dev_kernarg.txt"
