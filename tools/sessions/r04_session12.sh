#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s12"; mkdir -p "$O"
tools/ab.sh run E0 > "$O/ab_early_args_1.txt" 2>&1; tools/ab.sh run E0 > "$O/ab_early_args_2.txt" 2>&1; cat "$O"/ab_early_args_*.txt
for v in "" E0; do
  if [ -n "$v" ]; then export DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_$v.alt"; else unset DC_DDIM_LIB; fi
  DC_STAMPS=1 DC_DISABLE_GRAPH=1 python tools/stage_stamps.py > "$O/stamps_${v:-default}.txt" 2>&1; echo "== ${v:-default}"; grep -A4 "per-workgroup stamps" "$O/stamps_${v:-default}.txt"; sed -n 2,4p "$O/stamps_${v:-default}.txt"
done
