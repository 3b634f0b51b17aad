#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s31"; mkdir -p "$O"
tools/ab.sh run R1 > "$O/ab_fp16.txt" 2>&1; cat "$O/ab_fp16.txt"
for i in 1 2; do for v in "" R1; do
  if [ -n "$v" ]; then export DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_$v.alt"; else unset DC_DDIM_LIB; fi
  echo -n "mixed ${v:-default}: "; python bench.py --precision mixed --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep -o "k_layer [0-9.]*ms\|ms_per_step\": [0-9.]*" | tr "\n" " "; echo
done; done > "$O/ab_mixed.txt" 2>&1; cat "$O/ab_mixed.txt"
