#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s35"; mkdir -p "$O"
for i in 1 2 3 4; do for v in "" R4; do
  if [ -n "$v" ]; then export DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_$v.alt"; else unset DC_DDIM_LIB; fi
  echo -n "variant ${v:-default}: "; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | grep -o "k_layer [0-9.]*ms\|ms_per_step\": [0-9.]*" | tr "\n" " "; echo
done; done > "$O/ab.txt" 2>&1; cat "$O/ab.txt"
