#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s36"; mkdir -p "$O"
for i in 1 2 3; do for v in "" N1; do
  if [ -n "$v" ]; then export DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_$v.alt"; else unset DC_DDIM_LIB; fi
  echo -n "noeff ${v:-default}: "; python bench.py --no-eff --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep -o "k_layer[a-z_]* [0-9.]*ms\|ms_per_step\": [0-9.]*" | tr "\n" " "; echo
done; done > "$O/ab.txt" 2>&1; cat "$O/ab.txt"
