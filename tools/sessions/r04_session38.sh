#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s38"; mkdir -p "$O"
export DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_W1.alt"
for i in 1 2 3 4 5 6; do for v in "" DC_FUSE_WIDE_EXTRA=1; do
  echo -n "${v:-prepended}: "; env $v python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | grep -o "k_film_gemm [0-9.]*ms\|ms_per_step\": [0-9.]*" | tr "\n" " "; echo
done; done > "$O/ab.txt" 2>&1; cat "$O/ab.txt"
