#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s39"; mkdir -p "$O"
for i in 1 2 3; do for v in "" E1; do
  if [ -n "$v" ]; then export DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_$v.alt"; else unset DC_DDIM_LIB; fi
  echo -n "variant ${v:-default}: "; python tools/profile_encoder.py 2>/dev/null | tail -1
done; done > "$O/ab.txt" 2>&1; cat "$O/ab.txt"
export DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_E1.alt"
timeout 600 python -m pytest tests -m gpu -q -k "encode_music or music" 2>&1 | tail -2
