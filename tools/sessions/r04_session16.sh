#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s16"; mkdir -p "$O"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_robust.py -x -q -k "layer16 or config1_golden or small or invarian or narrow or clip_layouts or status" > "$O/pytest.txt" 2>&1; tail -3 "$O/pytest.txt"
for i in 1 2; do
for v in "" H; do
  if [ -n "$v" ]; then export DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_$v.alt"; else unset DC_DDIM_LIB; fi
  echo "== ${v:-default}"; timeout 300 python tools/time_small_batch.py 1 2 3 4 2>&1 | grep "bs="
done; done > "$O/small_fuse_extra.txt" 2>&1; cat "$O/small_fuse_extra.txt"
unset DC_DDIM_LIB
tools/ab.sh run H > "$O/ab_headline.txt" 2>&1; cat "$O/ab_headline.txt"
