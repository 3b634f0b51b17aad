#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s13"; mkdir -p "$O"
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "layer16 or config1_golden or forward_golden or bs32_full or clip_layouts" > "$O/pytest.txt" 2>&1; tail -3 "$O/pytest.txt"
tools/ab.sh run E0 > "$O/ab_early_args.txt" 2>&1; cat "$O/ab_early_args.txt"
for v in "" E0; do
  if [ -n "$v" ]; then export DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_$v.alt"; else unset DC_DDIM_LIB; fi
  echo "== ${v:-default}"; timeout 300 python tools/time_small_batch.py 1 4 2>&1 | grep "bs="
done > "$O/small_early_args.txt" 2>&1; cat "$O/small_early_args.txt"
