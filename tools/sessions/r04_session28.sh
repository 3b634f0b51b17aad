#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s28"; mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "layer16" > "$O/pytest.txt" 2>&1; tail -3 "$O/pytest.txt"
for i in 1 2; do
for v in "" H; do
  if [ -n "$v" ]; then export DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_$v.alt"; else unset DC_DDIM_LIB; fi
  echo "== ${v:-default}"; timeout 300 python tools/time_small_batch.py 1 4 8 2>&1 | grep "bs=" | sed 's/(\[[^]]*\])//g'
done; done > "$O/small_lds78.txt" 2>&1; cat "$O/small_lds78.txt"
unset DC_DDIM_LIB
echo "== DC_LAYER16_CUS=2" > "$O/small_2percu.txt"
DC_LAYER16_CUS=2 timeout 600 python tools/time_small_batch.py 9 10 12 14 16 17 2>&1 | grep "bs=" | sed 's/(\[[^]]*\])//g' >> "$O/small_2percu.txt"; cat "$O/small_2percu.txt"
