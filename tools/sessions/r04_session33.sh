#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s33"; mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "config1_golden or forward_golden or bs32_full or clip_layouts or narrow or no_eff_ddim50_long" > "$O/pytest.txt" 2>&1; tail -2 "$O/pytest.txt"
run() { for i in 1 2 3; do for v in "" P0; do
  if [ -n "$v" ]; then export DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_$v.alt"; else unset DC_DDIM_LIB; fi
  echo -n "$1 ${v:-default}: "; python bench.py $2 --no-cpu-baseline --no-extras 2>&1 | grep -o "k_layer[a-z_]* [0-9.]*ms\|ms_per_step\": [0-9.]*" | tr "\n" " "; echo
done; done; }
{ run fp16 "--steps 10 --warmup 2"; run mixed "--precision mixed --steps 5 --warmup 1"; run noeff "--no-eff --steps 3 --warmup 1"; run t900 "--frames 900 --bs 128 --steps 5 --warmup 1"; } > "$O/ab.txt" 2>&1; cat "$O/ab.txt"
unset DC_DDIM_LIB
for v in "" P0; do
  if [ -n "$v" ]; then export DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_$v.alt"; else unset DC_DDIM_LIB; fi
  echo "== ${v:-default}"; timeout 300 python tools/time_small_batch.py 1 8 12 16 2>&1 | grep "bs=" | sed 's/(\[[^]]*\])//g'
done > "$O/small.txt" 2>&1; cat "$O/small.txt"
