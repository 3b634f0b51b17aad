#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s34"; mkdir -p "$O"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -x -q -k "config1_golden or forward_golden or bs32_full or every_clip or clip_layouts or odd_shapes or stage_taps or block_golden or straddling or record_path" > "$O/pytest.txt" 2>&1; tail -2 "$O/pytest.txt"
run() { for i in 1 2 3; do for v in "" C0; do
  if [ -n "$v" ]; then export DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_$v.alt"; else unset DC_DDIM_LIB; fi
  echo -n "$1 ${v:-default}: "; python bench.py $2 --no-cpu-baseline --no-extras 2>&1 | grep -o "k_layer[a-z_]* [0-9.]*ms\|ms_per_step\": [0-9.]*" | tr "\n" " "; echo
done; done; }
{ run fp16 "--steps 10 --warmup 2"; run mixed "--precision mixed --steps 5 --warmup 1"; run t900 "--frames 900 --bs 128 --steps 5 --warmup 1"; } > "$O/ab.txt" 2>&1; cat "$O/ab.txt"
