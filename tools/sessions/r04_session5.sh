#!/bin/bash
# round 4, GPU session 5: 16-token layer kernel with the one-round-trip combine: tests + same-box timing; no_eff reference-point moves
R="$(pwd)"; O="$R/gpurun_out/r04_s5"; mkdir -p "$O"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -s -k "layer16 or clip_layouts or narrow_workgroups or config1_golden or peaky" > "$O/pytest_layer16.txt" 2>&1; echo "pytest rc=$?" | tee -a "$O/pytest_layer16.txt"
grep -E "layer16 B=|peaky|passed|failed|Error|error" "$O/pytest_layer16.txt" | head -30
timeout 600 python tools/time_small_batch.py 1 2 4 8 > "$O/time_small_batch.txt" 2>&1; cat "$O/time_small_batch.txt"
DC_T=900 timeout 600 python tools/time_small_batch.py 1 4 16 >> "$O/time_small_batch.txt" 2>&1; tail -3 "$O/time_small_batch.txt"
DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_M.alt" timeout 600 python tools/noeff_moves.py > "$O/noeff_moves.txt" 2>&1; cat "$O/noeff_moves.txt"
