#!/bin/bash
# round 4, GPU session 4: the 16-token layer kernel for small batches: tests, then same-box timing against the 32-token narrow form
R="$(pwd)"; O="$R/gpurun_out/r04_s4"; mkdir -p "$O"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -s -k "layer16 or clip_layouts or narrow_workgroups or config1_golden or forward_golden" > "$O/pytest_layer16.txt" 2>&1; echo "pytest rc=$?" | tee -a "$O/pytest_layer16.txt"
grep -E "layer16 B=|passed|failed|Error|error" "$O/pytest_layer16.txt" | head -30
timeout 900 python -m pytest tests/test_gpu_robust.py -x -q -k "seeded or eta_needs or status_word or smoothing" >> "$O/pytest_layer16.txt" 2>&1; tail -3 "$O/pytest_layer16.txt"
timeout 600 python tools/time_small_batch.py 1 2 4 8 > "$O/time_small_batch.txt" 2>&1; cat "$O/time_small_batch.txt"
DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_M.alt" timeout 600 python tools/noeff_moves.py > "$O/noeff_moves.txt" 2>&1; cat "$O/noeff_moves.txt"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "g10" > "$O/pytest_g10.txt" 2>&1; grep -E "no_eff DDIM|passed|failed" "$O/pytest_g10.txt"
