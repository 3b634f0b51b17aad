#!/bin/bash
# round 4, GPU session 8: k_layer16 with two record batches up front: tests, timing, stamps
R="$(pwd)"; O="$R/gpurun_out/r04_s8"; mkdir -p "$O"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -s -k "layer16 or clip_layouts or config1_golden" > "$O/pytest_layer16.txt" 2>&1; echo "pytest rc=$?" | tee -a "$O/pytest_layer16.txt"
grep -E "layer16 B=|passed|failed|Error|error" "$O/pytest_layer16.txt" | head -30
timeout 600 python tools/time_small_batch.py 1 2 4 8 > "$O/time_small_batch.txt" 2>&1; cat "$O/time_small_batch.txt"
DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_L.alt" DC_L16_STAMPS=1 DC_DISABLE_GRAPH=1 timeout 300 python tools/stage_stamps16.py 1 > "$O/stamps16_bs1.txt" 2>&1; cat "$O/stamps16_bs1.txt"
