#!/bin/bash
# round 4, GPU session 7: stage stamps of k_layer16; narrow 32-token form without the AGPR half (same-box timing at bs=12, 16)
R="$(pwd)"; O="$R/gpurun_out/r04_s7"; mkdir -p "$O"
DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_L.alt" DC_L16_STAMPS=1 DC_DISABLE_GRAPH=1 timeout 300 python tools/stage_stamps16.py 1 > "$O/stamps16_bs1.txt" 2>&1; cat "$O/stamps16_bs1.txt"
DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_L.alt" DC_L16_STAMPS=1 DC_DISABLE_GRAPH=1 timeout 300 python tools/stage_stamps16.py 8 > "$O/stamps16_bs8.txt" 2>&1; tail -21 "$O/stamps16_bs8.txt"
timeout 600 python tools/time_small_batch.py 1 12 16 > "$O/time_small_batch.txt" 2>&1; cat "$O/time_small_batch.txt"
