#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s9"; mkdir -p "$O"
timeout 1200 python tools/parity_bs32_all_clips.py > "$O/parity_bs32_all_clips.txt" 2>&1; cat "$O/parity_bs32_all_clips.txt" | grep -v amdgpu
