#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s15"; mkdir -p "$O"
for b in 1 4; do
DC_STAMP_BS=$b DC_STAMPS=1 DC_DISABLE_GRAPH=1 timeout 300 python tools/stage_stamps.py > "$O/stamps_bs$b.txt" 2>&1; grep -A30 "^k_film_gemm:" "$O/stamps_bs$b.txt"
done
