#!/bin/bash
# round 4, GPU session 3: the whole GPU suite after the PERS removal / tracked prefetch; mixed-mode stylization prefetch A/B
R="$(pwd)"; O="$R/gpurun_out/r04_s3"; mkdir -p "$O"
python -m pytest tests -m gpu -q -s > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?" | tee -a "$O/pytest_gpu.txt"
tail -8 "$O/pytest_gpu.txt"
BENCH_ARGS="--precision mixed" tools/ab_env.sh "" "DC_DDIM_LIB=$R/diffusion-conductor_amd/libdc_ddim_P1.alt" "DC_DDIM_LIB=$R/diffusion-conductor_amd/libdc_ddim_P0.alt" > "$O/ab_mixed_stylpf.txt" 2>&1; cat "$O/ab_mixed_stylpf.txt"
DC_STAMP_PREC=mixed DC_STAMPS=1 DC_DISABLE_GRAPH=1 python tools/stage_stamps.py > "$O/stamps_mixed.txt" 2>&1; sed -n 2,17p "$O/stamps_mixed.txt"
