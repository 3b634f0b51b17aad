#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s25"; mkdir -p "$O"
DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_H4.alt" timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "config1_golden or forward_golden" > "$O/pytest_H4.txt" 2>&1; tail -2 "$O/pytest_H4.txt"
tools/ab.sh run H4 H2 > "$O/ab.txt" 2>&1; cat "$O/ab.txt"
