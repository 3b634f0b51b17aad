#!/bin/bash
# round 4, GPU session 1: same-box A/B of diagnostic builds + stage stamps (what bounds k_layer's launch period)
R="$(pwd)"; O="$R/gpurun_out/r04_s1"; mkdir -p "$O"
tools/ab.sh run S NB H N > "$O/ab.txt" 2>&1
for v in "" NB H N; do
  if [ -n "$v" ]; then export DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_$v.alt"; else unset DC_DDIM_LIB; fi
  DC_STAMPS=1 DC_DISABLE_GRAPH=1 python tools/stage_stamps.py > "$O/stamps_${v:-default}.txt" 2>&1
done
unset DC_DDIM_LIB
python -m pytest tests/test_gpu_parity.py -k "interior" -x -q -s > "$O/pytest_interior.txt" 2>&1
tail -5 "$O/pytest_interior.txt"; cat "$O/ab.txt"
