#!/bin/bash
# template of a GPU session (copy to tools/sessions/cur.sh, edit, run through tools/gpurun_retry.sh)
R="$(pwd)"; O="$R/gpurun_out/rNN_sK"; mkdir -p "$O"
# same-box A/B of compile-time variants B and C against the default library (ms per loop, per-kernel ms):
timeout 900 bash tools/ab.sh run B C > "$O/ab.txt" 2>&1; cat "$O/ab.txt"
# a subset of the GPU suite on a variant:
DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_B.alt" timeout 600 python -m pytest tests -m gpu -q -k "golden" 2>&1 | tail -3
