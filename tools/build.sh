#!/bin/bash
# Build libdc_ddim.so for gfx950 and print register/spill usage of the kernels matching $1 (default: k_layer).
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R/diffusion-conductor_amd/csrc"
PAT="${1:-k_layerIDF16_Lb0ELb0ELb0ELb1ELb0}"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -shared -Wno-unused-value dc_kernels.hip dc_api.hip dc_music.hip dc_layer16.hip -o ../libdc_ddim.so \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs:|ScratchSize|VGPRs Spill|Occupancy" \
  | grep -A5 -E "error|Name: _Z[0-9]+($PAT)" | grep -E "error|Name|VGPRs:|AGPRs|Scratch|VGPRs Spill" | sed 's/.*remark: *//' | cut -c1-100
