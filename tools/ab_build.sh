#!/bin/bash
# Fast variant build for tools/ab.sh run: recompiles only dc_kernels.hip and dc_api.hip with the extra flags and links them with the
# default build's dc_music.o / dc_layer16.o (run __graft_entry__.build() first).  usage: tools/ab_build.sh <V> [-DFLAG ...]
# (variants whose flags reach dc_layer16.hip or dc_music.hip need the full build: tools/ab.sh build <V> <flags>)
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
V="$1"; shift
B="$R/diffusion-conductor_amd/build/ab_$V"; mkdir -p "$B"
C="$R/diffusion-conductor_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-value"
hipcc $F "$@" -c "$C/dc_kernels.hip" -o "$B/dc_kernels.o" &
hipcc $F "$@" -c "$C/dc_api.hip" -o "$B/dc_api.o" &
wait
hipcc --offload-arch=gfx950 -shared -fPIC "$B/dc_kernels.o" "$B/dc_api.o" "$R/diffusion-conductor_amd/build/dc_music.o" "$R/diffusion-conductor_amd/build/dc_layer16.o" \
  -o "$R/diffusion-conductor_amd/libdc_ddim_$V.alt"
ls -la "$R/diffusion-conductor_amd/libdc_ddim_$V.alt"
