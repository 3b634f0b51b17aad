#!/bin/bash
# A/B two builds of libdc_ddim.so on the SAME GPU box (the pool's boxes differ by +-5 %, so numbers from two gpurun calls
# do not compare).  Usage (here): git stash / edit ... ; tools/ab.sh build B   # builds the working tree as variant B
#                  tools/ab.sh build C -DSOME_MACRO=1                           # extra hipcc flags after the variant name
#                  (on the box):  tools/ab.sh run [B C ...]                     # alternates default lib and the variants (default: B)
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
case "$1" in
  build)
    cd "$R/diffusion-conductor_amd/csrc"
    V="$2"; shift 2
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -shared -Wno-unused-value "$@" dc_kernels.hip dc_api.hip dc_music.hip dc_layer16.hip -o "../libdc_ddim_$V.alt"
    ls -la "../libdc_ddim_$V.alt" ;;
  run)
    cd "$R"
    shift
    VARIANTS="${*:-B}"
    for i in 1 2; do
      for v in "" $VARIANTS; do
        if [ -n "$v" ]; then export DC_DDIM_LIB="$R/diffusion-conductor_amd/libdc_ddim_$v.alt"; else unset DC_DDIM_LIB; fi
        echo -n "variant ${v:-default}: "
        python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep -o "k_film_gemm [0-9.]*ms\|k_embed_front [0-9.]*ms\|k_layer [0-9.]*ms\|ms_per_step\": [0-9.]*" | tr "\n" " "; echo
      done
    done ;;
esac
