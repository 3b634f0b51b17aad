#!/bin/bash
# GPU box: no_eff parity tests + the --no-eff bench line twice per library (same-box numbers for A/B of k_layer_full changes);
# arguments: variant names built with tools/ab.sh build <V> <flags>; NOTEST=1 skips the parity tests
cd "$(dirname "$0")/.."
[ -z "$NOTEST" ] && python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k no_eff 2>&1 | grep -v "^$" | tail -7
for v in "" "$@"; do
  if [ -n "$v" ]; then export DC_DDIM_LIB="$PWD/diffusion-conductor_amd/libdc_ddim_$v.alt"; else unset DC_DDIM_LIB; fi
  for i in 1 2; do echo -n "variant ${v:-default}: "; python bench.py --no-eff --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep -o "k_layer[a-z_]* [0-9.]*ms/[0-9]*\|ms_per_step\": [0-9.]*" | tr "\n" " "; echo; done
done
