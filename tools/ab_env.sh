#!/bin/bash
# same-box A/B of run-time switches: tools/ab_env.sh "" DC_NO_ALIGN=1 DC_NO_PAD=1   (each argument: an environment assignment or "")
# BENCH_ARGS="--precision mixed" adds bench.py arguments
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
for i in 1 2 3; do
  for v in "$@"; do
    echo -n "${v:-default}: "
    env $v python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras $BENCH_ARGS 2>&1 | grep -o "k_film_gemm [0-9.]*ms\|k_embed_front [0-9.]*ms\|k_attn_combine [0-9.]*ms\|k_layer [0-9.]*ms\|ms_per_step\": [0-9.]*" | tr "\n" " "; echo
  done
done
