#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s26"; mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "rides_in_the_film or layer16_small" > "$O/pytest.txt" 2>&1; tail -3 "$O/pytest.txt"
for i in 1 2; do
for v in "" 0 2; do
  if [ -n "$v" ]; then export DC_FUSE_EXTRA=$v; else unset DC_FUSE_EXTRA; fi
  echo "== DC_FUSE_EXTRA=${v:-default}"; timeout 600 python tools/time_small_batch.py 3 4 5 6 8 12 16 2>&1 | grep "bs=" | sed 's/(\[[^]]*\])//g'
done; done > "$O/small_extra_rule.txt" 2>&1; cat "$O/small_extra_rule.txt"
