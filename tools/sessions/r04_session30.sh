#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s30"; mkdir -p "$O"
python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.txt" 2>&1; tail -2 "$O/smoke.txt"
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "config1_golden or forward_golden or rides_in_the_film or layer16_small or bs32_full" > "$O/pytest.txt" 2>&1; tail -2 "$O/pytest.txt"
python bench.py --no-cpu-baseline 2> "$O/bench.err" | tail -1 > "$O/bench.json"; grep -o "ms_per_step\": [0-9.]*" "$O/bench.json" | head -1; grep "end to end\|bs=1" "$O/bench.err"
