#!/bin/bash
TAG="${1:-s3}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "persistent or graph or bs32 or config1 or captured" > gpurun_out/t_pers_$TAG.log 2>&1
echo "rc=$?" >> gpurun_out/t_pers_$TAG.log
grep -E "passed|failed|rel-L2|Error" gpurun_out/t_pers_$TAG.log | tail -15
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_pers_$TAG.json 2> gpurun_out/bench_pers_$TAG.err
DC_NO_PERSIST=1 timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_nopers_$TAG.json 2> gpurun_out/bench_nopers_$TAG.err
python - <<PY
import json
for n in ("pers","nopers"):
    try:
        d=json.loads(open("gpurun_out/bench_%s_$TAG.json"%n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"].get("time_share_by_kernel"))
    except Exception as e:
        print(n, "FAILED", e); print(open("gpurun_out/bench_%s_$TAG.err"%n).read()[-1500:])
PY
done
