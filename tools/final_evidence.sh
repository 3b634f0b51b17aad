#!/bin/bash
# GPU box: the round's evidence in one call - full gpu suite, rocprofv3 passes, stage stamps, the bench lines of every configuration.
# usage: bash tools/final_evidence.sh r04
TAG="${1:-r04}"
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/${TAG}_pytest_gpu.txt
bash tools/collect_profiles.sh $TAG > $O/${TAG}_collect.log 2>&1
DC_STAMPS=1 DC_DISABLE_GRAPH=1 python tools/stage_stamps.py > $O/${TAG}_stage_stamps.txt 2>&1
DC_STAMP_PREC=mixed DC_STAMPS=1 DC_DISABLE_GRAPH=1 python tools/stage_stamps.py > $O/${TAG}_stage_stamps_mixed.txt 2>&1
python bench.py 2> $O/${TAG}_bench_default.err | tail -1 > $O/${TAG}_bench_default.json
python bench.py --ddim 1000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${TAG}_bench_ddim1000.json
python bench.py --bs 128 --frames 900 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${TAG}_bench_t900.json
python bench.py --no-eff --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${TAG}_bench_noeff.json
python bench.py --precision mixed --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${TAG}_bench_mixed.json
timeout 600 python tools/time_small_batch.py 1 2 3 4 8 2>/dev/null > $O/${TAG}_small_batch.txt
cat $O/${TAG}_pytest_gpu.txt; for f in default ddim1000 t900 noeff mixed; do echo -n "$f: "; grep -o "ms_per_step\": [0-9.]*" $O/${TAG}_bench_$f.json; done
