#!/bin/bash
# GPU box: rocprofv3 kernel traces of the secondary configurations (program directly after --): no_eff at bs=32, one clip per call
R="$(pwd)"; OUT="$R/gpurun_out/kt_extra"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/noeff" -- /usr/bin/python3 $R/bench.py --no-eff --steps 1 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/noeff.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bs1" -- /usr/bin/python3 $R/bench.py --bs 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/bs1.log" 2>&1
cd "$R"
python3 tools/pmc_summary.py "$OUT/noeff" > "$R/gpurun_out/r06_noeff_summary_kernel_trace.txt" 2>&1
python3 tools/pmc_summary.py "$OUT/bs1" > "$R/gpurun_out/r06_summary_kernel_trace_bs1.txt" 2>&1
grep -h '^{' "$OUT/noeff.log" | tail -1 > "$R/gpurun_out/r06_noeff_bench_line_under_profiler.json"
grep -h '^{' "$OUT/bs1.log" | tail -1 > "$R/gpurun_out/r06_bs1_bench_line_under_profiler.json"
rm -rf "$OUT"
head -12 "$R/gpurun_out/r06_noeff_summary_kernel_trace.txt" | cut -c1-160; head -12 "$R/gpurun_out/r06_summary_kernel_trace_bs1.txt" | cut -c1-160
