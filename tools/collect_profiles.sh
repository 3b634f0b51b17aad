#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: collects the rocprofv3 evidence for bench.py's numbers into
# gpurun_out/profiles_<tag>/ ; copy the summaries you want judged into profiles/ afterwards (tools/make_tables.py reads them there).
#   1. kernel trace + stats of the default bench command
#   2. PMC pass A: SQ occupancy / wait / MFMA-busy counters
#   3. PMC pass B: FETCH_SIZE ; pass C: WRITE_SIZE  (TCC counters cannot share a pass, MI355X_MICROARCH.md)
#   4. PMC passes D, E: LDS / VMEM / SALU side of the stall attribution (round 3)
# Every pass is kernel trace + PMC only (no other trace domain), the program directly after `--`.
TAG="${1:-r01}"
R="$(pwd)"
OUT="$R/gpurun_out/profiles_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="/usr/bin/python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $BENCH > "$OUT/kt.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES \
    --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM \
    --kernel-trace --output-format csv -d "$OUT/pmc_lds" -- $BENCH > "$OUT/pmc_lds.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES \
    --kernel-trace --output-format csv -d "$OUT/pmc_issue" -- $BENCH > "$OUT/pmc_issue.log" 2>&1
cd "$R"
for p in kt pmc_sq pmc_fetch pmc_write pmc_lds pmc_issue; do
  n=$p; [ $p = kt ] && n=kernel_trace
  python3 tools/pmc_summary.py "$OUT/$p" > "$OUT/summary_$n.txt" 2>&1
done
cp "$OUT"/kt/*/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
grep -h '^{' "$OUT/kt.log" | tail -1 > "$OUT/bench_line_under_profiler.json"
# keep only the summaries (the raw per-dispatch CSVs are tens of MB)
rm -rf "$OUT/kt" "$OUT/pmc_sq" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_lds" "$OUT/pmc_issue"
ls -la "$OUT"
