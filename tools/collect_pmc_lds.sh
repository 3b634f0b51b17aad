#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: the LDS / VMEM / SALU side of k_layer's stall attribution
# (round 3: the SQ pass of collect_profiles.sh shows ~37 % of SIMD cycles with no wave issuing; these passes say where the
# waves wait).  Two PMC passes of 8 SQ counters each, kernel trace only beside them (no other trace domain).
TAG="${1:-r03}"
R="$(pwd)"
OUT="$R/gpurun_out/profiles_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="/usr/bin/python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM \
    --kernel-trace --output-format csv -d "$OUT/pmc_lds" -- $BENCH > "$OUT/pmc_lds.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES \
    --kernel-trace --output-format csv -d "$OUT/pmc_issue" -- $BENCH > "$OUT/pmc_issue.log" 2>&1
cd "$R"
python3 tools/pmc_summary.py "$OUT/pmc_lds" > "$OUT/summary_pmc_lds.txt" 2>&1
python3 tools/pmc_summary.py "$OUT/pmc_issue" > "$OUT/summary_pmc_issue.txt" 2>&1
grep -i -E "SQ_(LDS|INSTS|WAIT|ACTIVE|INST_|VALU)" "$OUT/counters_available.txt" | head -200 > "$OUT/counters_sq.txt"
rm -rf "$OUT/pmc_lds" "$OUT/pmc_issue" "$OUT/counters_available.txt"
ls -la "$OUT"
