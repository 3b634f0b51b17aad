#!/bin/bash
# GPU box: SQ counters of the no_eff layer kernel (two --pmc passes, kernel trace only beside them, program directly after --)
R="$(pwd)"; OUT="$R/gpurun_out/pmc_noeff"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
BENCH="/usr/bin/python3 $R/bench.py --no-eff --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES \
    --kernel-trace --output-format csv -d "$OUT/sq" -- $BENCH > "$OUT/sq.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES \
    --kernel-trace --output-format csv -d "$OUT/issue" -- $BENCH > "$OUT/issue.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d "$OUT/mix" -- $BENCH > "$OUT/mix.log" 2>&1
cd "$R"
for p in sq issue mix; do python3 tools/pmc_summary.py "$OUT/$p" > "$R/gpurun_out/r06_noeff_summary_pmc_$p.txt" 2>&1; done
rm -rf "$OUT"
grep -h "k_layer_full\|Counter_Name" $R/gpurun_out/r06_noeff_summary_pmc_*.txt | cut -c1-330
