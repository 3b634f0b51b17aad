#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s10"; mkdir -p "$O"
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -E "ifetch|icache|SQC_|INST_CACHE|IFETCH" | head -60 > "$O/counters.txt"; cat "$O/counters.txt" | cut -c1-200
BENCH="/usr/bin/python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE \
    --kernel-trace --output-format csv -d "$O/pmc_ifetch" -- $BENCH > "$O/pmc_ifetch.log" 2>&1
cd "$R"; python3 tools/pmc_summary.py "$O/pmc_ifetch" > "$O/summary_pmc_ifetch.txt" 2>&1; head -40 "$O/summary_pmc_ifetch.txt" | cut -c1-220; tail -5 "$O/pmc_ifetch.log"
rm -rf "$O/pmc_ifetch"
