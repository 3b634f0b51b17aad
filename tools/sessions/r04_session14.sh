#!/bin/bash
R="$(pwd)"; O="$R/gpurun_out/r04_s14"; mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -k "rccl_world_size_one" > "$O/pytest_rccl.txt" 2>&1; tail -3 "$O/pytest_rccl.txt"
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_bs1" -- /usr/bin/python3 "$R/tools/time_small_batch.py" 1 > "$O/kt_bs1.log" 2>&1 )
python3 tools/pmc_summary.py "$O/kt_bs1" > "$O/summary_bs1.txt" 2>&1; head -30 "$O/summary_bs1.txt"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_bs4" -- /usr/bin/python3 "$R/tools/time_small_batch.py" 4 > "$O/kt_bs4.log" 2>&1 )
python3 tools/pmc_summary.py "$O/kt_bs4" > "$O/summary_bs4.txt" 2>&1; head -30 "$O/summary_bs4.txt"
DC_STAMP_BS=1 DC_STAMPS=1 DC_DISABLE_GRAPH=1 timeout 300 python tools/stage_stamps.py > "$O/stamps_bs1.txt" 2>&1; tail -40 "$O/stamps_bs1.txt"
rm -rf "$O"/kt_bs1/*/*.db "$O"/kt_bs4/*/*.db 2>/dev/null
find "$O" -name "*kernel_trace.csv" -size +8M -delete
