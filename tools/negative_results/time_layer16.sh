cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 32 16; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl$v -- python3 $R/tools/negative_results/time_layer16.py $v > /dev/null 2>&1
  echo "== layer$v"; python3 $R/tools/negative_results/time_layer16.py --report $R/gpurun_out/tl$v
done
