#!/bin/bash
# first GPU visits of the bf16x6w mode (conv_wino6.hip): parity beside f32w, a short bench line.  usage: gpu_wino6_1.sh TAG [bench]
TAG=${1:-x1}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
WINO_CHECK_MODES=${MODES:-bf16x6w} timeout 600 python tools/experiments/wino_check.py > gpurun_out/${TAG}_check.log 2>&1
echo "check rc=$?"
tail -45 gpurun_out/${TAG}_check.log
if [ -n "$2" ]; then
  for P in bf16x6w bf16x6; do
    timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-pmc --precision $P > gpurun_out/${TAG}_bench_$P.json 2> gpurun_out/${TAG}_bench_$P.err
    cut -c1-300 gpurun_out/${TAG}_bench_$P.json; tail -3 gpurun_out/${TAG}_bench_$P.err
  done
fi
