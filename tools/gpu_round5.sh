#!/bin/bash
# round-5 measurement bundle: the default bench line (timed), kernel stats + per-layer reports for bf16x6 / f32w / f32.
# usage: gpu_round5.sh TAG
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
T0=$(date +%s.%N); python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench wall: $(echo "$(date +%s.%N) - $T0" | bc) s" | tee gpurun_out/${TAG}_bench.time; cut -c1-700 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
bash tools/gpu_layers.sh $TAG bf16x6 | tail -2
bash tools/gpu_layers.sh ${TAG}_f32w f32w | tail -2
bash tools/gpu_layers.sh ${TAG}_f32 f32 | tail -2
