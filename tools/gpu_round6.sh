#!/bin/bash
# round-6 measurement bundle: the default bench line (timed), kernel stats + per-layer reports for bf16x6 / f32w / f32, smoke, GPU suite.
# usage: gpu_round6.sh TAG
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
T0=$(date +%s.%N); python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python -c "import time,sys; print('bench wall: %.1f s' % (time.time() - float(sys.argv[1])))" $T0 | tee gpurun_out/${TAG}_bench.time; cut -c1-500 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
python bench.py --precision f32 --no-alt --pmc --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_f32.json 2> gpurun_out/${TAG}_bench_f32.err; cut -c1-200 gpurun_out/${TAG}_bench_f32.json
bash tools/gpu_layers.sh $TAG bf16x6 | tail -2
bash tools/gpu_layers.sh ${TAG}_f32w f32w | tail -2
bash tools/gpu_layers.sh ${TAG}_f32 f32 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/${TAG}_gpu_tests.log 2>&1; tail -14 gpurun_out/${TAG}_gpu_tests.log
