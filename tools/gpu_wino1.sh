#!/bin/bash
# first GPU visit of the f32w mode: parity beside f32, a short bench line per mode, per-layer kernel trace.  usage: gpu_wino1.sh TAG
TAG=${1:-w1}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python tools/experiments/wino_check.py > gpurun_out/${TAG}_check.log 2>&1
tail -40 gpurun_out/${TAG}_check.log
for P in f32w f32; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-pmc --precision $P > gpurun_out/${TAG}_bench_$P.json 2> gpurun_out/${TAG}_bench_$P.err
  cut -c1-400 gpurun_out/${TAG}_bench_$P.json; tail -2 gpurun_out/${TAG}_bench_$P.err
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-pmc --no-profile --precision f32w > /tmp/prof_$TAG.json 2> /tmp/prof_$TAG.err
DB=$(ls /tmp/prof_$TAG/*.db /tmp/prof_$TAG/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB $R/gpurun_out/${TAG}_kernel_stats.txt | head -8
python $R/tools/conv_layer_report.py $DB > $R/gpurun_out/${TAG}_conv_layers.txt
tail -1 $R/gpurun_out/${TAG}_conv_layers.txt
