#!/bin/bash
# per-layer conv report under an env setting: gpu_layers.sh TAG PREC [ENV=VAL ...]
TAG=$1; PREC=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
env "$@" MISONET_BENCH_NOCHECK=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-profile --precision $PREC > /tmp/prof_$TAG.json 2> /tmp/prof_$TAG.err
DB=$(ls /tmp/prof_$TAG/*.db /tmp/prof_$TAG/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB $R/gpurun_out/${TAG}_kernel_stats.txt | head -12
python $R/tools/conv_layer_report.py $DB > $R/gpurun_out/${TAG}_conv_layers.txt
tail -1 $R/gpurun_out/${TAG}_conv_layers.txt
