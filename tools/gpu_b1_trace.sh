#!/bin/bash
# kernel trace of the single-utterance pass (B = 1: the reference harness' own batch size).  usage: gpu_b1_trace.sh TAG [ENV=VAL ...]
TAG=${1:-b1}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
env "$@" MISONET_BENCH_NOCHECK=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $R/bench.py --batch 1 --steps 10 --warmup 2 --no-cpu-baseline --no-alt --no-profile > /tmp/prof_$TAG.json 2> /tmp/prof_$TAG.err
cut -c1-160 /tmp/prof_$TAG.json
DB=$(ls /tmp/prof_$TAG/*.db /tmp/prof_$TAG/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB $R/gpurun_out/${TAG}_kernel_stats.txt | head -25
python $R/tools/conv_layer_report.py $DB 1 > $R/gpurun_out/${TAG}_conv_layers.txt
cat $R/gpurun_out/${TAG}_conv_layers.txt
