#!/bin/bash
# One GPU-box cycle: parity tests, bench line, rocprofv3 kernel trace summarised to text.
# usage: gpu_cycle.sh TAG [notest|test] [f32|bf16x3]
TAG=${1:-x}
PREC=${3:-f32}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
if [ "$2" != "notest" ]; then
  python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/${TAG}_tests.log 2>&1
  tail -3 gpurun_out/${TAG}_tests.log
fi
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --precision $PREC > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cat gpurun_out/${TAG}_bench.json
tail -3 gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --precision $PREC > /tmp/prof_$TAG.json 2> /tmp/prof_$TAG.err
DB=$(ls /tmp/prof_$TAG/*.db /tmp/prof_$TAG/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB $R/gpurun_out/${TAG}_kernel_stats.txt | head -8
python $R/tools/conv_layer_report.py $DB > $R/gpurun_out/${TAG}_conv_layers.txt
tail -1 $R/gpurun_out/${TAG}_conv_layers.txt
