#!/bin/bash
# bf16x6w (conv_wino6.hip): parity check, bench beside bf16x6, per-layer kernel trace.  usage: gpu_wino6_2.sh TAG
TAG=${1:-x2}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
WINO_CHECK_MODES=bf16x6w timeout 600 python tools/experiments/wino_check.py > gpurun_out/${TAG}_check.log 2>&1
echo "check rc=$?"; grep -E "T=32 vs|B=|float64" gpurun_out/${TAG}_check.log | cut -c1-150
for P in bf16x6w bf16x6; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-pmc --precision $P > gpurun_out/${TAG}_bench_$P.json 2> gpurun_out/${TAG}_bench_$P.err
  cut -c1-220 gpurun_out/${TAG}_bench_$P.json; tail -2 gpurun_out/${TAG}_bench_$P.err
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-pmc --no-profile --precision bf16x6w > /tmp/prof_$TAG.json 2> /tmp/prof_$TAG.err
DB=$(ls /tmp/prof_$TAG/*.db /tmp/prof_$TAG/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB $R/gpurun_out/${TAG}_kernel_stats.txt | head -6
python $R/tools/conv_layer_report.py $DB > $R/gpurun_out/${TAG}_conv_layers.txt
tail -1 $R/gpurun_out/${TAG}_conv_layers.txt
