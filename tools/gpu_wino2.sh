#!/bin/bash
# f32w iteration: parity beside f32 (short), bench line, per-layer trace, ablations of the experiment build.  usage: gpu_wino2.sh TAG [DBGS]
TAG=${1:-w3}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python tools/experiments/wino_check.py > gpurun_out/${TAG}_check.log 2>&1
grep -E "f32w" gpurun_out/${TAG}_check.log | grep -v "tap" | tail -12
tail -3 gpurun_out/${TAG}_check.log | grep -i -E "error|Traceback" 
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-pmc --precision f32w > gpurun_out/${TAG}_bench_f32w.json 2> gpurun_out/${TAG}_bench_f32w.err
cut -c1-330 gpurun_out/${TAG}_bench_f32w.json; tail -2 gpurun_out/${TAG}_bench_f32w.err
bash tools/gpu_layers.sh ${TAG} f32w | tail -2
grep -E "enc1.db.c1 |enc1.db.c5|dec5.db.c5|enc0.db.c5|enc3.db.c3|dec6.db.c5" gpurun_out/${TAG}_conv_layers.txt
for D in ${2:-}; do
  echo "== DBG $D"
  bash $R/tools/gpu_layers.sh ${TAG}d$D f32w MISONET_LIB_PATH=$R/misonet_amd/libmisonet_hip_exp.so MISONET_WINO_DBG=$D 2>&1 | tail -1
  grep -E "enc1.db.c1 |enc1.db.c5|dec5.db.c5|enc0.db.c5|enc3.db.c3" $R/gpurun_out/${TAG}d${D}_conv_layers.txt
done
