#!/bin/bash
# bf16x6w quick loop: parity (T = 32 golden + ragged shapes) and a 5-step bench line beside bf16x6 on the SAME box.  usage: gpu_wino6_3.sh TAG
TAG=${1:-x5}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out; cd $R
WINO_CHECK_MODES=bf16x6w timeout 600 python tools/experiments/wino_check.py > gpurun_out/${TAG}_check.log 2>&1
echo "check rc=$?"; grep -E "T=32 vs|B=|float64" gpurun_out/${TAG}_check.log | cut -c1-150
for P in bf16x6w bf16x6 bf16x6w; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-pmc --precision $P > gpurun_out/${TAG}_bench_$P.json 2> gpurun_out/${TAG}_bench_$P.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/${TAG}_bench_$P.json") if l.startswith("{")][-1]); print("$P", d["value"], d["roofline"]["time_share"])
PY
done
