#!/bin/bash
# bf16x6w ablations (experiment build): MISONET_WINO6_DBG variants, conv ms per step.  usage: gpu_wino6_dbg.sh TAG "0 1 2 ..."
TAG=${1:-xd}
R=${GRAFT_REPO_ROOT:-/root/repo}
export MISONET_LIB_PATH=$R/misonet_amd/libmisonet_hip_exp.so
[ -f "$MISONET_LIB_PATH" ] || { echo "missing $MISONET_LIB_PATH: run make -C misonet_amd/csrc exp" >&2; exit 1; }
mkdir -p $R/gpurun_out; cd $R
for D in ${2:-0 1 2 4 8 12}; do
  MISONET_WINO6_DBG=$D MISONET_BENCH_NOCHECK=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-pmc --precision bf16x6w > gpurun_out/${TAG}_dbg$D.json 2> gpurun_out/${TAG}_dbg$D.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/${TAG}_dbg$D.json") if l.startswith("{")][-1])
    print("DBG=$D", d["value"], "utt/s", d["ms_per_step"], "ms/step", d["roofline"].get("time_share"))
except Exception as e:
    print("DBG=$D failed", e); print(open("gpurun_out/${TAG}_dbg$D.err").read()[-600:])
PY
done
