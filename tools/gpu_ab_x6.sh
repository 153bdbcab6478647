#!/bin/bash
# A/B of bf16x6 kernel variants ON ONE BOX (boxes differ by >10 % in sustained clocks): alternate the arms.
# usage: gpu_ab_x6.sh "ENV_A" "ENV_B" [rounds]
# the kernel switches this script sets exist only in the EXPERIMENT build of the library (make -C misonet_amd/csrc exp)
export MISONET_LIB_PATH=${MISONET_LIB_PATH:-${GRAFT_REPO_ROOT:-/root/repo}/misonet_amd/libmisonet_hip_exp.so}
[ -f "$MISONET_LIB_PATH" ] || { echo "missing $MISONET_LIB_PATH: run make -C misonet_amd/csrc exp" >&2; exit 1; }
R=${GRAFT_REPO_ROOT:-/root/repo}
A="$1"; B="$2"; N=${3:-3}
for i in $(seq $N); do
  for ARM in "$A" "$B"; do
    V=$(env $ARM timeout 300 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-profile 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
    echo "round $i  [$ARM]  $V utt/s"
  done
done
