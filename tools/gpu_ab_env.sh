#!/bin/bash
# A/B of two ENVIRONMENT settings of the same build on one box, arms alternated.  usage: gpu_ab_env.sh "A=1" "A=0" [rounds] [bench args]
# the kernel switches this script sets exist only in the EXPERIMENT build of the library (make -C misonet_amd/csrc exp)
export MISONET_LIB_PATH=${MISONET_LIB_PATH:-${GRAFT_REPO_ROOT:-/root/repo}/misonet_amd/libmisonet_hip_exp.so}
[ -f "$MISONET_LIB_PATH" ] || { echo "missing $MISONET_LIB_PATH: run make -C misonet_amd/csrc exp" >&2; exit 1; }
R=${GRAFT_REPO_ROOT:-/root/repo}
A="$1"; B="$2"; N=${3:-3}; shift 3
for i in $(seq $N); do
  for ARM in "$A" "$B"; do
    V=$(env $ARM timeout 300 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print(d['value'], 'utt/s  conv', r.get('achieved'), 'TF/s', (r.get('time_share') or {}))")
    echo "round $i  [$ARM]  $V"
  done
done
