#!/bin/bash
# A/B of an environment toggle: usage gpu_ab.sh TAG "ENV=0" "ENV=1" [prec ...]
# the kernel switches this script sets exist only in the EXPERIMENT build of the library (make -C misonet_amd/csrc exp)
export MISONET_LIB_PATH=${MISONET_LIB_PATH:-${GRAFT_REPO_ROOT:-/root/repo}/misonet_amd/libmisonet_hip_exp.so}
[ -f "$MISONET_LIB_PATH" ] || { echo "missing $MISONET_LIB_PATH: run make -C misonet_amd/csrc exp" >&2; exit 1; }
TAG=$1; A=$2; B=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out; cd $R
for PREC in "$@"; do
  for E in "$A" "$B"; do
    echo "== $PREC $E"
    env $E python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --precision $PREC 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], 'utt/s', d['ms_per_step'], 'ms/step conv', r['time_share'], 'achieved', r['achieved'])"
  done
done
