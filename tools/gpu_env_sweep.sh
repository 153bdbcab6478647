#!/bin/bash
# usage: gpu_env_sweep.sh PREC "ENV1" "ENV2" ...   (each ENV is a space-free VAR=VAL, or "none")
# the kernel switches this script sets exist only in the EXPERIMENT build of the library (make -C misonet_amd/csrc exp)
export MISONET_LIB_PATH=${MISONET_LIB_PATH:-${GRAFT_REPO_ROOT:-/root/repo}/misonet_amd/libmisonet_hip_exp.so}
[ -f "$MISONET_LIB_PATH" ] || { echo "missing $MISONET_LIB_PATH: run make -C misonet_amd/csrc exp" >&2; exit 1; }
PREC=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for E in "$@"; do
  echo "== $PREC $E"
  if [ "$E" = "none" ]; then EE="MISONET_NOP=1"; else EE="$E"; fi
  env $EE MISONET_BENCH_NOCHECK=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --precision $PREC 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], 'utt/s', d['ms_per_step'], 'ms/step', r['time_share'])"
done
