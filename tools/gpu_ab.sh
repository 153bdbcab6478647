#!/bin/bash
# A/B of an environment toggle: usage gpu_ab.sh TAG "ENV=0" "ENV=1" [prec ...]
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
