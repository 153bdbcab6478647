#!/bin/bash
# same-box A/B of environment settings with ONE library (the experiment build): alternating short bench runs.
# usage: gpu_ab_envs.sh PRECISION ROUNDS LIB "ENV1" "ENV2" ...     (an ENV is a space-separated list of VAR=value, or "-")
R=${GRAFT_REPO_ROOT:-/root/repo}
PREC=$1; N=$2; LIB=$3; shift 3
for i in $(seq $N); do
  for E in "$@"; do
    EV=""; [ "$E" != "-" ] && EV="$E"
    V=$(env $EV MISONET_BENCH_NOCHECK=1 MISONET_LIB_PATH=$R/$LIB timeout 300 python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-pmc --no-profile --precision $PREC 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], 'utt/s', d['ms_per_step'], 'ms')")
    echo "round $i [$E] $V"
  done
done
