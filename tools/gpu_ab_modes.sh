#!/bin/bash
# same-box A/B of arithmetic MODES of the product library: alternating short bench runs.  usage: gpu_ab_modes.sh ROUNDS MODE1 MODE2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
for i in $(seq $N); do
  for P in "$@"; do
    V=$(MISONET_BENCH_NOCHECK=1 timeout 300 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-pmc --no-profile --precision $P 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], 'utt/s', d['ms_per_step'], 'ms')")
    echo "round $i [$P] $V"
  done
done
