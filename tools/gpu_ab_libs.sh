#!/bin/bash
# same-box A/B of several builds of the library (boxes differ by 3-8 %): alternating short bench runs.
# usage: gpu_ab_libs.sh PRECISION ROUNDS lib1.so lib2.so ...   (paths relative to the repo root)
R=${GRAFT_REPO_ROOT:-/root/repo}
PREC=$1; N=$2; shift 2
for i in $(seq $N); do
  for L in "$@"; do
    V=$(MISONET_BENCH_NOCHECK=1 MISONET_LIB_PATH=$R/$L timeout 300 python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-pmc --no-profile --precision $PREC 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], 'utt/s', d['ms_per_step'], 'ms')")
    echo "round $i [$L] $V"
  done
done
