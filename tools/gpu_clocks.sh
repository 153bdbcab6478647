#!/bin/bash
# sample sclk / power while bench.py runs: gpu_clocks.sh PREC [ENV=VAL ...]
PREC=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > /tmp/clk.log &
CP=$!
env "$@" MISONET_BENCH_NOCHECK=1 python bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-alt --no-profile --precision $PREC 2>/dev/null | cut -c60-140
wait $CP
sort /tmp/clk.log | uniq -c | sort -rn | head -8
