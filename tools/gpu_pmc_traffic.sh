#!/bin/bash
# HBM traffic passes only (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only).  usage: gpu_pmc_traffic.sh TAG PREC
TAG=${1:-x}; PREC=${2:-bf16x3}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-alt --precision $PREC"
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_$P
  timeout 600 rocprofv3 --pmc $P --kernel-trace -d /tmp/pmc_$P -o pmc -- $CMD > /tmp/pmc_$P.out 2> /tmp/pmc_$P.err
  DB=$(ls /tmp/pmc_$P/*.db /tmp/pmc_$P/*/*.db 2>/dev/null | head -1)
  echo "== $P"
  python $R/tools/rocpd_pmc.py $DB $R/gpurun_out/${TAG}_pmc_$P.txt | head -12
done
