#!/bin/bash
# L2 (TCC) counters for the conv kernels.  usage: gpu_pmc2.sh TAG [f32|bf16x3]
TAG=${1:-x}
PREC=${2:-bf16x3}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-alt --precision $PREC"
for P in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  N=$(echo $P | cut -d' ' -f1)
  rm -rf /tmp/pmc_$N
  timeout 600 rocprofv3 --pmc $P --kernel-trace -d /tmp/pmc_$N -o pmc -- $CMD > /tmp/pmc_$N.out 2> /tmp/pmc_$N.err
  DB=$(ls /tmp/pmc_$N/*.db /tmp/pmc_$N/*/*.db 2>/dev/null | head -1)
  echo "== $P"
  python $R/tools/rocpd_pmc.py $DB $R/gpurun_out/${TAG}_pmc_$N.txt | grep -E "kernel|conv3x3" | head -12
  tail -1 /tmp/pmc_$N.err
done
