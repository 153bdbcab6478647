#!/bin/bash
# one PMC pass with a custom counter list: gpu_pmc_custom.sh TAG PREC "COUNTER1 COUNTER2 ..."
TAG=$1; PREC=$2; P=$3
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_c
timeout 600 rocprofv3 --pmc $P --kernel-trace -d /tmp/pmc_c -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-alt --precision $PREC > /tmp/pmc_c.out 2> /tmp/pmc_c.err
DB=$(ls /tmp/pmc_c/*.db /tmp/pmc_c/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_pmc.py $DB $R/gpurun_out/${TAG}_pmc_custom.txt | head -16
tail -3 /tmp/pmc_c.err
