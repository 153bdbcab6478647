#!/bin/bash
# PMC passes over the f32w bench (conv3x3_wino_f32): matrix-pipe busy, clock, wait breakdown, LDS conflicts.  usage: gpu_wino_pmc.sh TAG [precision]
TAG=${1:-wp}
PREC=${2:-f32w}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-alt --no-pmc --precision $PREC"
i=0
for P in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 600 rocprofv3 --pmc $P --kernel-trace -d /tmp/pmc_$i -o pmc -- $CMD > /tmp/pmc_$i.out 2> /tmp/pmc_$i.err
  DB=$(ls /tmp/pmc_$i/*.db /tmp/pmc_$i/*/*.db 2>/dev/null | head -1)
  echo "== $P"
  python $R/tools/rocpd_pmc.py $DB $R/gpurun_out/${TAG}_pmc_$i.txt | head -14
  python $R/tools/rocpd_stats.py $DB $R/gpurun_out/${TAG}_pmc_${i}_durations.txt | head -4
done
