#!/bin/bash
# PMC passes (separate runs, --kernel-trace only, as the guide prescribes).  usage: gpu_pmc.sh TAG
TAG=${1:-x}
PREC=${2:-f32}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --precision $PREC"
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES"; do
  N=$(echo $P | cut -d' ' -f1)
  rm -rf /tmp/pmc_$N
  timeout 600 rocprofv3 --pmc $P --kernel-trace -d /tmp/pmc_$N -o pmc -- $CMD > /tmp/pmc_$N.out 2> /tmp/pmc_$N.err
  DB=$(ls /tmp/pmc_$N/*.db /tmp/pmc_$N/*/*.db 2>/dev/null | head -1)
  echo "== $P ($DB)"
  python $R/tools/rocpd_pmc.py $DB $R/gpurun_out/${TAG}_pmc_$N.txt | head -14
  tail -2 /tmp/pmc_$N.err
done
