#!/bin/bash
# round-2 measurement bundle for one precision mode: per-layer report + kernel stats + PMC (MFMA busy, traffic).
# usage: gpu_round2.sh TAG PREC
TAG=$1; PREC=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/gpu_layers.sh $TAG $PREC
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-alt --precision $PREC"
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  N=$(echo $P | cut -d' ' -f1)
  rm -rf /tmp/pmc_$N
  timeout 600 rocprofv3 --pmc $P --kernel-trace -d /tmp/pmc_$N -o pmc -- $CMD > /tmp/pmc_$N.out 2> /tmp/pmc_$N.err
  DB=$(ls /tmp/pmc_$N/*.db /tmp/pmc_$N/*/*.db 2>/dev/null | head -1)
  echo "== $P"
  python $R/tools/rocpd_pmc.py $DB $R/gpurun_out/${TAG}_pmc_$N.txt | head -12
  python $R/tools/rocpd_stats.py $DB $R/gpurun_out/${TAG}_pmc_${N}_durations.txt > /dev/null
done
