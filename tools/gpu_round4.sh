#!/bin/bash
# round-4 measurement bundle (bf16x6 headline): full bench line, kernel stats + per-layer report, PMC passes, batch 32.
# usage: gpu_round3.sh TAG
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cut -c1-600 gpurun_out/${TAG}_bench.json
bash tools/gpu_layers.sh $TAG bf16x6
bash tools/gpu_layers.sh ${TAG}_f32 f32
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-alt --precision bf16x6"
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  N=$(echo $P | cut -d' ' -f1)
  rm -rf /tmp/pmc_$N
  timeout 600 rocprofv3 --pmc $P --kernel-trace -d /tmp/pmc_$N -o pmc -- $CMD > /tmp/pmc_$N.out 2> /tmp/pmc_$N.err
  DB=$(ls /tmp/pmc_$N/*.db /tmp/pmc_$N/*/*.db 2>/dev/null | head -1)
  echo "== $P"
  python $R/tools/rocpd_pmc.py $DB $R/gpurun_out/${TAG}_pmc_$N.txt | head -12
  python $R/tools/rocpd_stats.py $DB $R/gpurun_out/${TAG}_pmc_${N}_durations.txt > /dev/null
done
