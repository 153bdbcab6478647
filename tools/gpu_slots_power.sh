#!/bin/bash
# Is the bf16x6 conv kernel power-limited ITSELF?  Run the bench with 32 / 24 / 16 / 8 persistent workgroups per XCD (= 256 / 192 /
# 128 / 64 active CUs): a kernel bound by its own pipeline loses throughput in proportion to the CUs taken away and keeps its
# clock; a power-limited one gets part of it back as clock.  Per setting: utt/s, conv TF/s, the PMC clock and matrix-pipe busy
# fraction of the same run (bench.py --pmc).   usage: gpu_slots_power.sh TAG      (table: profiles/r04_slots_power.txt)
# the kernel switches this script sets exist only in the EXPERIMENT build of the library (make -C misonet_amd/csrc exp)
export MISONET_LIB_PATH=${MISONET_LIB_PATH:-${GRAFT_REPO_ROOT:-/root/repo}/misonet_amd/libmisonet_hip_exp.so}
[ -f "$MISONET_LIB_PATH" ] || { echo "missing $MISONET_LIB_PATH: run make -C misonet_amd/csrc exp" >&2; exit 1; }
TAG=${1:-slots}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${TAG}_slots_power.txt
mkdir -p $R/gpurun_out
echo "# slots/XCD | active CUs | utt/s | conv TF/s (algorithmic) | PMC clock GHz | matrix pipe busy (all 256 CUs)" > $OUT
for S in 32 24 16 8; do
  L=$(MISONET_X6_SLOTS=$S timeout 600 python $R/bench.py --steps 5 --warmup 2 --no-alt --no-cpu-baseline --pmc 2>/dev/null | tail -1)
  V=$(echo "$L" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['value'], '|', r['achieved'], '|', r.get('clock_ghz_observed_pmc'), '|', r.get('mfma_busy_frac_pmc'))")
  echo "$S | $((S * 8)) | $V" >> $OUT
done
cat $OUT
