#!/bin/bash
# sustained clock / issued TF/s / socket power of synthetic loads with the composition of the bf16x6 conv
# kernel (tools/micro/power_mix.hip), next to the real bench.  usage: gpu_power_table.sh TAG
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${TAG}_power_table.txt
mkdir -p $R/gpurun_out
BIN=$R/tools/micro/bin/power_mix
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o $BIN $R/tools/micro/power_mix.hip
sample() {   # median sclk / power while PID $1 lives (skipping its first 1.5 s)
  sleep 1.5
  local P=() C=()
  while kill -0 $1 2>/dev/null; do
    L=$(rocm-smi --showclocks --showpower 2>/dev/null)
    c=$(echo "$L" | grep -m1 "sclk" | sed 's/.*(\([0-9]*\)Mhz).*/\1/')
    p=$(echo "$L" | grep -m1 -E "Socket Graphics|Power \(W\)" | sed 's/.*: *\([0-9.]*\).*/\1/')
    [ -n "$c" ] && C+=($c); [ -n "$p" ] && P+=($p)
    sleep 0.2
  done
  med() { printf '%s\n' "$@" | sort -n | awk '{a[NR]=$1} END{print (NR?a[int((NR+1)/2)]:"-")}'; }
  echo "sclk_smi $(med "${C[@]}") MHz  power $(med "${P[@]}") W  (${#P[@]} samples)"
}
{
echo "# synthetic loads: one 512-thread workgroup per CU, 4 MFMA waves (v_mfma_f32_32x32x16_bf16, random operands) + 4 streaming waves"
echo "# columns: load | in-kernel duty, clock, issued TF/s, HBM TB/s | rocm-smi sclk, socket power"
for V in "6 0 0 0" "6 0 0 0 1" "6 2 0 0" "6 5 0 0" "6 8 0 0" "6 12 0 0" "6 0 24 0" "6 0 24 16" "6 5 24 16" "6 8 24 16" "6 8 24 32" "6 12 24 32"; do
  $BIN $V > /tmp/pm.out 2>&1 &
  PID=$!
  S=$(sample $PID)
  wait $PID
  echo "$(cat /tmp/pm.out) | $S"
done
echo "# the real thing: bench.py --precision bf16x6, 150 steps (conv kernels = 96 % of the step)"
cd $R
python bench.py --steps 150 --warmup 2 --no-cpu-baseline --no-alt --no-profile --precision bf16x6 > /tmp/pt_bench.json 2>/dev/null &
PID=$!
sleep 15
S=$(sample $PID)
wait $PID
python - <<PY
import json
d=json.loads(open('/tmp/pt_bench.json').read().strip().splitlines()[-1])
print("bench bf16x6: %.1f utt/s, %.2f ms/step | $S" % (d['value'], d['ms_per_step']))
PY
echo "# f32 mode for comparison"
python bench.py --steps 100 --warmup 2 --no-cpu-baseline --no-alt --no-profile --precision f32 > /tmp/pt_bench.json 2>/dev/null &
PID=$!
sleep 15
S=$(sample $PID)
wait $PID
python - <<PY
import json
d=json.loads(open('/tmp/pt_bench.json').read().strip().splitlines()[-1])
print("bench f32: %.1f utt/s, %.2f ms/step | $S" % (d['value'], d['ms_per_step']))
PY
} 2>&1 | tee $OUT
