#!/bin/bash
# sample sclk / power while bench.py runs a long timed region: gpu_clocks.sh PREC [steps]
PREC=${1:-bf16x3}; STEPS=${2:-150}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-alt --no-profile --precision $PREC > /tmp/clk_bench.json 2>/dev/null &
BP=$!
sleep 12
for i in $(seq 1 30); do
  kill -0 $BP 2>/dev/null || break
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics" | sed 's/.*sclk clock level: [^(]*(\([0-9]*Mhz\)).*/sclk \1/; s/.*Power (W): \([0-9.]*\).*/power \1 W/' | tr '\n' ' '; echo
  sleep 0.3
done
wait $BP
cut -c60-130 /tmp/clk_bench.json
