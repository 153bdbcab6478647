#!/bin/bash
# Timeline + throughput of the bf16x6 conv kernel with parts of it switched off (MISONET_WS_DEBUG bits: 1 consumers skip
# the MFMAs, 4 skip the epilogue, 64 producers skip the DMA).  Results are wrong with any bit set; timing only.
# the kernel switches this script sets exist only in the EXPERIMENT build of the library (make -C misonet_amd/csrc exp)
export MISONET_LIB_PATH=${MISONET_LIB_PATH:-${GRAFT_REPO_ROOT:-/root/repo}/misonet_amd/libmisonet_hip_exp.so}
[ -f "$MISONET_LIB_PATH" ] || { echo "missing $MISONET_LIB_PATH: run make -C misonet_amd/csrc exp" >&2; exit 1; }
R=${GRAFT_REPO_ROOT:-/root/repo}
for D in ${DBGS:-0 64 1 65 4}; do
  echo "== MISONET_WS_DEBUG=$D $*"
  env "$@" MISONET_WS_DEBUG=$D MISONET_TIMELINE=96 timeout 300 python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-profile 2>&1 | grep -E "timeline|\"value\"" | head -3 | cut -c1-420
done
