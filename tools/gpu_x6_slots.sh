#!/bin/bash
# DMA-only (MISONET_WS_DEBUG=1: consumers skip the MFMAs) and full timelines with 1, 4, 16, 32 workgroups per XCD:
# is the LDS-DMA rate of a CU its own limit or a share of the L2 / fabric?
# the kernel switches this script sets exist only in the EXPERIMENT build of the library (make -C misonet_amd/csrc exp)
export MISONET_LIB_PATH=${MISONET_LIB_PATH:-${GRAFT_REPO_ROOT:-/root/repo}/misonet_amd/libmisonet_hip_exp.so}
[ -f "$MISONET_LIB_PATH" ] || { echo "missing $MISONET_LIB_PATH: run make -C misonet_amd/csrc exp" >&2; exit 1; }
R=${GRAFT_REPO_ROOT:-/root/repo}
for D in 1 0; do for S in 1 4 16 32; do
  echo "== MISONET_WS_DEBUG=$D MISONET_X6_SLOTS=$S"
  MISONET_X6_SLOTS=$S MISONET_WS_DEBUG=$D MISONET_TIMELINE=96 timeout 600 python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-profile 2>&1 | grep -E "timeline-x6. Cin" | head -1 | cut -c1-300
done; done
