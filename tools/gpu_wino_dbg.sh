#!/bin/bash
# where the time of conv3x3_wino_f32 goes: per-layer reports of the experiment build with parts switched off
# (MISONET_WINO_DBG: 1 = no staging side work, 2 = no epilogue, 3 = neither).  usage: gpu_wino_dbg.sh TAG
TAG=${1:-wd}
R=${GRAFT_REPO_ROOT:-/root/repo}
for D in ${DBGS:-0 1 2 3}; do
  echo "== DBG $D"
  bash $R/tools/gpu_layers.sh ${TAG}$D f32w MISONET_LIB_PATH=$R/misonet_amd/libmisonet_hip_exp.so MISONET_WINO_DBG=$D 2>&1 | tail -4
  grep -E "enc1.db.c1 |enc1.db.c5|dec5.db.c5|enc0.db.c5|enc3.db.c3" $R/gpurun_out/${TAG}${D}_conv_layers.txt
done
