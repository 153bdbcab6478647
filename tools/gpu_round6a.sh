#!/bin/bash
# round-6 first check: parity of the changed kernels, per-layer report of f32w, same-box A/B against the round-5 library.
# usage: gpu_round6a.sh TAG
TAG=${1:-r06a}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_hygiene.py -m gpu -q -p no:cacheprovider --maxfail=15 -k "not bench" > gpurun_out/${TAG}_tests.log 2>&1; tail -15 gpurun_out/${TAG}_tests.log
bash tools/gpu_layers.sh ${TAG}_f32w f32w | tail -3
bash tools/gpu_layers.sh ${TAG}_f32 f32 | tail -2
bash tools/gpu_ab_libs.sh f32w 2 misonet_amd/libmisonet_hip_r5.so misonet_amd/libmisonet_hip.so
bash tools/gpu_ab_libs.sh bf16x6 1 misonet_amd/libmisonet_hip.so
