#!/bin/bash
# Device assembly of one csrc/*.hip translation unit and the instruction mix of one kernel (matched by a substring
# of its mangled name).  usage: isa_stats.sh conv_bf16x6 'bf16x6ILi0ELi8E' [out.s]
R=/root/repo/misonet_amd/csrc
F=$1; K=$2; OUT=${3:-/tmp/isa_$F.s}
EXTRA=""
case $F in conv_bf16|conv_bf16_dma|conv_bf16x6) EXTRA="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $EXTRA --cuda-device-only -S $R/$F.hip -o $OUT 2>&1 | grep -v "warning\|^$"
awk -v k="$K" '$0 ~ "^_Z[A-Za-z0-9_]*"k"[A-Za-z0-9_]*:" {p=1} p {print} p && /\.end_amdhsa_kernel/ {exit}' $OUT > ${OUT%.s}_k.s
echo "kernel lines: $(wc -l < ${OUT%.s}_k.s)"
grep -E "\.vgpr_count|\.vgpr_spill_count|\.sgpr_spill_count|\.lds_size|\.name:" $OUT | grep -A4 "$K" | head -6
awk '{print $1}' ${OUT%.s}_k.s | grep -E "^(v_|s_nop|ds_|buffer_|s_waitcnt|s_barrier|global_|scratch_)" | sed -E 's/_e32$|_e64$//' | sort | uniq -c | sort -rn | head -${4:-28}
