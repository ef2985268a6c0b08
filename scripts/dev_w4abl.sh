#!/bin/bash
# developer ablation: rebuild conv_wino4 with -D flags on the GPU box and time it inside the batch step
#   CERB_VARIANTS="-DW4_TQ=20;-DW4_ABL_NOPATCH" scripts/dev_w4abl.sh
cd "$(dirname "$0")/.."
IFS=';'
for FL in ${CERB_VARIANTS:-""}; do
  unset IFS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000 ${W4_FORM--mllvm -amdgpu-mfma-vgpr-form} $FL -c cerberus_amd/csrc/${W4_SRC:-conv_wino4}.hip -o cerberus_amd/csrc/${W4_SRC:-conv_wino4}.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/*.o || exit 1
  echo "=== flags: [$FL]"
  timeout 120 python -u scripts/dev_conv_ab.py ${W4_ALGO:-5} ${W4_ALGO:-5} 2>&1 | grep "conv_algo" | cut -c1-230
  IFS=';'
done
