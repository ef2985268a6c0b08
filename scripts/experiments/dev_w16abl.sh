#!/bin/bash
# developer ablation: rebuild conv_wino16 with -D flags on the GPU box and time it inside the batch step
cd "$(dirname "$0")/.."
IFS=';'
for FL in ${CERB_VARIANTS:-""}; do
  unset IFS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FL -c cerberus_amd/csrc/${W16_SRC:-conv_wino16}.hip -o cerberus_amd/csrc/${W16_SRC:-conv_wino16}.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/*.o || exit 1
  echo "=== flags: [$FL]"
  timeout 120 python -u scripts/dev_conv_ab.py ${W16_ALGO:-3} ${W16_ALGO:-3} 2>&1 | grep "conv_algo" | cut -c1-200
  IFS=';'
done
