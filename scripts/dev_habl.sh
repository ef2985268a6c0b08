#!/bin/bash
# developer ablation: rebuild net_kernels with -D flags on the GPU box and time the heads inside the batch step
cd "$(dirname "$0")/.."
IFS=';'
for FL in ${CERB_VARIANTS:-""}; do
  unset IFS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FL -c cerberus_amd/csrc/net_kernels.hip -o cerberus_amd/csrc/net_kernels.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/*.o || exit 1
  echo "=== flags: [$FL]"
  timeout 120 python -u scripts/dev_head_ab.py 1 1 2>&1 | grep "head_algo 1" | head -2
  IFS=';'
done
