#!/bin/bash
# developer: rebuild net_kernels.hip with -D flags on the GPU box and time the head
cd "$(dirname "$0")/.."
for FL in ${CERB_VARIANTS:-""}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC ${FL//@/ } -c cerberus_amd/csrc/net_kernels.hip -o cerberus_amd/csrc/net_kernels.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/*.o || exit 1
  echo "=== flags: [$FL]"
  timeout 100 python -u tests/tools/dev_check_net.py 256 2 2>&1 | grep -E "TYPE  out|INST  out" | tr "\n" " "; echo
  timeout 60 python -u scripts/dev_profile_layers.py 32 2>&1 | grep -E "^head |^total|^stem|^maxpool"
done
