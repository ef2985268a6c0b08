#!/bin/bash
cd "$(dirname "$0")/.."
for FL in ${CERB_VARIANTS:-""}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC ${FL//@/ } -c cerberus_amd/csrc/conv_wino.hip -o cerberus_amd/csrc/conv_wino.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/*.o || exit 1
  echo "=== flags: [$FL]"
  timeout 100 python -u tests/tools/dev_check_net.py 256 2 2>&1 | grep -E "INST   out|INST  out|batch 32"
done
