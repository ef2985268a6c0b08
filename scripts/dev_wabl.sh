#!/bin/bash
# developer ablation: rebuild conv_wino with -D flags on the GPU box and print the kernel-family table
cd "$(dirname "$0")/.."
for FL in ${CERB_VARIANTS:-""}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FL -c cerberus_amd/csrc/conv_wino.hip -o cerberus_amd/csrc/conv_wino.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/*.o || exit 1
  echo "=== flags: [$FL]"
  timeout 60 python -u scripts/dev_profile_layers.py 32 2>&1 | grep -E "^backbone.layer1.0.conv2|^backbone.layer3.1.conv1|^backbone.layer4.1.conv1|^dec.3.1|^conv_wino|^total"
done
