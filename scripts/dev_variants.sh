#!/bin/bash
# developer A/B: rebuild conv_igemm with each flag set given as arguments (quote each set) and print the kernel-family table
cd "$(dirname "$0")/.."
for FL in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FL -c cerberus_amd/csrc/conv_igemm.hip -o cerberus_amd/csrc/conv_igemm.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/*.o || exit 1
  echo "=== flags: [$FL]"
  for rep in 1 2; do timeout 60 python -u scripts/dev_profile_layers.py 32 2>&1 | grep -E "^conv_igemm<ks3,s1|^total" | tr '\n' '|'; echo; done
done
