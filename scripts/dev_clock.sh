#!/bin/bash
cd "$(dirname "$0")/.."
for FL in "-DCERB_X=0" "-DCERB_ABL_NOSTAGE -DCERB_ABL_NOEPI -DCERB_ABL_NOWLOAD"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FL -c cerberus_amd/csrc/conv_igemm.hip -o cerberus_amd/csrc/conv_igemm.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/conv_igemm.o cerberus_amd/csrc/net_kernels.o cerberus_amd/csrc/postproc.o cerberus_amd/csrc/slide_kernels.o cerberus_amd/csrc/cerb_api.o || exit 1
  echo "=== flags: [$FL]"
  CERB_CLOCK_PROBE=1 timeout 100 python -u scripts/dev_profile_layers.py 32 2>&1 | grep "clock-probe" | grep -E "dec\.|layer1.0.conv1|layer3.2.conv1|layer4.1.conv1" | tail -12
done
