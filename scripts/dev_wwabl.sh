#!/bin/bash
# developer ablation of conv_wgrad_wino.hip on the GPU box: rebuild with -D flags (results wrong, timings not), time the training step's wgrad layers
#   scripts/dev_wwabl.sh ";-DWW_ABL_NOLOAD;-DWW_ABL_NOE;..."
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
cp cerberus_amd/csrc/conv_wgrad_wino.o /tmp/ww_keep.o; cp cerberus_amd/libcerberus_hip.so /tmp/lib_keep.so
IFS=";"
for FL in $1; do
  unset IFS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FL -c cerberus_amd/csrc/conv_wgrad_wino.hip -o cerberus_amd/csrc/conv_wgrad_wino.o 2>/tmp/cc.err || { echo "=== flags: [$FL] DOES NOT COMPILE"; tail -3 /tmp/cc.err; IFS=";"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/*.o
  echo "=== flags: [$FL]"
  timeout 300 python scripts/dev_train_layers.py wgrad_wino4 2>&1 | grep -E "dec\.3\.1|dec\.2\.1|dec\.0\.0|layer1.0.conv1|layer3.1.conv1|total"
  IFS=";"
done
cp /tmp/ww_keep.o cerberus_amd/csrc/conv_wgrad_wino.o; cp /tmp/lib_keep.so cerberus_amd/libcerberus_hip.so
