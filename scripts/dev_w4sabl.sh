#!/bin/bash
# developer ablation: rebuild conv_wino4s with -D flags on the GPU box, check it bitwise against conv_wino4p and time it inside the batch step
#   CERB_VARIANTS=";-DS4_WD=11;-DS4_RING=18 -DS4_WD=12" scripts/dev_w4sabl.sh        (an empty variant = the defaults)
cd "$(dirname "$0")/.."
IFS=';'
for FL in ${CERB_VARIANTS:-""}; do
  unset IFS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000 $FL -c cerberus_amd/csrc/conv_wino4s.hip -o cerberus_amd/csrc/conv_wino4s.o 2>/tmp/cc.err || { echo "=== flags: [$FL] DOES NOT COMPILE"; tail -3 /tmp/cc.err; IFS=";"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/*.o || exit 1
  echo "=== flags: [$FL]"
  timeout 200 python scripts/dev_w4s_check.py --quick --time 2>&1 | grep -E "bitwise|planar 2|Error" | head -4
  IFS=';'
done
