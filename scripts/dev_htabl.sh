#!/bin/bash
# developer A/B of head_train.hip on the GPU box: rebuild with -D flags, time the training step's head families
#   scripts/dev_htabl.sh ";-DHB2_BULK_FETCH"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
cp cerberus_amd/csrc/head_train.o /tmp/ht_keep.o; cp cerberus_amd/libcerberus_hip.so /tmp/lib_keep.so
IFS=";"
for FL in $1; do
  unset IFS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FL -c cerberus_amd/csrc/head_train.hip -o cerberus_amd/csrc/head_train.o 2>/tmp/cc.err || { echo "=== flags: [$FL] DOES NOT COMPILE"; tail -3 /tmp/cc.err; IFS=";"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/*.o
  echo "=== flags: [$FL]"
  timeout 300 python scripts/dev_train_layers.py head_ 2>&1 | grep -E "head_bwd2|head_bwd1|head_fwd|total" | sort | uniq -c | sort -k3 | head -40
  IFS=";"
done
cp /tmp/ht_keep.o cerberus_amd/csrc/head_train.o; cp /tmp/lib_keep.so cerberus_amd/libcerberus_hip.so
