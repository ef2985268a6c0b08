#!/bin/bash
# developer ablation: rebuild net_kernels.hip with -D flags on the GPU box and time the planar up-sampling inside the batch step
#   CERB_VARIANTS=";-DUPP_PLAIN_STORE;-DUPP_GRID=32" scripts/dev_uppabl.sh
cd "$(dirname "$0")/.."
IFS=';'
for FL in ${CERB_VARIANTS:-""}; do
  unset IFS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FL -c cerberus_amd/csrc/net_kernels.hip -o cerberus_amd/csrc/net_kernels.o 2>/tmp/cc.err || { echo "=== flags: [$FL] DOES NOT COMPILE"; IFS=";"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/*.o || exit 1
  echo "=== flags: [$FL]"
  timeout 120 python bench.py --mode batch --steps 20 --warmup 3 --no-cpu-baseline > /tmp/uppabl.json 2>/tmp/uppabl.err || { tail -5 /tmp/uppabl.err; IFS=';'; continue; }
  python - <<PY
import json
d = json.load(open("/tmp/uppabl.json"))
for k in d["kernels"]:
    if k["kernel"].startswith("upsample2_add") or k["kernel"] in ("head_group", "maxpool3x3s2"):
        print("   %-24s x%d %.4f ms frac %.4f" % (k["kernel"], k["launches"], k["ms_per_step"], k.get("frac") or 0))
print("   step %.3f ms" % d["ms_per_step"])
PY
  IFS=';'
done
