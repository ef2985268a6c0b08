#!/bin/bash
# developer ablation: rebuild conv_wino4p with -D flags on the GPU box and time it inside the batch step
#   CERB_VARIANTS=";-DP4_TQ=20;-DP4_ABL_NOPATCH" scripts/dev_w4pabl.sh        (an empty variant = the defaults)
cd "$(dirname "$0")/.."
IFS=';'
for FL in ${CERB_VARIANTS:-""}; do
  unset IFS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000 $FL -c cerberus_amd/csrc/conv_wino4p.hip -o cerberus_amd/csrc/conv_wino4p.o 2>/tmp/cc.err || { echo "=== flags: [$FL] DOES NOT COMPILE"; IFS=";"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/*.o || exit 1
  echo "=== flags: [$FL]"
  timeout 120 python bench.py --mode batch --steps 20 --warmup 3 --no-cpu-baseline > /tmp/w4pabl.json 2>/tmp/w4pabl.err || { tail -5 /tmp/w4pabl.err; IFS=';'; continue; }
  python - <<PY
import json
d = json.load(open("/tmp/w4pabl.json"))
r = [k for k in d["kernels"] if k["kernel"].startswith("conv_wino4p")][0]
print("step %.3f ms | conv_wino4p x%d %.4f ms (%.4f per launch) executed-MFMA frac %.4f" % (d["ms_per_step"], r["launches"], r["ms_per_step"], r["ms_per_step"] / r["launches"], r["frac"]))
PY
  IFS=';'
done
