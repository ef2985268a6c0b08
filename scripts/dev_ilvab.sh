#!/bin/bash
# developer A/B: the interleaved step (P4_ILV / W4B_ILV) against the plain one, both Winograd kernels, same box
cd "$(dirname "$0")/.."
run() {
  timeout 200 python bench.py --mode batch --steps 20 --warmup 3 --no-cpu-baseline > /tmp/ilv.json 2>/tmp/ilv.err || { tail -3 /tmp/ilv.err; return; }
  python - <<PY
import json
d = json.load(open("/tmp/ilv.json"))
print("   step %.3f ms" % d["ms_per_step"], " | ".join("%s x%d %.3f" % (k["kernel"].split("<")[0] + ("res" if "res" in k["kernel"] else "") + ("/2" if "half" in k["kernel"] else ""), k["launches"], k["ms_per_step"]) for k in d["kernels"] if k["kernel"].startswith("conv_wino4")))
PY
}
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000"
for V in "-DP4_ILV=0 -DW4B_ILV=0" "-DP4_ILV=3 -DW4B_ILV=0" "-DP4_ILV=3 -DW4B_ILV=1" "-DP4_ILV=0 -DW4B_ILV=0" "-DP4_ILV=3 -DW4B_ILV=1"; do
  /opt/rocm/bin/hipcc $FL $V -c cerberus_amd/csrc/conv_wino4p.hip -o cerberus_amd/csrc/conv_wino4p.o && /opt/rocm/bin/hipcc $FL $V -c cerberus_amd/csrc/conv_wino4b.hip -o cerberus_amd/csrc/conv_wino4b.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/*.o || exit 1
  echo "=== $V"; run
done
