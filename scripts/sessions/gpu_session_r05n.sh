#!/bin/bash
# round 5, session n: the default bench line (40000^2 slide job incl. dat / ref_tiling / train legs and the CPU baseline sweep) and the train line
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05n; mkdir -p $O
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
tail -3 $O/bench_default.time
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05n/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "slide", d["config"]["slide"], "streams", d["config"]["streams"])
print("inference_s", d["config"]["inference_s"], "tail", d["config"]["postproc_and_stitch_s"], "end_to_end", d.get("end_to_end_Mpx_s"))
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic") if k in d["roofline"]})
print("postproc", d["postproc"])
print("dat", {k: v for k, v in (d.get("dat") or {}).items() if k != "note"})
print("ref_tiling", {k: v for k, v in (d.get("ref_tiling") or {}).items() if k not in ("note", "rank0")})
print("batch_step", d["batch_step"])
ts = d.get("train_step") or {}
print("train_step", ts.get("ms_per_step"), ts.get("tiles_s"), (ts.get("roofline") or {}).get("traffic"))
cb = d.get("cpu_baseline") or {}
print("cpu_baseline", cb.get("value"), cb.get("cores"), cb.get("sweep"), cb.get("all_cores"), cb.get("postproc_nuclei_Mpx_s_1core"))
PY
timeout 600 python bench.py --mode train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05n/bench_train.json").read().strip().splitlines()[-1])
print("train", d["ms_per_step"], "ms/step", d["value"], d["unit"], "attributed", d["roofline"]["attributed_ms"], "traffic", d["roofline"].get("traffic"))
for r in d["kernels"][:24]:
    print("   %-44s %3d %8.3f ms  frac %s  traffic %s  x_alg %s" % (r["kernel"][:44], r["launches"], r["ms_per_step"], r.get("frac"), r.get("traffic"), r.get("traffic_over_algorithmic")))
print("cpu", d.get("cpu_baseline"))
PY
