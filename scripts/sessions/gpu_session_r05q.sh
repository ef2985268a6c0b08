#!/bin/bash
# round 5, session q: the evidence of the final tree -- rocprofv3 summaries (scripts/profile_r05.sh), the default bench line (40000^2 slide job),
# BASELINE configs[2] (20000^2) and the training line; everything lands in gpurun_out/r05q and gpurun_out/prof_r05, copied into profiles/ afterwards
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05q; mkdir -p $O
( time timeout 900 python bench.py > $O/bench_wsi_40000.json 2> $O/bench_wsi_40000.err ) 2> $O/bench_wsi_40000.time
tail -3 $O/bench_wsi_40000.time
timeout 600 python bench.py --slide 20000 --no-cpu-baseline > $O/bench_wsi_20000.json 2> $O/bench_wsi_20000.err
timeout 600 python bench.py --mode train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err
python - <<'PY'
import json
for f in ("bench_wsi_40000", "bench_wsi_20000"):
    d = json.loads(open("gpurun_out/r05q/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "inference_s", d["config"]["inference_s"], "tail", d["config"]["postproc_and_stitch_s"],
          "e2e", d.get("end_to_end_Mpx_s"), "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"))
    print("   postproc", {t: (v["s"], v["Gpx_s"], v["n_inst"], v["local_bands"]) for t, v in d["postproc"].items()})
    print("   batch_step", d.get("batch_step"), "train", (d.get("train_step") or {}).get("ms_per_step"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
d = json.loads(open("gpurun_out/r05q/bench_train.json").read().strip().splitlines()[-1])
print("train", d["ms_per_step"], "ms/step", d["value"], d["unit"], "attributed", d["roofline"]["attributed_ms"], "traffic", d["roofline"].get("traffic"))
for r in d["kernels"][:16]:
    print("   %-44s %3d %8.3f ms  frac %s  x_alg %s" % (r["kernel"][:44], r["launches"], r["ms_per_step"], r.get("frac"), r.get("traffic_over_algorithmic")))
PY
bash scripts/profile_r05.sh > $O/profile.log 2>&1
tail -5 $O/profile.log
