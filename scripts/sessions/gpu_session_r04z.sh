#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r04z; mkdir -p $O
timeout 900 python -m pytest tests/test_drivers_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
timeout 600 python bench.py --no-dat --no-ref-tiling --no-train-leg --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --slide 14000 --no-dat --no-ref-tiling --no-train-leg --no-cpu-baseline > $O/bench14k.json 2> $O/bench14k.err
timeout 600 python bench.py --slide 14000 --streams 1 --no-dat --no-ref-tiling --no-train-leg --no-cpu-baseline > $O/bench14k_s1.json 2> $O/bench14k_s1.err
python - <<P
import json
for f in ("bench","bench14k","bench14k_s1"):
    d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["config"].get("streams"), d["config"].get("inference_Mpx_s"), d["postproc"]["Nuclei"]["s"])
P
