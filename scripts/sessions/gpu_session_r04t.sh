#!/bin/bash
# round 4, session t: two handles on two streams in the slide job
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r04t; mkdir -p $O
timeout 600 python -m pytest tests/test_drivers_gpu.py -x -q -m gpu -k "runner" 2>&1 | tail -3 | tee $O/tests.txt
for cfg in "1 96" "2 96" "2 64" "2 48"; do set -- $cfg
  CERB_WSI_BATCH=$2 timeout 600 python bench.py --streams $1 --no-dat --no-ref-tiling --no-train-leg --no-cpu-baseline > $O/bench_s$1_b$2.json 2> $O/bench_s$1_b$2.err
  python - <<P
import json
d=json.loads(open("$O/bench_s$1_b$2.json").read().strip().splitlines()[-1])
print("streams $1 batch $2: value", d["value"], "ms_per_step", d["ms_per_step"], "slide", d["config"].get("slide"))
P
done | tee $O/summary.txt
