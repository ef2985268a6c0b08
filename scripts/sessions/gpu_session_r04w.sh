#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r04w; mkdir -p $O
timeout 900 python -m pytest tests/test_drivers_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee $O/tests.txt
for ov in 0 1; do
  timeout 600 python bench.py --overlap-tail $ov --no-dat --no-ref-tiling --no-train-leg --no-cpu-baseline > $O/bench_ov$ov.json 2> $O/bench_ov$ov.err
  python - <<P
import json
d=json.loads(open("$O/bench_ov$ov.json").read().strip().splitlines()[-1])
print("overlap $ov: value", d["value"], "ms_per_step", d["ms_per_step"], {k: d["config"].get(k) for k in ("inference_s","tail_s","overlap_tail")}, d["postproc"]["Nuclei"])
P
done | tee $O/summary.txt
tail -3 $O/bench_ov1.err
