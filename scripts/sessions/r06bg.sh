#!/bin/bash
# the default line with the lossless-tile ingest leg
O=gpurun_out/r06bg; mkdir -p $O
S=$(date +%s)
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-leg --slide 3072 > $O/bench.json 2> $O/bench.err; echo "bench rc $? in $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06bg/bench.json').read().strip().splitlines()[-1])
print(d["value"], d["ingest"]["best"], d["ingest_40x"]["best"]); print(d["ingest_deflate"])
PY
tail -3 $O/bench.err
