#!/bin/bash
# bench.py's slide job past 2^31 pixels (47104^2 = 2.22 Gpx): structured maps with glands and lumina through the tail, dictionary leg included
O=gpurun_out/r06ad; mkdir -p $O
timeout 1500 python bench.py --slide 47104 --steps 2 --warmup 1 --no-train-leg --no-ingest-leg --no-cpu-baseline > $O/bench_47104.json 2> $O/bench_47104.err; echo "rc $?"
tail -5 $O/bench_47104.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r06ad/bench_47104.json') if x.startswith('{')]
d=json.loads(l[-1])
print({k:d[k] for k in ("value","ms_per_step","config")})
print(d.get("postproc")); print({k:v for k,v in d.get("dat",{}).items() if k!="note"})
PY
