#!/bin/bash
# round-6 profiles + the default bench line + train line on the current tree
bash scripts/profile_r06.sh > gpurun_out/profile_r06.log 2>&1
tail -5 gpurun_out/profile_r06.log
O=gpurun_out/r06j; mkdir -p $O
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
tail -3 $O/bench_default.time
python bench.py --mode train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err
python bench.py --slide 20000 --no-train-leg --no-ingest-leg --no-cpu-baseline > $O/bench_wsi_20000.json 2> $O/bench_wsi_20000.err
python - <<'PY'
import json
for f in ("bench_default","bench_train","bench_wsi_20000"):
    try:
        l=json.loads(open('gpurun_out/r06j/%s.json'%f).read().strip().splitlines()[-1])
        print(f, l['value'], l['unit'], l['ms_per_step'], (l.get('roofline') or {}).get('frac'))
    except Exception as e: print(f, "ERR", e)
PY
