#!/bin/bash
# the default bench line of the final tree (ingest + ingest_40x legs included), timed
O=gpurun_out/r06al; mkdir -p $O
SECONDS=0; python bench.py > $O/bench_wsi_40000.json 2> $O/bench.err; echo "rc $? in $SECONDS s"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06al/bench_wsi_40000.json') if l.startswith('{')][-1])
print(d["value"], d["roofline"]["frac"], d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
print("ingest", d["ingest"].get("best"), d["ingest"].get("error"))
print("ingest_40x", d["ingest_40x"].get("best"), d["ingest_40x"].get("error"), d["ingest_40x"].get("decode"))
PY
