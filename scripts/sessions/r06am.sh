#!/bin/bash
O=gpurun_out/r06am; mkdir -p $O
timeout 900 python -m pytest tests/test_drivers_gpu.py -q -x -k "stored_finer or pipelined" 2>&1 | tail -3
timeout 1500 python bench.py --mode ingest --slide 20000 --ingest-base-mpp 0.25 > $O/ingest40x_20000.json 2> $O/ingest40x.err; echo "rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06am/ingest40x_20000.json') if l.startswith('{')][-1])
i=d["ingest"]; print(d["value"], i["inference_resident"])
for e in i["end_to_end_from_file"]: print(e)
PY
timeout 1500 python bench.py --mode ingest --slide 20000 > $O/ingest_20000.json 2> $O/ingest.err; echo "rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06am/ingest_20000.json') if l.startswith('{')][-1])
i=d["ingest"]; print(d["value"], i["inference_resident"], i["best"])
for e in i["end_to_end_from_file"]: print(e)
PY
