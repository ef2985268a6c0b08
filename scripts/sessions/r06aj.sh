#!/bin/bash
O=gpurun_out/r06aj; mkdir -p $O
timeout 1500 python -m pytest tests/test_cli_gpu.py -q -x -k "ingest" 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt
timeout 1500 python bench.py --mode ingest --slide 20000 --ingest-base-mpp 0.25 > $O/ingest40x_20000.json 2> $O/ingest40x.err; echo "rc $?"; tail -3 $O/ingest40x.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06aj/ingest40x_20000.json') if l.startswith('{')][-1])
i=d["ingest"]; print(d["value"], i["stored"], i["inference_resident"]); print(i["decode"]["sweep"]); 
for e in i["end_to_end_from_file"]: print(e)
print(i["best"])
PY
