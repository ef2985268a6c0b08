#!/bin/bash
# the ingest leg on a JPEG 2000-tiled (Aperio 33005) TIFF
O=gpurun_out/r06bh; mkdir -p $O
timeout 110 python bench.py --mode ingest --slide 8192 --ingest-codec jp2k > $O/ingest_jp2k.json 2> $O/ingest_jp2k.err; echo "rc $?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r06bh/ingest_jp2k.json').read().strip().splitlines()[-1]); i=d["ingest"]
    print(d["value"], i["file"], i["decode"]["sweep"], i["inference_resident"], i["best"])
    print([(e["decode_threads"], e["decode_processes"], e["Mpx_s"]) for e in i["end_to_end_from_file"]])
except Exception as e:
    print("no line", e); print(open('gpurun_out/r06bh/ingest_jp2k.err').read()[-1200:])
PY
