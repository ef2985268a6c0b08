#!/bin/bash
O=gpurun_out/r06g; mkdir -p $O

python bench.py --mode ingest --slide 20000 > $O/ingest_12288.json 2> $O/ingest_12288.err
tail -3 $O/ingest_12288.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r06g/ingest_12288.json').read().strip().splitlines()[-1])
i=l['ingest']
for k in ('file','decode','inference_resident','upload','best'): print(k, i[k])
for e in i['end_to_end_from_file']: print(e)
PY
python -m pytest tests/test_drivers_gpu.py -q -m gpu -x -k "pipelined_band_upload" 2>&1 | tail -5
python -m pytest tests/test_cli_gpu.py -q -m gpu -x -k "pyramidal_tiff" 2>&1 | tail -5
