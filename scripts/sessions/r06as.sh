#!/bin/bash
O=gpurun_out/r06as; mkdir -p $O
timeout 1500 python -m pytest tests/test_cli_gpu.py -q -x -k "run_infer_wsi" 2>&1 | tail -4
GIANT_COPIES=2 timeout 1800 python scripts/dev_r06_giant_slide.py 49152 65536 $O/two_slides.json > $O/j.log 2>&1; echo "rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06as/two_slides.json'))
print(d["rc"], d["wall_s"], d["overall_times"])
for l in d["log"]:
    if any(k in l for k in ("Tissue Region","Overall")): print(l.split(" - INFO - ")[-1][:120])
PY
