#!/bin/bash
O=gpurun_out/r06av; mkdir -p $O
timeout 1500 python -m pytest tests/test_drivers_gpu.py tests/test_cli_gpu.py -q -x -k "streamed or streams" 2>&1 | tail -6
timeout 1800 python scripts/dev_r06_giant_slide.py 98304 98304 $O/giant_98304x98304.json > $O/b.log 2>&1; echo "B rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06av/giant_98304x98304.json'))
print(d["rc"], d["wall_s"], d["overall_times"], d["entries"], d["entries_past_int32_raster_index"]); print(d["stdout_tail"][-1]); print(d["stderr_tail"][-3:] if d["rc"] else "")
for l in d["log"]:
    if any(k in l for k in ("Inference Time","Labelling","Tissue Region","Dictionary","Overall")): print(l.split(" - INFO - ")[-1][:140])
PY
