#!/bin/bash
# a directory of three 3.2-Gpx slides through one process: every slide planned against the same free HBM, dictionaries written underneath the next slide
O=gpurun_out/r06an; mkdir -p $O
GIANT_COPIES=3 timeout 1800 python scripts/dev_r06_giant_slide.py 49152 65536 $O/three_slides.json > $O/j.log 2>&1; echo "rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06an/three_slides.json'))
print(d["rc"], d["wall_s"]); print(d["memory_plans"]); print(d["overall_times"]); print(d["stdout_tail"][-4:]); print(d.get("entries"))
PY
