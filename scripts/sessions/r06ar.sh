#!/bin/bash
# two ranks (sharing the box's GPU) on a 40x TIFF: each reads its own band through decode processes + the device reduction; one rank for comparison
O=gpurun_out/r06ar; mkdir -p $O
GIANT_BASE_MPP=0.2528 timeout 900 python scripts/dev_r06_giant_slide.py 32768 36864 $O/r1.json 0.3 1 > $O/r1.log 2>&1; echo "1 rank rc $?"
GIANT_BASE_MPP=0.2528 timeout 900 python scripts/dev_r06_giant_slide.py 32768 36864 $O/r2.json 0.3 2 > $O/r2.log 2>&1; echo "2 ranks rc $?"
python - <<'PY'
import json
for f in ("r1","r2"):
    d=json.load(open('gpurun_out/r06ar/%s.json'%f)); print(f, d["rc"], d["slide"], d["entries"], d["stdout_tail"][-1:], d["stderr_tail"][-3:] if d["rc"] else "")
PY
