#!/bin/bash
# the 9.7-Gpx streamed slide again on the final tree (allocator keeps its blocks between sub-bands and slides)
O=gpurun_out/r06au; mkdir -p $O
timeout 1800 python scripts/dev_r06_giant_slide.py 98304 98304 $O/giant_98304x98304.json > $O/b.log 2>&1; echo "B rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06au/giant_98304x98304.json'))
print(d["rc"], d["wall_s"], d["memory_plans"], d["overall_times"], d["entries"]); print(d["stdout_tail"][-1]); print(d["stderr_tail"][-3:] if d["rc"] else "")
PY
