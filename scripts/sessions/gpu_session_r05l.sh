#!/bin/bash
# round 5, session l: one-launch re-pack; train tests + bench
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05l; mkdir -p $O
timeout 900 python -m pytest tests/test_train_loss_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/train.json 2> $O/train.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05l/train.json").read().strip().splitlines()[-1])
print("train", d["ms_per_step"], "ms/step")
for r in d["kernels"]:
    if r["kernel"].startswith(("(", "head", "bn_")): print("   %-90s %8.3f ms" % (r["kernel"][:90], r["ms_per_step"]))
PY
