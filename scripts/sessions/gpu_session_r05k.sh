#!/bin/bash
# round 5, session k: BN statistics passes with adaptive rows per block; per-layer conv records; train tests + bench
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05k; mkdir -p $O
timeout 300 python scripts/dev_train_layers.py bn_ > $O/bn_layers.txt 2>&1
sort -k3 -n -r $O/bn_layers.txt | head -12; grep total $O/bn_layers.txt
timeout 300 python scripts/dev_train_layers.py conv_wino > $O/conv_layers.txt 2>&1
sort -k3 -n -r $O/conv_layers.txt | head -50; grep total $O/conv_layers.txt
timeout 900 python -m pytest tests/test_train_loss_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/train.json 2> $O/train.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05k/train.json").read().strip().splitlines()[-1])
print("train", d["ms_per_step"], "ms/step")
for r in d["kernels"][:30]:
    print("   %-60s %3d %8.3f ms  %s" % (r["kernel"][:60], r["launches"], r["ms_per_step"], r.get("frac")))
PY
