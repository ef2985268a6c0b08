#!/bin/bash
# round 5, session e: wgrad_wino with 8-byte loads + half-wave swap; streaming tests
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests/test_train_loss_gpu.py tests/test_drivers_gpu.py tests/test_cli_gpu.py -x -q -m gpu -k "winograd_domain or backward_pass or streamed or streams_a_slide" 2>&1 | tail -12 > $O/tests.log
cat $O/tests.log
timeout 300 python scripts/dev_train_layers.py wgrad_wino4 > $O/wgrad_layers.txt 2>&1; grep -E "dec\.|layer1.0.conv1|layer2.1.conv1|layer3.1.conv1|layer4.1.conv1|total" $O/wgrad_layers.txt
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/train.json 2> $O/train.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05e/train.json").read().strip().splitlines()[-1])
print("train", d["ms_per_step"], "ms/step")
for r in d["kernels"][:12]:
    print("   %-40s %3d %8.3f ms  %s" % (r["kernel"][:40], r["launches"], r["ms_per_step"], r.get("frac")))
PY
