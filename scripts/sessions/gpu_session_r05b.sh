#!/bin/bash
# round 5, session b: Winograd-domain weight gradients (conv_wgrad_wino.hip) -- A/B test, the train suite, the training bench line
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_train_loss_gpu.py -x -q -m gpu -k "winograd_domain or backward_pass or whole_train_step or fused_output" 2>&1 | tail -15 > $O/train_tests.log
cat $O/train_tests.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/train.json 2> $O/train.err
python - <<'PY'
import json
for n in ("train",):
    try:
        d = json.loads(open("gpurun_out/r05b/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], "ms/step")
        for r in d["kernels"][:18]:
            print("   %-40s %3d %8.3f ms  %s" % (r["kernel"][:40], r["launches"], r["ms_per_step"], r.get("frac")))
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 $O/train.err
