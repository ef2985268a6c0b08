#!/bin/bash
# round 5, session a: fused output heads (head_train.hip) -- parity tests, then the training bench line with its family table (fused and separate)
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_train_loss_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/train_tests.log
cat $O/train_tests.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/train_fused.json 2> $O/train_fused.err
CERB_HEAD_UNFUSED=1 timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/train_sep.json 2> $O/train_sep.err
python - <<'PY'
import json
for n in ("train_fused", "train_sep"):
    try:
        d = json.loads(open("gpurun_out/r05a/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], "ms/step")
        for r in d["kernels"][:16]:
            print("   %-40s %3d %8.3f ms  %s" % (r["kernel"][:40], r["launches"], r["ms_per_step"], r.get("frac")))
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 $O/train_fused.err
