#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r04ac; mkdir -p $O
timeout 900 python -m pytest tests/test_train_loss_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee $O/train_tests.txt
timeout 600 python bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > $O/train.json 2> $O/train.err
CERB_BN_STATS_PASS=1 timeout 600 python bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > $O/train_pass.json 2> $O/train_pass.err
timeout 900 python -m pytest tests/test_net_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/net_tests.txt
timeout 300 python bench.py --mode batch --steps 20 --warmup 3 --no-cpu-baseline > $O/batch.json 2> $O/batch.err
python - <<P
import json
for f in ("train","train_pass"):
    d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], [(k["kernel"],k["ms_per_step"]) for k in d["kernels"] if k["kernel"] in ("bn_fwd","conv_wino4<f4x4,16x16x2>","conv_wino4b<f4x4,16x16>")])
d=json.loads(open("$O/batch.json").read().strip().splitlines()[-1]); print("batch", d["value"], d["ms_per_step"])
P
