#!/bin/bash
# the driver's own bench command on the final tree, with its wall time
O=gpurun_out/r06bc; mkdir -p $O
S=$(date +%s)
timeout 420 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $? in $(( $(date +%s) - S )) s" | tee $O/wall.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06bc/bench.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["other_conv_algos"], d["batch_step"], d["train_step"]["ms_per_step"], d["ingest"]["best"], d["ingest_40x"]["best"] if "ingest_40x" in d else None, d["cpu_baseline"]["value"])
PY
tail -3 $O/bench.err
