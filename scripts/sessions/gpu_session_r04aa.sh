#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r04aa; mkdir -p $O
timeout 1500 python -m pytest tests/test_postproc_gpu.py tests/test_drivers_gpu.py tests/test_cli_gpu.py -x -q -m gpu -k "tiling or reference" 2>&1 | tail -8 | tee $O/tests.txt
timeout 900 python bench.py --no-train-leg --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["end_to_end_Mpx_s"], json.dumps(d["ref_tiling"])[:700])
P
