#!/bin/bash
O=gpurun_out/r06f; mkdir -p $O

python - <<"PY" || true
import json
l=json.loads(open('gpurun_out/r06f/b8192.json').read().strip().splitlines()[-1])
print(l['value'], l['postproc'], l['config']['tail_inputs']['INST probability maps'][:300])
PY
tail -3 $O/b8192.err
python -m pytest tests/test_cli_gpu.py -q -m gpu -x -k "own_inference_wrote" 2>&1 | tail -30 > $O/pytest.txt
tail -30 $O/pytest.txt
