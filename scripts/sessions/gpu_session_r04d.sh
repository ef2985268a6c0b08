#!/bin/bash
# round-4 GPU session d: the default bench line with the dat / ref_tiling legs, then the GPU suite
cd "$(dirname "$0")/.."
O=gpurun_out/r04d; mkdir -p $O
( time timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04d/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['inference_s'], d['config']['postproc_and_stitch_s'], d.get('end_to_end_Mpx_s'))
print(d.get('dat')); print(d.get('ref_tiling')); print(d.get('batch_step')); print(d['roofline']['whole_step']); print(d.get('train_step',{}).get('ms_per_step')); print(d.get('postproc'))
PY
tail -3 $O/bench.err; cat $O/bench.time
timeout 2700 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
