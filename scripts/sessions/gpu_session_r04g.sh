#!/bin/bash
# round-4 GPU session g: the full GPU suite (the gradient test's printed error figures kept), a direct-to-LDS semantics probe, the default bench line
cd "$(dirname "$0")/.."
ulimit -c 0
O=gpurun_out/r04g; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
timeout 300 python -m pytest tests/test_train_loss_gpu.py -m gpu -q -s -k "backward_pass or upstream" 2>&1 | grep -E "worst|element-wise|passed|failed" | tee $O/grad_bars.txt
if [ -f scripts/ubench/lds_dma_probe.hip ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/ubench/lds_dma_probe.hip -o /tmp/lds_dma_probe && timeout 60 /tmp/lds_dma_probe | tee $O/lds_dma_probe.txt
fi
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r04g/bench.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['config']['inference_s'], d['config']['postproc_and_stitch_s'], d.get('end_to_end_Mpx_s'))
    print(d.get('dat')); print(d.get('ref_tiling')); print(d.get('batch_step')); print(d.get('train_step',{}).get('ms_per_step'))
except Exception as e: print("no line", e)
PY
tail -3 $O/bench.err; cat $O/bench.time
