#!/bin/bash
# round-4 GPU session c: training step after the BatchNorm / copy changes, head task-count variants, wino4b boundary A/B, slide batch size sweep,
# the default bench line with the dat / ref_tiling legs, then the GPU suite
cd "$(dirname "$0")/.."
O=gpurun_out/r04c; mkdir -p $O
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/train.json 2> $O/train.err
echo "== train"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c/train.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'])
for r in d['kernels'][:8]: print(r)
PY
echo "== w4b boundary"
for px in 4096 16384; do CERB_W4B_MAX_PX=$px timeout 200 python scripts/dev_profile_layers.py 32 2>&1 | grep -E "layer1|^total" > $O/w4b_$px.txt; echo "-- max_px $px"; cat $O/w4b_$px.txt; done
echo "== slide batch"
for b in 96 128 192; do CERB_WSI_BATCH=$b timeout 400 python bench.py --slide 20000 --steps 10 --warmup 2 --no-cpu-baseline --no-train-leg --no-dat --no-ref-tiling 2>$O/wsib_$b.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch', d['config']['batch_tiles'], d['value'], d['config']['inference_Mpx_s'])"; done
echo "== default bench"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['inference_s'], d['config']['postproc_and_stitch_s'], d.get('end_to_end_Mpx_s'))
print(d.get('dat')); print(d.get('ref_tiling')); print(d.get('batch_step')); print(d['roofline']['whole_step']); print(d.get('train_step',{}).get('ms_per_step'))
PY
echo "== head variants"
cp cerberus_amd/csrc/net_kernels.o /tmp/nk_keep.o; cp cerberus_amd/libcerberus_hip.so /tmp/lib_keep.so
CERB_VARIANTS="-DHEAD_G_TPW=8;-DHEAD_G_TPW=16;-DHEAD_G_TPW=32;-DHEAD_G_TPW=16 -DHEAD_G_OCC=2" bash scripts/dev_habl.sh 2>&1 | tail -20 | tee $O/head_variants.txt
cp /tmp/nk_keep.o cerberus_amd/csrc/net_kernels.o; cp /tmp/lib_keep.so cerberus_amd/libcerberus_hip.so
echo "== tests"
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
