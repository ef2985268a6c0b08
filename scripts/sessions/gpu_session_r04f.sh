#!/bin/bash
# round-4 GPU session f: suite, bench legs on a small slide, the default bench line, the LDS-staged-patch ablation of conv_wino4p, r04 profiles
cd "$(dirname "$0")/.."
ulimit -c 0
O=gpurun_out/r04f; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
for leg in "--no-ref-tiling" "--no-dat"; do
  echo "== bench 12288 $leg"
  timeout 600 python bench.py --slide 12288 --steps 5 --warmup 2 --no-cpu-baseline --no-train-leg $leg > $O/b.json 2> $O/b.err; echo "rc $?"; tail -2 $O/b.err
  python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r04f/b.json').read().strip().splitlines()[-1])
    print(d['value'], d.get('dat'), d.get('ref_tiling'))
except Exception as e: print("no line", e)
PY
done
echo "== default bench"
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r04f/bench.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['config']['inference_s'], d['config']['postproc_and_stitch_s'], d.get('end_to_end_Mpx_s'))
    print(d.get('dat')); print(d.get('ref_tiling')); print(d.get('batch_step')); print(d['roofline']['whole_step']); print(d.get('train_step',{}).get('ms_per_step')); print(d.get('postproc'))
except Exception as e: print("no line", e)
PY
tail -3 $O/bench.err; cat $O/bench.time
echo "== wino4p LDS-patch ablation"
cp cerberus_amd/csrc/conv_wino4p.o /tmp/w4p_keep.o; cp cerberus_amd/libcerberus_hip.so /tmp/lib_keep.so
CERB_VARIANTS=";-DP4_ABL_NOPATCH;-DP4_ABL_LDSPATCH=0;-DP4_ABL_LDSPATCH=10;-DP4_ABL_LDSPATCH=5;" bash scripts/dev_w4pabl.sh 2>&1 | tee $O/w4p_ldspatch.txt
cp /tmp/w4p_keep.o cerberus_amd/csrc/conv_wino4p.o; cp /tmp/lib_keep.so cerberus_amd/libcerberus_hip.so
echo "== profiles"
bash scripts/profile_r04.sh > $O/profile.log 2>&1; tail -60 $O/profile.log | cut -c1-160
