#!/bin/bash
# round 4, session m: conv_wino4s with hand-counted waits (weight / bias loads as inline assembly) against the compiler-counted control
cd "$(dirname "$0")/.."
ulimit -c 0
O=gpurun_out/r04m; mkdir -p $O
cp cerberus_amd/csrc/conv_wino4s.o /tmp/w4s_keep.o; cp cerberus_amd/libcerberus_hip.so /tmp/lib_keep.so
CERB_VARIANTS=";-DS4_BUILTIN_WEIGHT_LOADS;-DS4_RING=12 -DS4_WD=11;-DS4_RING=18 -DS4_WD=12;-DS4_ABL_NODMA -DS4_ABL_NOWAIT" bash scripts/dev_w4sabl.sh 2>&1 | tee $O/w4s_manual_waits.txt
cp /tmp/w4s_keep.o cerberus_amd/csrc/conv_wino4s.o; cp /tmp/lib_keep.so cerberus_amd/libcerberus_hip.so
# maxpool with the XCD-contiguous block order: the encoder fixtures + a batch step
timeout 900 python -m pytest tests/test_net_gpu.py -x -q -m gpu -k "golden or planar or maxpool" 2>&1 | tail -3 | tee $O/net_tests.txt
timeout 300 python bench.py --mode batch --steps 20 --warmup 3 --no-cpu-baseline > $O/batch.json 2> $O/batch.err; tail -c 600 $O/batch.json
