#!/bin/bash
# round-4 GPU session j: where conv_wino4s loses its time (ablations: results wrong, timings not), test durations
cd "$(dirname "$0")/.."
ulimit -c 0
O=gpurun_out/r04j; mkdir -p $O
cp cerberus_amd/csrc/conv_wino4s.o /tmp/w4s_keep.o; cp cerberus_amd/libcerberus_hip.so /tmp/lib_keep.so
CERB_VARIANTS=";-DS4_ABL_NOWAIT;-DS4_ABL_NOWAIT -DS4_ABL_NOMIDBAR;-DS4_ABL_NODMA -DS4_ABL_NOWAIT;-DS4_ABL_NOREAD;-DS4_ABL_NOXF;-DS4_ABL_NOVWRITE;-DS4_ABL_NODMA -DS4_ABL_NOWAIT -DS4_ABL_NOREAD -DS4_ABL_NOXF -DS4_ABL_NOVWRITE" bash scripts/dev_w4sabl.sh 2>&1 | grep -v MISMATCH | tee $O/w4s_ablations.txt
cp /tmp/w4s_keep.o cerberus_amd/csrc/conv_wino4s.o; cp /tmp/lib_keep.so cerberus_amd/libcerberus_hip.so
timeout 3000 python -m pytest tests -m gpu -x -q --durations=25 > $O/pytest.txt 2>&1
grep -A30 "slowest" $O/pytest.txt | head -34; tail -3 $O/pytest.txt
