#!/bin/bash
# round-4 GPU session k: where conv_wino4s loses its time (ablations: results wrong, timings not)
cd "$(dirname "$0")/.."
ulimit -c 0
O=gpurun_out/r04k; mkdir -p $O
cp cerberus_amd/csrc/conv_wino4s.o /tmp/w4s_keep.o; cp cerberus_amd/libcerberus_hip.so /tmp/lib_keep.so
CERB_VARIANTS=";-DS4_ABL_NOWAIT -DS4_ABL_NOMIDBAR;-DS4_ABL_NODMA -DS4_ABL_NOWAIT;-DS4_ABL_NOREAD;-DS4_ABL_NOXF;-DS4_ABL_NOVWRITE;-DS4_ABL_NOREAD -DS4_ABL_NOVWRITE;-DS4_ABL_NODMA -DS4_ABL_NOWAIT -DS4_ABL_NOREAD -DS4_ABL_NOXF -DS4_ABL_NOVWRITE;-DS4_ABL_NODMA -DS4_ABL_NOWAIT -DS4_ABL_NOMIDBAR -DS4_ABL_NOREAD -DS4_ABL_NOXF -DS4_ABL_NOVWRITE" bash scripts/dev_w4sabl.sh 2>&1 | tee $O/w4s_ablations.txt
cp /tmp/w4s_keep.o cerberus_amd/csrc/conv_wino4s.o; cp /tmp/lib_keep.so cerberus_amd/libcerberus_hip.so
