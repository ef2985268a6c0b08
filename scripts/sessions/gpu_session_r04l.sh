#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
O=gpurun_out/r04l; mkdir -p $O
cp cerberus_amd/csrc/conv_wino4s.o /tmp/w4s_keep.o; cp cerberus_amd/libcerberus_hip.so /tmp/lib_keep.so
CERB_VARIANTS=";-DS4_RING=18 -DS4_WD=16;-DS4_RING=18 -DS4_WD=12;-DS4_RING=12 -DS4_WD=11;-DS4_RING=18 -DS4_WD=16 -DS4_ABL_NODMA -DS4_ABL_NOWAIT" bash scripts/dev_w4sabl.sh 2>&1 | tee $O/w4s_variants2.txt
cp /tmp/w4s_keep.o cerberus_amd/csrc/conv_wino4s.o; cp /tmp/lib_keep.so cerberus_amd/libcerberus_hip.so
