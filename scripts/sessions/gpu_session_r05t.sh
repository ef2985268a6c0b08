#!/bin/bash
# round 5, session t: conv_wino4p ablation row for the fused decoder entry (16 more patch loads + 36 packed adds per thread and chunk)
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05t; mkdir -p $O
cp cerberus_amd/csrc/conv_wino4p.o /tmp/w4p_keep.o; cp cerberus_amd/libcerberus_hip.so /tmp/lib_keep.so
CERB_AUTO_PRECISION=0 CERB_VARIANTS=";-DP4_ABL_PREVADD=1;-DP4_ABL_PREVADD=2;;-DP4_ABL_PREVADD=2" bash scripts/dev_w4pabl.sh 2>&1 | tee $O/w4p_prevadd_ablation.txt
cp /tmp/w4p_keep.o cerberus_amd/csrc/conv_wino4p.o; cp /tmp/lib_keep.so cerberus_amd/libcerberus_hip.so
