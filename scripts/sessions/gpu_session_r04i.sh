#!/bin/bash
# round-4 GPU session i: conv_wino4s after the spill fixes: full check + variants; then the GPU suite
cd "$(dirname "$0")/.."
ulimit -c 0
O=gpurun_out/r04i; mkdir -p $O
timeout 600 python scripts/dev_w4s_check.py --time 2>&1 | grep -v "amdgpu.ids" | tee $O/w4s_check.txt
cp cerberus_amd/csrc/conv_wino4s.o /tmp/w4s_keep.o; cp cerberus_amd/libcerberus_hip.so /tmp/lib_keep.so
CERB_VARIANTS="-DS4_WD=11;-DS4_RING=18 -DS4_WD=12;-DS4_RING=18 -DS4_WD=16;-DS4_WD=6;-DS4_RQ=18 -DS4_NDMA_LATE=1" bash scripts/dev_w4sabl.sh 2>&1 | tee $O/w4s_variants.txt
cp /tmp/w4s_keep.o cerberus_amd/csrc/conv_wino4s.o; cp /tmp/lib_keep.so cerberus_amd/libcerberus_hip.so
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
