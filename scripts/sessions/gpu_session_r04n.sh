#!/bin/bash
# round 4, session n: vmcnt ordering probe; conv_wino4s diagnostics (hand-counted waits ignoring the patch loads); default build bitwise
cd "$(dirname "$0")/.."
ulimit -c 0
O=gpurun_out/r04n; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/order_probe scripts/ubench/lds_dma_order_probe.hip 2>/dev/null && timeout 120 /tmp/order_probe | tee $O/lds_dma_order_probe.txt
cp cerberus_amd/csrc/conv_wino4s.o /tmp/w4s_keep.o; cp cerberus_amd/libcerberus_hip.so /tmp/lib_keep.so
CERB_VARIANTS=";-DS4_MANUAL_WAITS -DS4_WAIT_IGNORE_DMA;-DS4_MANUAL_WAITS" bash scripts/dev_w4sabl.sh 2>&1 | tee $O/w4s_manual_waits2.txt
cp /tmp/w4s_keep.o cerberus_amd/csrc/conv_wino4s.o; cp /tmp/lib_keep.so cerberus_amd/libcerberus_hip.so
timeout 900 python -m pytest tests/test_net_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/net_tests.txt
