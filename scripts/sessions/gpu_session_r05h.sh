#!/bin/bash
# round 5, session h: fused nuclei front kernel -- parity + timeline; conv_wino4p output-stage ablations
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05h; mkdir -p $O
timeout 1500 python -m pytest tests/test_postproc_gpu.py -x -q -m gpu 2>&1 | tail -8 > $O/pp_tests.log
cat $O/pp_tests.log
timeout 120 python scripts/dev_pp_nuclei_only.py 8192 2>&1 | tail -1
timeout -k 5 200 rocprofv3 --kernel-trace -d $O/ptrace -o p -- python scripts/dev_pp_nuclei_only.py 8192 > $O/pp.log 2>&1
python scripts/rocprof_summary.py timeline "$(find $O/ptrace -name '*.db' | head -1)" nuc_front $O/timeline.txt
rm -rf $O/ptrace
head -36 $O/timeline.txt
cp cerberus_amd/csrc/conv_wino4p.o /tmp/w4p_keep.o; cp cerberus_amd/libcerberus_hip.so /tmp/lib_keep.so
CERB_VARIANTS=";-DP4_ABL_NOVPASS;-DP4_ABL_NOSTORE;-DP4_ABL_STOREOOB;-DP4_ABL_NOOUT" bash scripts/dev_w4pabl.sh 2>&1 | tee $O/w4p_output_stage_ablations.txt
cp /tmp/w4p_keep.o cerberus_amd/csrc/conv_wino4p.o; cp /tmp/lib_keep.so cerberus_amd/libcerberus_hip.so
