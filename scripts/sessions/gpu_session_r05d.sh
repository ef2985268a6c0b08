#!/bin/bash
# round 5, session d: streaming tests again; per-layer table + SQ counters of the Winograd-domain weight gradient
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests/test_drivers_gpu.py tests/test_cli_gpu.py -x -q -m gpu -k "streamed or streams_a_slide" 2>&1 | tail -25 > $O/stream_tests.log
cat $O/stream_tests.log
timeout 300 python scripts/dev_train_layers.py wgrad_wino4 > $O/wgrad_layers.txt 2>&1; tail -40 $O/wgrad_layers.txt
bash scripts/dev_pmc_train.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_LDS" "wgrad_wino_kernel" > $O/wgrad_pmc.txt 2>&1
cat $O/wgrad_pmc.txt
