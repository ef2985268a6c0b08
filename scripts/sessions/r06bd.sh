#!/bin/bash
# the multi-rank GPU tests after halo_exchange's dense-buffer change (device tensors through the facade and through one-rank RCCL)
O=gpurun_out/r06bd; mkdir -p $O
S=$(date +%s)
timeout 400 python -m pytest tests/test_drivers_gpu.py tests/test_cli_gpu.py -q -x -k "sharded_postproc_on_device or halo_exchange_of_cuda or several_ranks or per_rank or two_ranks_equals_one_rank or nccl_branch_at_world_one or eight_ranks" > $O/pytest.txt 2>&1; echo "pytest rc $? in $(( $(date +%s) - S )) s" | tee -a $O/pytest.txt
tail -8 $O/pytest.txt
