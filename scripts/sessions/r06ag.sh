#!/bin/bash
O=gpurun_out/r06ag; mkdir -p $O
timeout 600 python scripts/experiments/dev_r06_decode_scaling.py 2>&1 | grep -v amdgpu.ids | tee $O/decode_scaling.txt
