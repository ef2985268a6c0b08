#!/bin/bash
O=gpurun_out/r06af; mkdir -p $O
timeout 900 python -m pytest tests/test_drivers_gpu.py -q -x -k "stored_finer or pipelined" 2>&1 | tail -25 > $O/pytest.txt; cat $O/pytest.txt
for mpp in 0.25 0.2528 0.5; do
GIANT_BASE_MPP=$mpp timeout 1500 python scripts/dev_r06_giant_slide.py 32768 32768 $O/base${mpp}_32768.json > $O/h$mpp.log 2>&1; echo "mpp $mpp rc $?"; grep -E "Inference Time|Mpx/s|rc\"" $O/h$mpp.log
done
