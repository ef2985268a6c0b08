#!/bin/bash
# two ranks on the one GPU, 125 GB each: both must stream their bands of a 2.15-Gpx odd-sized slide with 40 % glass
O=gpurun_out/r06v; mkdir -p $O
CERB_HBM_BUDGET_GB=125 timeout 1500 python scripts/dev_r06_giant_slide.py 46349 46351 $O/odd_glass_2ranks_streamed.json 0.4 2 > $O/d.log 2>&1; echo "D rc $?"; tail -c 3000 $O/d.log
grep -n "Error\|error\|Traceback" -A12 $O/odd_glass_2ranks_streamed.json.stderr.txt 2>/dev/null | head -80
