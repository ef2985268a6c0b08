#!/bin/bash
# slides past 2^31 pixels through run_infer_wsi.py: one resident (3.2 Gpx), one that the plan must stream against the real free HBM (9.7 Gpx)
O=gpurun_out/r06t; mkdir -p $O
free -g | head -2 > $O/host.txt; nproc >> $O/host.txt; df -h /tmp /dev/shm | tail -2 >> $O/host.txt
timeout 900 python -m pytest tests/test_postproc_gpu.py -q -x -k "2_31 or contour" 2>&1 | tail -5
timeout 1200 python scripts/dev_r06_giant_slide.py 49152 65536 $O/giant_49152x65536.json > $O/a.log 2>&1; echo "A rc $?"; tail -c 2500 $O/a.log
timeout 1800 python scripts/dev_r06_giant_slide.py 98304 98304 $O/giant_98304x98304.json > $O/b.log 2>&1; echo "B rc $?"; tail -c 3500 $O/b.log
