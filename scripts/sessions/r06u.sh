#!/bin/bash
# past 2^31 pixels with odd sizes (no 4-pixel-wide kernels, odd half-resolution maps, partial edge tiles) and 40 % glass; then the same on two ranks
O=gpurun_out/r06u; mkdir -p $O
timeout 1200 python scripts/dev_r06_giant_slide.py 46349 46351 $O/odd_glass_1rank.json 0.4 1 > $O/c.log 2>&1; echo "C rc $?"; tail -c 2500 $O/c.log
timeout 1500 python scripts/dev_r06_giant_slide.py 46349 46351 $O/odd_glass_2ranks.json 0.4 2 > $O/d.log 2>&1; echo "D rc $?"; tail -c 3500 $O/d.log
