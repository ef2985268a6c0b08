#!/bin/bash
# --reference_tiling (the reference's 4096^2 nuclei tiles) on a slide past 2^31 pixels
O=gpurun_out/r06ab; mkdir -p $O
GIANT_EXTRA_FLAGS="--reference_tiling" timeout 1500 python scripts/dev_r06_giant_slide.py 49152 65536 $O/ref_tiling_49152x65536.json > $O/f.log 2>&1; echo "F rc $?"; tail -c 3000 $O/f.log
grep -n "Error\|Traceback" -A14 $O/ref_tiling_49152x65536.json.stderr.txt 2>/dev/null | head -60
