#!/bin/bash
# --msk_dir on a slide past 2^31 pixels (40 % glass): patches selected by the mask, nuclei through row bands, gland / lumen per tissue region
O=gpurun_out/r06w; mkdir -p $O
timeout 1500 python scripts/dev_r06_giant_slide.py 49152 65536 $O/mask_49152x65536.json 0.4 1 mask > $O/e.log 2>&1; echo "E rc $?"; tail -c 3500 $O/e.log
grep -n "Error\|Traceback" -A14 $O/mask_49152x65536.json.stderr.txt 2>/dev/null | head -60
