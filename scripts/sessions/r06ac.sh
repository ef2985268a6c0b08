#!/bin/bash
# glands and lumina at slide scale past 2^31 pixels: larger foreground shares for their heads so that instances survive the size filters
O=gpurun_out/r06ac; mkdir -p $O
GIANT_Q='{"Gland": 0.25, "Lumen": 0.10, "default": 0.02}' timeout 1500 python scripts/dev_r06_giant_slide.py 49152 65536 $O/glands_49152x65536.json > $O/g.log 2>&1; echo "G rc $?"; tail -c 2600 $O/g.log
grep -n "Error\|Traceback" -A14 $O/glands_49152x65536.json.stderr.txt 2>/dev/null | head -60
