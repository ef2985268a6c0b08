#!/bin/bash
O=gpurun_out/r06i; mkdir -p $O
python scripts/dev_r06_layout_ablation.py > $O/layout_ablation.txt 2>&1
cat $O/layout_ablation.txt | cut -c1-220
