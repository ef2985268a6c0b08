#!/bin/bash
O=gpurun_out/r06o; mkdir -p $O
python scripts/dev_r06_precision_table.py > $O/precision_table.txt 2>&1
cat $O/precision_table.txt
