#!/bin/bash
python -m pytest tests/test_cli_gpu.py -q -m gpu -x -k "logit_guard_counts" 2>&1 | tail -15
