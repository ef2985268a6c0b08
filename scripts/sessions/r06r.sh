#!/bin/bash
python -m pytest tests/test_cli_gpu.py -q -m gpu -x -k "ingest_mode or contract_single" 2>&1 | tail -15
