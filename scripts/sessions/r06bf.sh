#!/bin/bash
# the JPEG ingest legs and the default line's contract after bench.py's codec option and the reader's PIL-codec list
O=gpurun_out/r06bf; mkdir -p $O
S=$(date +%s)
timeout 330 python -m pytest tests/test_cli_gpu.py -q -x -k "ingest_mode_file_fed or contract_single_and_two_ranks" > $O/pytest.txt 2>&1; echo "pytest rc $? in $(( $(date +%s) - S )) s" | tee -a $O/pytest.txt
tail -6 $O/pytest.txt
