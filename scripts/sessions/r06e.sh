#!/bin/bash
O=gpurun_out/r06e; mkdir -p $O
python -m pytest tests/test_drivers_gpu.py -q -m gpu -x -k "streamed" 2>&1 | tail -40 > $O/pytest_stream.txt
tail -30 $O/pytest_stream.txt
python -m pytest tests/test_cli_gpu.py -q -m gpu -x -k "streams_a_slide" 2>&1 | tail -40 > $O/pytest_stream_cli.txt
tail -30 $O/pytest_stream_cli.txt
