#!/bin/bash
O=gpurun_out/r06k; mkdir -p $O
python -m pytest tests/test_postproc_gpu.py -q -m gpu -x 2>&1 | tail -5
python scripts/dev_pp_nuclei_only.py 8192 2>&1 | tail -5
CERB_DEV_LIB=1 CERB_PP_SEAM_COLUMNS=0 python scripts/dev_pp_nuclei_only.py 8192 2>&1 | tail -3
python scripts/dev_pp_nuclei_only.py 8192 2>&1 | tail -3
