#!/bin/bash
python scripts/experiments/dev_r06_second_slide_prep.py 2>&1 | grep -v amdgpu.ids
