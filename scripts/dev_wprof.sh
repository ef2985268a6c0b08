#!/bin/bash
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWPROF -c cerberus_amd/csrc/conv_wino.hip -o cerberus_amd/csrc/conv_wino.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/*.o || exit 1
timeout 120 python -u scripts/dev_wprof.py
