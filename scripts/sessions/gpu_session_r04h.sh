#!/bin/bash
# round-4 GPU session h: direct-to-LDS semantics probe, conv_wino4s check (bitwise + timing), then the GPU suite
cd "$(dirname "$0")/.."
ulimit -c 0
O=gpurun_out/r04h; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/ubench/lds_dma_probe.hip -o /tmp/lds_dma_probe 2>/dev/null && timeout 60 /tmp/lds_dma_probe | tee $O/lds_dma_probe.txt
timeout 600 python scripts/dev_w4s_check.py --time 2>&1 | grep -v "amdgpu.ids" | tee $O/w4s_check.txt
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
