#!/bin/bash
# re-entry check of the round's final tree: the whole GPU suite with durations, then smoke()
O=gpurun_out/r06ba; mkdir -p $O
S=$(date +%s)
timeout 1150 python -m pytest tests -q -m gpu -x --durations=45 > $O/pytest_gpu.txt 2>&1; echo "pytest rc $? in $(( $(date +%s) - S )) s" | tee -a $O/pytest_gpu.txt
tail -60 $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
