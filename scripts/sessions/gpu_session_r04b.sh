#!/bin/bash
# round-4 GPU session b: parity-achieved table, head A/B, per-layer tables at small batches (Infinity-Cache residency probe), GPU suite
cd "$(dirname "$0")/.."
O=gpurun_out/r04b; mkdir -p $O
timeout 600 python scripts/dev_parity_achieved.py > $O/parity.json 2> $O/parity.err
timeout 300 python scripts/dev_head_ab.py 2 1 2 1 > $O/head_ab.txt 2>&1
for nb in 1 2 4 8; do timeout 200 python scripts/dev_profile_layers.py $nb 2>&1 | grep -E "dec\.[23]|heads|total" > $O/layers_b$nb.txt; done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt; cat $O/head_ab.txt; head -c 1500 $O/parity.json; for nb in 1 2 4 8; do echo "== batch $nb"; cat $O/layers_b$nb.txt; done
