#!/bin/bash
# round 5, session aq: inference at the reference's default geometry (448 -> 144) with packed items on / off; smoke; default bench (final check of the tree)
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05aq; mkdir -p $O
timeout 300 python scripts/dev_packed_infer.py 16 448 144 > $O/packed_448_144.txt 2>&1; grep packed_items $O/packed_448_144.txt
timeout 300 python scripts/dev_packed_infer.py 16 448 448 > $O/packed_448_448.txt 2>&1; grep packed_items $O/packed_448_448.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
