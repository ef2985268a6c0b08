#!/bin/bash
O=gpurun_out/r06m; mkdir -p $O
python -m pytest tests/test_postproc_gpu.py tests/test_drivers_gpu.py -q -m gpu -x 2>&1 | tail -4
python tests/tools/dev_fuzz_pp.py 300 777 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o f -- python $R/scripts/dev_pp_nuclei_only.py 8192 > /tmp/pf.log 2>&1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o w -- python $R/scripts/dev_pp_nuclei_only.py 8192 > /tmp/pw.log 2>&1
cd $R
python scripts/rocprof_summary.py pmc "$(find /tmp/pf -name '*.db' | head -1)" "$(find /tmp/pw -name '*.db' | head -1)" $O/pp_pmc.json
python scripts/rocprof_summary.py pmc_table $O/pp_pmc.json 4 67108864 12 $O/pp_bytes_per_pass.txt
head -30 $O/pp_bytes_per_pass.txt | cut -c1-150
