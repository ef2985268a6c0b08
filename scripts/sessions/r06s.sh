#!/bin/bash
# end-of-round fuzz sweep with seeds no earlier session used: drivers, network, N-rank streaming, post-processing
O=gpurun_out/r06s; mkdir -p $O
timeout 900 python tests/tools/dev_fuzz_drivers.py 24 20260929 2>&1 | tail -6 > $O/fuzz_drivers.txt; cat $O/fuzz_drivers.txt
timeout 900 python tests/tools/dev_fuzz_net.py 36 929 2>&1 | tail -4 > $O/fuzz_net.txt; cat $O/fuzz_net.txt
timeout 1200 python tests/tools/dev_fuzz_stream_ranks.py 24 929 2>&1 | tail -4 > $O/fuzz_stream_ranks.txt; cat $O/fuzz_stream_ranks.txt
timeout 900 python tests/tools/dev_fuzz_pp.py 400 929 2>&1 | tail -4 > $O/fuzz_pp.txt; cat $O/fuzz_pp.txt
