#!/bin/bash
# the native tile reader (libcerberus_host.so) on the GPU box: the GPU tests that read deflate / LZW / JPEG TIFFs, then the ingest leg on 12288^2 deflate and LZW files
O=gpurun_out/r06be; mkdir -p $O
S=$(date +%s)
timeout 300 python -m pytest tests/test_cli_gpu.py tests/test_drivers_gpu.py -q -x -k "pyramidal_tiff or lossless_tiles or stored_finer or pipelined_band_upload" > $O/pytest.txt 2>&1; echo "pytest rc $? in $(( $(date +%s) - S )) s" | tee -a $O/pytest.txt
tail -5 $O/pytest.txt
for c in lzw deflate; do
  timeout 330 python bench.py --mode ingest --slide 12288 --ingest-codec $c > $O/ingest_$c.json 2> $O/ingest_$c.err; echo "ingest $c rc $?"
  python - $c <<'PY'
import json, sys
c = sys.argv[1]
try:
    d = json.loads(open('gpurun_out/r06be/ingest_%s.json' % c).read().strip().splitlines()[-1])
    i = d["ingest"]
    print(c, d["value"], i["file"], i["decode"]["sweep"], i["inference_resident"], i["best"])
    print([(e["decode_threads"], e["Mpx_s"], e["decode_s_in_producer"]) for e in i["end_to_end_from_file"]])
except Exception as e:
    print(c, "no line:", e); print(open('gpurun_out/r06be/ingest_%s.err' % c).read()[-1500:])
PY
done
