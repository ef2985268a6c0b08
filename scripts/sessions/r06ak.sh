#!/bin/bash
# tile mode at a realistic load: 96 PNG tiles of 1000 x 1000 (stain-field texture), the reference's default 448 -> 144 geometry and 256 -> 256
O=gpurun_out/r06ak; mkdir -p $O
python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, "tests/tools"); sys.path.insert(0, ".")
from PIL import Image
from model_dir import stain_atlas, write_sparse_model_dir
atlas = stain_atlas(16)
os.makedirs("/tmp/tiles/in", exist_ok=True)
rs = np.random.RandomState(1)
for i in range(96):
    pick = rs.randint(0, 16, (4, 4))
    img = np.concatenate([np.concatenate([atlas[j] for j in row], axis=1) for row in pick], axis=0)[:1000, :1000]
    Image.fromarray(img).save("/tmp/tiles/in/t%03d.png" % i)
write_sparse_model_dir("/tmp/tiles/model", np.stack(atlas[:4]))
PY
for geo in "256 256 32" "448 144 16"; do set -- $geo
rm -rf /tmp/tiles/out
SECONDS=0; python run_infer_tile.py --model=/tmp/tiles/model --input_dir=/tmp/tiles/in --output_dir=/tmp/tiles/out --patch_input_shape=$1 --patch_output_shape=$2 --batch_size=$3 > $O/tile_$1.log 2> $O/tile_$1.err
echo "geometry $1 -> $2: rc $? in $SECONDS s (CERB_TILE_IO_THREADS=${CERB_TILE_IO_THREADS:-4})"; grep -c "Done Assembling" $O/tile_$1.log
rm -rf /tmp/tiles/out; SECONDS=0; CERB_TILE_IO_THREADS=0 python run_infer_tile.py --model=/tmp/tiles/model --input_dir=/tmp/tiles/in --output_dir=/tmp/tiles/out --patch_input_shape=$1 --patch_output_shape=$2 --batch_size=$3 > /dev/null 2>&1; echo "  main thread only: $SECONDS s"
done
python - <<'PY'
import cProfile, pstats, sys, os, io
sys.argv = ["run_infer_tile.py", "--model=/tmp/tiles/model", "--input_dir=/tmp/tiles/in", "--output_dir=/tmp/tiles/out2", "--patch_input_shape=256", "--patch_output_shape=256", "--batch_size=32"]
import runpy
pr = cProfile.Profile(); pr.enable()
try:
    runpy.run_path("run_infer_tile.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); open("gpurun_out/r06ak/profile_256.txt", "w").write(s.getvalue())
PY
head -30 $O/profile_256.txt | cut -c1-160
timeout 900 python -m pytest tests/test_cli_gpu.py -q -x -k tile 2>&1 | tail -3
