#!/bin/bash
# degenerate slide sizes through ONE run_infer_wsi.py process (a directory of slides), then --reference_tiling past 2^31 pixels
O=gpurun_out/r06aa; mkdir -p $O
D=/tmp/degenerate; rm -rf $D; mkdir -p $D/in
i=0
for s in 100x100 255x257 256x256 257x255 300x70000 70000x300 513x513 2x4000 4000x2; do i=$((i+1)); echo "synthetic:$s:$i" > $D/in/s$(printf %02d $i)_$s.txt; done
timeout 900 python run_infer_wsi.py --synthetic --input_dir=$D/in --wsi_file_ext=.txt --output_dir=$D/out --logging_dir=$D/log --batch_size=16 --patch_input_shape=256 --patch_output_shape=256 > $O/degenerate.log 2>&1
echo "degenerate rc $?" | tee -a $O/degenerate.log; tail -25 $O/degenerate.log
ls $D/out/dat | tee $O/degenerate_dat.txt
python - <<'PY' | tee -a $O/degenerate.log
import joblib, glob, os
for f in sorted(glob.glob('/tmp/degenerate/out/dat/*.dat')):
    d = joblib.load(f)
    print(os.path.basename(f), d["proc_dimensions"].tolist(), {k: len(v) for k, v in d.items() if isinstance(v, dict) and k in ("Nuclei", "Gland", "Lumen")})
PY
# same, the reference's default 448 -> 144 geometry
timeout 900 python run_infer_wsi.py --synthetic --input_dir=$D/in --wsi_file_ext=.txt --output_dir=$D/out448 --logging_dir=$D/log448 --batch_size=8 > $O/degenerate448.log 2>&1
echo "degenerate448 rc $?" | tee -a $O/degenerate448.log; tail -12 $O/degenerate448.log; ls $D/out448/dat | wc -l
