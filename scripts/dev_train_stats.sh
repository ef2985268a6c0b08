cd "$(dirname "$0")/.."; export TMPDIR=/tmp; O=gpurun_out/r04ab; mkdir -p $O
timeout -k 5 300 rocprofv3 --kernel-trace -d $O/ts -o t -- python bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > $O/train.json 2> $O/t.log
python - <<P
import sqlite3, glob
db=glob.glob("$O/ts/**/*.db", recursive=True)[0]
c=sqlite3.connect(db)
rows=list(c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
for pat in ("bn_partial_kernel","bn_finalize_kernel","pw_mfma_kernel<64, 96"):
    d=[(round((e-s)/1e3,1),g,w) for n,s,e,g,w in rows if pat in n]
    print(pat, len(d), d[-3:])
P
rm -rf $O/ts
