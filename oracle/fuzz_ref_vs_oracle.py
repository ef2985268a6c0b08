"""Pin the C oracle harder: random maps (same distribution as tests/tools/dev_fuzz_pp.py, tie-heavy kinds included) through the
REFERENCE's own post_process (real scikit-image / scipy under /opt/conda/bin/python3.9, cv2 stand-in) and through
oracle/postproc_ref.c.  TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference).

Run:  /opt/conda/bin/python3.9 oracle/fuzz_ref_vs_oracle.py [n_cases] [seed]"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

import oracle.cv2_standin as cv2_standin  # noqa: E402

sys.modules["cv2"] = cv2_standin
from loader.postproc import PostProcInstErodedContourMap as RefPP  # noqa: E402  (reference)

from oracle import postproc_ref as pr  # noqa: E402
from oracle import synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
bad = 0
for i in range(n_cases):
    tissue = ["Nuclei", "Nuclei", "Gland", "Lumen"][rs.randint(4)]
    H, W = int(rs.randint(17, 500)), int(rs.randint(17, 500))
    seed = int(rs.randint(1 << 30))
    noise = float(rs.choice([0.0, 0.02, 0.1, 0.3]))
    if tissue == "Nuclei":
        dens = float(rs.choice([100, 600, 2000, 6000]))
        kind = rs.randint(4)
        if kind == 0:
            m = synth.nuclei_maps(H, W, seed, dens, noise=noise)
        elif kind == 1:
            m = np.round(synth.nuclei_maps(H, W, seed, dens, noise=noise) * 4) / 4
        elif kind == 2:
            m = synth.blob_maps(H, W, seed, max(3, H * W // 6000), 6.0, 30.0, rim=2.0, sharp=float(rs.choice([0.5, 1.5])), noise=noise, border_bias=True)
        else:
            m = rs.rand(H, W, 2).astype(np.float32) * np.array([1.2, 0.4], np.float32)
        ds = 1.0
    else:
        ds = float(rs.choice([1.0, 0.5, 0.3])) if tissue == "Gland" else float(rs.choice([1.0, 0.5]))
        m = synth.blob_maps(H, W, seed, max(2, H * W // int(rs.choice([3000, 10000, 40000]))), 5.0, float(rs.choice([15, 40, 90])), rim=float(rs.choice([2.0, 4.0])),
                            sharp=1.0, noise=noise, holes=float(rs.choice([0.0, 0.3, 0.7])), border_bias=bool(rs.randint(2)))
    m = np.ascontiguousarray(m.astype(np.float32))
    ref, typ = RefPP.post_process(m, {"%s-INST" % tissue: [0, 2]}, tissue, ds_factor=ds)
    got = pr.proc(m, tissue, ds)
    if ref.shape != got.shape or not np.array_equal(np.asarray(ref).astype(np.int64), np.asarray(got).astype(np.int64)):
        bad += 1
        print("MISMATCH case %d: %s %dx%d seed %d noise %.2f ds %.1f  (%d px)" % (i, tissue, H, W, seed, noise, ds, int((np.asarray(ref) != np.asarray(got)).sum())), flush=True)
print("reference vs C oracle: %d cases, %d mismatches" % (n_cases, bad))
