"""Generate tests/golden/pp_*.npz by running the REFERENCE's own post-processing in this container.

Run:  /opt/conda/bin/python3.9 oracle/gen_golden_postproc.py
(the only interpreter here with the real scikit-image (0.18.3; reference pins 0.19.2) and scipy (1.7.1; pins 1.7.3).
OpenCV is not installed anywhere: cv2 is the functional stand-in oracle/cv2_standin.py, so getStructuringElement /
erode / dilate are restated, not pinned.)  Never runs on the GPU box.

For every case the script (1) calls the reference loader/postproc.py PostProcInstErodedContourMap.post_process,
(2) checks the C oracle (oracle/postproc_ref.c) reproduces it bit-exactly, (3) stores input map + expected label map.
"""
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

IDX = {"X-INST": [0, 2]}


def ref_proc(m, tissue, ds=1.0):
    idx = {"%s-INST" % tissue: [0, 2]}
    inst, typ = RefPP.post_process(m, idx, tissue, ds_factor=ds)
    assert typ is None
    return inst


def quant(m, levels):
    return (np.round(m * levels) / levels).astype(np.float32)


def cases():
    c = []
    # ---- nuclei -------------------------------------------------------------------------------------------
    c.append(("nuc_generic", "Nuclei", 1.0, synth.nuclei_maps(256, 256, 10, 1500.0, noise=0.02)))
    c.append(("nuc_touching", "Nuclei", 1.0, synth.nuclei_maps(192, 192, 11, 5000.0, noise=0.03)))
    c.append(("nuc_empty", "Nuclei", 1.0, np.zeros((64, 80, 2), np.float32)))
    c.append(("nuc_plateau_ties", "Nuclei", 1.0, quant(synth.nuclei_maps(192, 192, 12, 3500.0, sharp=0.8), 4)))
    c.append(("nuc_ties8", "Nuclei", 1.0, quant(synth.nuclei_maps(160, 224, 13, 4000.0, sharp=0.6, noise=0.05), 8)))
    c.append(("nuc_border", "Nuclei", 1.0, synth.nuclei_maps(160, 160, 14, 3000.0, border_bias=True)))
    c.append(("nuc_holes", "Nuclei", 1.0, synth.blob_maps(192, 192, 15, 40, 6.0, 14.0, holes=0.7, noise=0.02)))
    c.append(("nuc_small", "Nuclei", 1.0, synth.blob_maps(128, 128, 16, 120, 0.8, 2.6, rim=0.5, sharp=2.5)))
    c.append(("nuc_fp16", "Nuclei", 1.0, synth.nuclei_maps(200, 136, 17, 3000.0, noise=0.04).astype(np.float16).astype(np.float32)))
    c.append(("nuc_ragged", "Nuclei", 1.0, synth.nuclei_maps(97, 131, 18, 3500.0, noise=0.02)))
    c.append(("nuc_saturated", "Nuclei", 1.0, quant(synth.nuclei_maps(160, 160, 19, 4500.0, sharp=6.0), 1)))
    full = np.ones((48, 48, 2), np.float32)
    full[..., 1] = 0.0
    c.append(("nuc_all_fg", "Nuclei", 1.0, full))
    # ---- gland / lumen -----------------------------------------------------------------------------------------
    g = synth.blob_maps(448, 448, 20, 16, 22.0, 48.0, noise=0.02, rim=4.0, sharp=1.0)
    c.append(("gland_generic", "Gland", 1.0, g))
    c.append(("lumen_generic", "Lumen", 1.0, synth.blob_maps(320, 320, 21, 14, 8.0, 30.0, noise=0.02, rim=3.0)))
    c.append(("gland_border", "Gland", 1.0, synth.blob_maps(384, 384, 22, 14, 20.0, 45.0, border_bias=True, rim=4.0, sharp=1.0)))
    c.append(("gland_holes", "Gland", 1.0, synth.blob_maps(400, 400, 23, 9, 30.0, 60.0, holes=1.0, rim=4.0, sharp=1.0)))
    c.append(("gland_ds05", "Gland", 0.5, synth.blob_maps(320, 320, 24, 8, 25.0, 50.0, rim=4.0, sharp=1.0)))
    c.append(("lumen_ds05", "Lumen", 0.5, synth.blob_maps(224, 224, 25, 14, 6.0, 20.0, rim=2.0)))
    c.append(("lumen_small", "Lumen", 1.0, synth.blob_maps(200, 200, 26, 30, 3.0, 9.0, rim=1.0)))
    c.append(("gland_touching", "Gland", 1.0, synth.blob_maps(416, 352, 27, 30, 20.0, 34.0, noise=0.03, rim=4.0, sharp=1.0)))
    c.append(("gland_empty", "Gland", 1.0, np.zeros((96, 96, 2), np.float32)))
    return c


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    store = {}
    names = []
    for name, tissue, ds, m in cases():
        m = np.ascontiguousarray(m, dtype=np.float32)
        f32 = name in ("nuc_generic", "gland_generic")
        if not f32:  # keep the fixture small: inputs exactly representable in fp16 (also makes ties likelier)
            m = m.astype(np.float16).astype(np.float32)
        ref = ref_proc(m.copy(), tissue, ds)
        orc = pr.proc(m, tissue, ds)
        same = orc.dtype == ref.dtype and np.array_equal(orc, ref)
        nmis = int((orc != ref).sum())
        print("%-18s %-6s ds=%.1f shape=%-10s ref dtype=%-7s n_inst=%4d fg=%.3f  oracle==reference: %s (%d px differ)"
              % (name, tissue, ds, m.shape[:2], ref.dtype, int(ref.max()), float((ref > 0).mean()), same, nmis))
        assert same, name
        names.append(name)
        store["in/" + name] = m if f32 else m.astype(np.float16)
        store["out/" + name] = ref.astype(np.int32)
        store["dtype/" + name] = str(ref.dtype)
        store["tissue/" + name] = tissue
        store["ds/" + name] = np.float32(ds)
    store["names"] = np.array(names)
    path = os.path.join(out_dir, "pp_cases.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")

    # ---- primitive pins: scipy label / fill_holes, skimage watershed on random tie-heavy inputs -------------------
    from scipy import ndimage as ndi
    from skimage.segmentation import watershed
    rs = np.random.RandomState(99)
    nfail = 0
    for t in range(300):
        H, W = rs.randint(1, 40), rs.randint(1, 40)
        mask = rs.rand(H, W) < rs.uniform(0.2, 0.9)
        lab, n = pr.label4(mask)
        lab2, n2 = ndi.label(mask)
        assert n == n2 and np.array_equal(lab, lab2)
        assert np.array_equal(pr.fill_holes(mask).astype(bool), ndi.binary_fill_holes(mask))
        levels = rs.choice([1, 2, 3, 5, 17, 1000])
        img = (np.round(rs.rand(H, W) * levels) / levels).astype(np.float32)
        mk = (rs.rand(H, W) < 0.15)
        mk_lab, _ = ndi.label(mk)
        wmask = rs.rand(H, W) < 0.85
        ws_ref = watershed(-img, mk_lab, mask=wmask)
        ws = pr.watershed(-img, mk_lab.astype(np.int32), wmask)
        if not np.array_equal(ws, ws_ref):
            nfail += 1
    print("random primitive trials: 300, watershed mismatches:", nfail)
    assert nfail == 0


if __name__ == "__main__":
    main()
