"""Pins the C oracle of the post-processing path (oracle/postproc_ref.c) to the golden vectors produced by the
reference's own loader/postproc.py run against the real scikit-image / scipy (oracle/gen_golden_postproc.py)."""
import os

import numpy as np
import pytest

from oracle import postproc_ref as pr


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "pp_cases.npz"))
    for name in [str(n) for n in g["names"]]:
        yield name, str(g["tissue/" + name]), float(g["ds/" + name]), g["in/" + name].astype(np.float32), g["out/" + name], str(g["dtype/" + name])


def test_oracle_reproduces_reference_label_maps(golden_dir):
    n = 0
    for name, tissue, ds, m, ref, dt in _cases(golden_dir):
        got = pr.proc(m, tissue, ds)
        assert str(got.dtype) == dt, name  # int32 (nuclei watershed branch) / float64 (postproc.py:290,331,380)
        assert np.array_equal(got.astype(np.int32), ref), name
        n += 1
    assert n >= 20


def test_post_process_protocol(golden_dir):
    """post_process(raw_map, idx_dict, tissue_mode, ds_factor) -> (inst_map, type_map|None)  (postproc.py:383-407)."""
    for name, tissue, ds, m, ref, dt in _cases(golden_dir):
        if name != "nuc_generic":
            continue
        raw = np.zeros(m.shape[:2] + (9,), np.float32)
        raw[..., 4:6] = m
        raw[..., 6] = 3.0
        idx = {"Lumen-INST": [0, 2], "Gland-INST": [2, 4], "Nuclei-INST": [4, 6], "Nuclei-TYPE": [6, 7], "Gland-TYPE": [7, 8], "Patch-Class": [8, 9]}
        inst, typ = pr.PostProcInstErodedContourMap.post_process(raw, idx, "Nuclei")
        assert np.array_equal(inst, ref) and typ.shape == m.shape[:2] and (typ == 3).all()
        inst, typ = pr.PostProcInstErodedContourMap.post_process(raw, idx, "Lumen")
        assert typ is None and inst.dtype == np.float64
        with pytest.raises(AssertionError):
            pr.PostProcInstErodedContourMap.post_process(raw, idx, "Stroma")


def test_ellipse_structuring_elements():
    """Row spans of cv2.getStructuringElement(MORPH_ELLIPSE): 3x3 is the cross; 10x10 / 5x5 / 2x2 / 1x1 as derived in DESIGN.md."""
    one = np.zeros((31, 31), np.uint8)
    one[15, 15] = 1
    d3 = pr.dilate_ellipse(one, 3)[14:17, 14:17]
    assert d3.tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    d2 = pr.dilate_ellipse(one, 2)
    assert sorted(zip(*np.nonzero(d2))) == [(15, 15), (15, 16), (16, 15)]  # reflected SE [[0,1],[1,1]] about its anchor (1,1)
    assert pr.dilate_ellipse(one, 1).sum() == 1
    d10 = pr.dilate_ellipse(one, 10)
    assert d10.sum() == 1 + 7 + 9 + 5 * 10 + 9 + 7
    d5 = pr.dilate_ellipse(one, 5)
    assert d5.sum() == 1 + 5 + 5 + 5 + 1
    e = pr.erode_cross3(np.ones((4, 5), np.uint8))
    assert e.all()  # the border never wins the min (cv2 default borderValue)
