"""Pins the C oracle of the post-processing path (oracle/postproc_ref.c) to the golden vectors produced by the
reference's own loader/postproc.py run against the real scikit-image / scipy (oracle/gen_golden_postproc.py)."""
import ctypes as C
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


# ---- external pins of the OpenCV pieces (OpenCV itself is absent from the image) -------------------------------------------
def _cv2_doc(golden_dir):
    import json

    return json.load(open(os.path.join(golden_dir, "cv2_documented.json")))


def test_cv2_documented_structuring_elements(golden_dir):
    from oracle import cv2_standin as cv2

    doc = _cv2_doc(golden_dir)
    for k, key in ((5, "getStructuringElement_MORPH_ELLIPSE_5x5"), (3, "getStructuringElement_MORPH_ELLIPSE_3x3")):
        want = np.array(doc[key]["value"], np.uint8)
        assert np.array_equal(cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (k, k)), want)
        # the C oracle's row spans (what the gland / lumen dilation uses) describe the same element
        j1, j2 = (C.c_int * 64)(), (C.c_int * 64)()
        pr.lib().ref_ellipse_spans(k, j1, j2)
        got = np.zeros((k, k), np.uint8)
        for i in range(k):
            got[i, j1[i]:j2[i]] = 1
        assert np.array_equal(got, want)


def test_cv2_documented_contours_and_resize(golden_dir):
    from oracle import cv2_standin as cv2

    doc = _cv2_doc(golden_dir)
    d = doc["findContours_filled_rectangle"]
    m = np.zeros(d["mask_shape"], np.uint8)
    m[d["rect_rows"][0]:d["rect_rows"][1], d["rect_cols"][0]:d["rect_cols"][1]] = 1
    cs, hier = cv2.findContours(m, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)
    assert len(cs) == 1 and cs[0].reshape(-1, 2).tolist() == d["contour_xy"] and hier.tolist() == [[[-1, -1, -1, -1]]]
    d = doc["findContours_RETR_TREE_nesting"]
    m = np.zeros((20, 20), np.uint8)
    m[1:12, 1:12] = 1
    m[3:10, 3:10] = 0
    m[5:8, 5:8] = 1
    m[15:18, 2:6] = 1
    cs, hier = cv2.findContours(m, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)
    assert hier[0].tolist() == d["hierarchy"]
    assert [c.reshape(-1, 2)[0].tolist() for c in cs] == d["first_points_xy"]
    # element [0][0] (all loader/postproc.py:29-33 reads) is the top-level border found last -- here the blob, NOT the island
    # whose start pixel is... earlier; move the island's piece below the ring's start row and it still is not element 0:
    m2 = m.copy()
    m2[15:18, 2:6] = 0
    assert cv2.findContours(m2, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)[0][0].reshape(-1, 2)[0].tolist() == [1, 1]
    assert cv2.findContours_first_piece(m2).reshape(-1, 2)[0].tolist() == [5, 5]  # the shortcut differs exactly in this case
    d = doc["resize_INTER_LINEAR_fx_0p5"]
    row = np.array([d["row"]], np.float32)
    assert cv2.resize(row, (0, 0), fx=0.5, fy=1.0, interpolation=cv2.INTER_LINEAR)[0].tolist() == d["value"]


def test_contour_shortcut_equals_full_hierarchy_on_postproc_label_maps(golden_dir):
    """The device kernels follow the border of the piece found last in the raster scan (cerb_inst_contour_start); the full
    RETR_TREE hierarchy puts a different border first only when a piece of an instance lies inside a hole of another piece of the
    SAME instance.  post_process cannot emit that (watershed regions are 4-connected; gland / lumen instances are hole-filled
    before later ids overwrite them and an overwriting instance has no holes): checked on every instance of every golden map."""
    from oracle import cv2_standin as cv2

    g = np.load(os.path.join(golden_dir, "pp_cases.npz"))
    n_inst = 0
    for name in [str(n) for n in g["names"]]:
        lab = g["out/" + name].astype(np.int32)
        for iid in np.unique(lab)[1:]:
            ys, xs = np.nonzero(lab == iid)
            crop = (lab[ys.min():ys.max() + 1, xs.min():xs.max() + 1] == iid).astype(np.uint8)
            full = cv2.findContours(crop, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)[0][0]
            assert np.array_equal(full, cv2.findContours_first_piece(crop)), (name, int(iid))
            n_inst += 1
    assert n_inst > 500


# ---- W2: the reference's tiled nuclei post-processing against the untiled labelling ------------------------------------------------
@pytest.mark.parametrize("seed,dens,hw,tile,margin", [(5, 300.0, (700, 900), 256, 32), (6, 600.0, (640, 1000), 320, 32), (7, 150.0, (1100, 1300), 512, 64)])
def test_reference_tile_scheme_keeps_a_subset_of_the_untiled_instances(seed, dens, hw, tile, margin):
    """infer/wsi.py:81-268 + 642-682 (4096^2 tiles, 64-px margins, vertical / horizontal strips, cross sections; here scaled down so that
    a small map has many seams) restated in oracle/wsi_tiles_ref.py, on structured maps:
      * every instance the reference's scheme keeps is an instance of the UNTILED labelling with the identical bounding box, none twice
        -- the scheme is an approximation of the whole-map result, which is what cerberus_amd/shard_postproc.py computes exactly
        (tests/test_postproc_gpu.py::test_sharded_postproc_equals_whole_map);
      * what it loses are instances lying wholly inside a margin zone whose box TOUCHES the edge line of the strip that should have
        re-found them (shapely's closed-interval `query` / `contains`): a few per cent of the instances at this seam density, ~0.1 % at
        the real 4096 / 64 geometry.  The band scheme keeps those."""
    from oracle import synth, wsi_tiles_ref as wt

    m = synth.nuclei_maps(hw[0], hw[1], seed, dens, noise=0.02)
    ref = wt.reference_tiled_nuclei(m, tile_shape=tile, margin=margin, patch_output_shape=16)
    whole = wt.whole_map_nuclei(m)
    assert len(ref) == len(set(ref)) and len(whole) == len(set(whole)) and len(whole) > 150
    lost = sorted(set(whole) - set(ref))
    assert not (set(ref) - set(whole)), sorted(set(ref) - set(whole))[:5]
    assert len(lost) <= 0.05 * len(whole), (len(lost), len(whole))
    for x0, y0, x1, y1 in lost:  # each lost instance sits within one margin of an internal tile boundary
        near_x = min(abs(e - k * tile) for k in range(1, hw[1] // tile + 1) for e in (x0, x1))
        near_y = min(abs(e - k * tile) for k in range(1, hw[0] // tile + 1) for e in (y0, y1))
        assert min(near_x, near_y) <= margin, (x0, y0, x1, y1)


def test_reference_tile_info_sets():
    from oracle import wsi_tiles_ref as wt

    info = wt.get_tile_info((1000, 700), [256, 256], 32, [16, 16])
    assert [len(b) for b, _ in info] == [12, 9, 8, 6]  # 4 x 3 grid; strips astride the 3 x 3 inner vertical / 4 x 2 horizontal edges; 3 x 2 crosses
    grid, flags = info[0]
    assert flags[0].tolist() == [0, 1, 0, 1] and flags[5].tolist() == [1, 1, 1, 1] and flags[11].tolist() == [1, 0, 1, 0]
    assert info[1][0][0].tolist() == [256 - 32, 0, 256 + 32, 256] and info[1][1][0].tolist() == [0, 1, 0, 0]
    assert info[3][0][0].tolist() == [256 - 64, 256 - 64, 256 + 64, 256 + 64]
    small = wt.get_tile_info((200, 100), [256, 256], 32, [16, 16])
    assert len(small) == 1 and small[0][1].tolist() == [[0, 0, 0, 0]]


def test_instance_dictionary_oracle_vs_reference_fixture(golden_dir):
    """oracle/postproc_ref.py::inst_info_ref against what the REFERENCE's get_inst_info_dict (loader/postproc.py:12-98) returned for the
    golden label maps (tests/golden/inst_info.npz, oracle/gen_golden_instinfo.py): boxes, the < 3-point skip, majority type with the
    background rule and the stable tie order, type_prob, key order, ds_factor rounding.  cv2.moments / findContours went through the
    stand-in on both sides (OpenCV is installed nowhere here): those two stay unpinned."""
    from oracle import instinfo_fixture as fx

    n = 0
    for tag, lab, typ, ds, with_type, ref in fx.cases(golden_dir):
        fx.check(pr.inst_info_ref(lab, typ if with_type else None, ds_factor=ds), ref, with_type, tag)
        n += 1
    assert n >= 20
