"""Tissue-mask path on the GPU against the oracle (oracle/wsi_ref.py): mask components, masked x0.5 resize, Patch-Class tissue
map, per-region gland / lumen label maps (bit-exact given identical probability maps) and instance dictionaries."""
import numpy as np
import pytest
import torch
from scipy import ndimage

from cerberus_amd.tissue import TissueRegions, half_inst_region, pclass_tissue_map, postprocess_regions
from oracle import cv2_standin as cv2
from oracle import synth, wsi_ref

pytestmark = pytest.mark.gpu


def _blob_mask(h, w, seed, n=4):
    rs = np.random.RandomState(seed)
    m = np.zeros((h, w), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n):
        cy, cx, r = rs.uniform(0, h), rs.uniform(0, w), rs.uniform(0.1, 0.3) * min(h, w)
        m[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = 1
    return m


@pytest.mark.parametrize("seed", range(6))
def test_label_mask_equals_scipy_label(seed):
    rs = np.random.RandomState(seed)
    h, w = int(rs.randint(1, 300)), int(rs.randint(1, 300))
    m = (rs.rand(h, w) < rs.choice([0.0, 0.3, 0.6, 1.0])).astype(np.uint8) * int(rs.choice([1, 255]))
    if seed == 5:
        m = _blob_mask(211, 307, 9)
    reg = TissueRegions(torch.from_numpy(m).cuda())
    lab, info = wsi_ref.tissue_regions((m > 0).astype(np.uint8))
    assert np.array_equal(reg.lab.cpu().numpy(), lab) and reg.n == lab.max()
    assert reg.boxes == info


@pytest.mark.parametrize("hw,mhw", [((64, 80), None), ((67, 45), None), ((63, 81), (63, 81)), ((200, 120), (50, 30)), ((157, 203), (23, 31)), ((40, 40), (97, 53))])
def test_half_inst_region_equals_cv2_resize_of_masked_map(hw, mhw):
    rs = np.random.RandomState(hw[0])
    a = rs.rand(hw[0] + 6, hw[1] + 9, 3).astype(np.float32)
    win = a[3:3 + hw[0], 5:5 + hw[1]]  # a strided window of a wider canvas, like a region crop
    dev = torch.from_numpy(a).cuda()[3:3 + hw[0], 5:5 + hw[1]]
    if mhw is None:
        got = half_inst_region(dev)
        exp = cv2.resize(np.ascontiguousarray(win[..., :2]), (0, 0), fx=0.5, fy=0.5)
    else:
        lab = rs.randint(0, 3, mhw).astype(np.int32)
        got = half_inst_region(dev, torch.from_numpy(lab).cuda(), 2)
        mi = cv2.resize((lab == 2).astype(np.uint8), (hw[1], hw[0]), interpolation=cv2.INTER_NEAREST)
        exp = cv2.resize(np.ascontiguousarray(win[..., :2]) * mi[..., None], (0, 0), fx=0.5, fy=0.5)
    assert np.array_equal(got.cpu().numpy(), exp)


@pytest.mark.parametrize("hw,mhw", [((1000, 1200), (250, 300)), ((1001, 1203), (63, 76)), ((333, 97), None)])
def test_pclass_tissue_map_equals_oracle(hw, mhw):
    rs = np.random.RandomState(hw[1])
    p = rs.randint(0, 9, hw).astype(np.float32)
    m = None if mhw is None else (rs.rand(*mhw) < 0.5).astype(np.uint8)
    got = pclass_tissue_map(torch.from_numpy(p).cuda(), None if m is None else torch.from_numpy(m).cuda())
    exp = wsi_ref.pclass_tissue_map(p, np.ones(hw, np.uint8) if m is None else m)
    assert np.array_equal(got.cpu().numpy(), exp)


def _canvases(H, W, seed):
    g = synth.blob_maps(H, W, seed, max(4, H * W // 15000), 30.0, 70.0, rim=5.0, sharp=1.0)
    l = synth.blob_maps(H, W, seed + 1, max(6, H * W // 3000), 8.0, 20.0, rim=2.5)
    t = np.random.RandomState(seed).randint(0, 3, (H, W)).astype(np.uint8)
    return {"Gland-INST": g.astype(np.float32), "Lumen-INST": l.astype(np.float32), "Gland-TYPE": t}


def _same_info(a, b):
    assert list(a.keys()) == list(b.keys())
    for k in a:
        for f in ("box", "centroid", "contour"):
            assert np.array_equal(np.asarray(a[k][f]), np.asarray(b[k][f])), (k, f)
        if "type" in b[k]:
            assert a[k]["type"] == b[k]["type"] and abs(a[k]["type_prob"] - b[k]["type_prob"]) < 1e-6


@pytest.mark.parametrize("H,W,ratio,seed", [(700, 900, 0.25, 1), (611, 833, 1.0, 2), (900, 700, 1 / 16.0, 3), (512, 512, None, 4)])
def test_postprocess_regions_equals_oracle(H, W, ratio, seed):
    canv = _canvases(H, W, seed)
    dev = {k: torch.from_numpy(v).cuda() for k, v in canv.items()}
    if ratio is None:
        mask, regions = np.ones((H, W), np.uint8), None
    else:
        mask = _blob_mask(max(1, int(H * ratio)), max(1, int(W * ratio)), seed, n=3)
        regions = TissueRegions(torch.from_numpy(mask).cuda())
    got = postprocess_regions(dev, (H, W), regions)
    exp = wsi_ref.gland_lumen_regions(canv, mask, (H, W))
    assert len(got) == len(exp) and len(got) >= 1
    n_inst = 0
    for g, e in zip(got, exp):
        assert g["topleft"] == e["topleft"]
        for t in ("Gland", "Lumen"):
            assert np.array_equal(g["inst"][t].cpu().numpy(), e["inst"][t]), t
            _same_info(g["info"][t], e["info"][t])
            n_inst += len(e["info"][t])
    assert n_inst > 0


def test_empty_mask_keeps_nothing():
    canv = _canvases(300, 300, 7)
    dev = {k: torch.from_numpy(v).cuda() for k, v in canv.items()}
    regions = TissueRegions(torch.zeros((30, 30), dtype=torch.uint8, device="cuda"))
    assert regions.n == 0 and regions.boxes == [[0, 30, 0, 30]]
    got = postprocess_regions(dev, (300, 300), regions)
    exp = wsi_ref.gland_lumen_regions(canv, np.zeros((30, 30), np.uint8), (300, 300))
    assert len(got) == len(exp) == 1
    for t in ("Gland", "Lumen"):
        assert int(got[0]["inst"][t].max()) == 0 and exp[0]["inst"][t].max() == 0
