"""ORACLE (test infrastructure, never imported by the product path): CPU restatement of the tissue-mask handling of the
reference's slide driver.

  mask reading / binarisation           infer/wsi.py:533-539
  patch filter                          infer/wsi.py:559-569 -> tiatoolbox 1.3.1 SemanticSegmentor.filter_coordinates (un-vendored:
                                        restated from its published source -- a patch is kept when its OUTPUT box, scaled to the mask
                                        with np.ceil, holds at least one mask pixel; parity unpinned)
  Patch-Class tissue map                infer/wsi.py:688-716
  tissue regions                        infer/wsi.py:381-391, 724-725 (scipy.ndimage.label, get_bounding_box misc/utils.py:82-91)
  per-region gland / lumen              infer/wsi.py:730-835 (crop, mask, x0.5 cv2.resize, post_process(ds 0.5), lumen-in-gland,
                                        get_inst_info_dict(ds 0.5) shifted by the region's top-left)

cv2 is oracle/cv2_standin.py (OpenCV absent -> unpinned), post_process is oracle/postproc_ref.py (pinned against the real
reference by oracle/fuzz_ref_vs_oracle.py)."""
from collections import OrderedDict

import numpy as np
from scipy import ndimage

from . import cv2_standin as cv2
from . import postproc_ref as pr


def binarise_mask(gray):
    m = np.array(gray, dtype=np.uint8)
    m[m > 0] = 1
    return m


def filter_coordinates(mask, boxes_xyxy, slide_hw):
    """boxes: int [P, 4] (x0, y0, x1, y1) at slide resolution -> bool [P]"""
    scale = np.array([mask.shape[1] / slide_hw[1], mask.shape[0] / slide_hw[0]] * 2)
    sel = np.zeros(len(boxes_xyxy), bool)
    for i, b in enumerate(np.asarray(boxes_xyxy)):
        x0, y0, x1, y1 = np.ceil(scale * b).astype(np.int32)
        sel[i] = np.sum(mask[y0:y1, x0:x1] > 0) > 0
    return sel


def pclass_tissue_map(pclass, mask):
    p = cv2.resize(np.array(pclass, np.float32), (0, 0), fx=0.25, fy=0.25, interpolation=cv2.INTER_NEAREST)
    lo = cv2.resize(mask, (p.shape[1], p.shape[0]), interpolation=cv2.INTER_NEAREST)
    return p * lo


def tissue_regions(mask):
    """-> (label map, [[rmin, rmax, cmin, cmax], ...]) in mask coordinates; one whole-mask region when the mask is empty"""
    lab = ndimage.label(mask)[0]
    info = []
    ids = np.unique(lab).tolist()
    if len(ids) > 1:
        for rid in ids[1:]:
            m = lab == rid
            rows, cols = np.any(m, axis=1), np.any(m, axis=0)
            rmin, rmax = np.where(rows)[0][[0, -1]]
            cmin, cmax = np.where(cols)[0][[0, -1]]
            info.append([int(rmin), int(rmax) + 1, int(cmin), int(cmax) + 1])
    else:
        info.append([0, lab.shape[0], 0, lab.shape[1]])
    return lab, info


def _info_scaled(inst_map, type_map, ds):
    info = pr.inst_info_ref(inst_map, type_map)
    for d in info.values():  # loader/postproc.py:78-96
        d["box"] = np.round(d["box"] / ds).astype("int")
        d["centroid"] = np.round(d["centroid"] / ds).astype("int")
        d["contour"] = np.round(d["contour"] / ds).astype("int")
    return info


def gland_lumen_regions(canv, mask, slide_hw, type_subsample=True):
    """canv: {'Gland-INST': (H,W,2) f32, 'Lumen-INST': ..., optional 'Gland-TYPE': (H,W) uint8}.
    -> list of regions: {'topleft': [cmin, rmin], 'inst': {'Gland': int32 map, 'Lumen': int32 map}, 'info': {'Gland': {...}, 'Lumen': {...}}}
    type_subsample: the class map handed to get_inst_info_dict is the masked crop sub-sampled [::2, ::2] (the build's documented
    deviation) instead of the bilinear resize of class ids (infer/wsi.py:783-788)."""
    ratio = mask.shape[0] / slide_hw[0]
    lab, regions = tissue_regions(mask)
    out = []
    for idx, (r0, r1, c0, c1) in enumerate(regions):
        rmin, rmax = int(round(r0 / ratio)), int(round(r1 / ratio))
        cmin, cmax = int(round(c0 / ratio)), int(round(c1 / ratio))
        mask_idx = lab[r0:r1, c0:c1] == idx + 1  # an empty mask keeps nothing (0 == 1 nowhere)
        inst, tmaps = OrderedDict(), {}
        for tissue in ("Gland", "Lumen"):
            m = np.array(canv[tissue + "-INST"][rmin:rmax, cmin:cmax], np.float32)
            mi = mask_idx
            if m.shape[:2] != mi.shape:
                mi = cv2.resize(mi.astype("uint8"), (m.shape[1], m.shape[0]), interpolation=cv2.INTER_NEAREST)
            half = cv2.resize(m * mi[..., None].astype(np.float32), (0, 0), fx=0.5, fy=0.5)
            inst[tissue] = pr.proc(np.ascontiguousarray(half), tissue, 0.5).astype(np.int32)
            tm = canv.get(tissue + "-TYPE")
            if tm is not None:
                t = np.array(tm[rmin:rmax, cmin:cmax]) * mi.astype(tm.dtype)
                tmaps[tissue] = t[::2, ::2][: half.shape[0], : half.shape[1]]
        inst["Lumen"] = inst["Lumen"] * (inst["Gland"] > 0)
        info = OrderedDict()
        for tissue in ("Gland", "Lumen"):
            d = _info_scaled(inst[tissue], tmaps.get(tissue), 0.5)
            for v in d.values():  # infer/wsi.py:812-826
                v["box"] = v["box"] + np.array([cmin, rmin])
                v["contour"] = v["contour"] + np.array([cmin, rmin])
                v["centroid"] = v["centroid"] + np.array([cmin, rmin])
                b = v["box"]
                v["box"] = np.array([b[0][1], b[0][0], b[1][1], b[1][0]])
            info[tissue] = d
        out.append({"topleft": [cmin, rmin], "inst": inst, "info": info})
    return out
