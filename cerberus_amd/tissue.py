"""Tissue-mask handling of the slide driver (reference infer/wsi.py:533-569, 688-835), device-resident.

  load_mask            cv2.imread + BGR2GRAY + `> 0` (infer/wsi.py:533-536), through PIL (OpenCV is not a dependency)
  select_patches       which output boxes hold tissue (infer/wsi.py:559-569 -> tiatoolbox filter_coordinates): host geometry on
                       the low-resolution mask, like the patch list itself
  TissueRegions        cerb_label_mask + cerb_inst_table on the GPU: the connected components of the mask and their bounding
                       boxes (infer/wsi.py:381-391, 724-725)
  pclass_tissue_map    cerb_pclass_tissue_map (infer/wsi.py:688-716)
  postprocess_regions  per tissue region: crop the gland / lumen probability canvases, keep the region's own mask pixels, x0.5
                       resize (one fused kernel, cerb_downsample2_inst_region), post-process at ds_factor 0.5, lumen inside
                       gland, instance dictionary shifted to slide coordinates (infer/wsi.py:730-835)

Without a mask the reference builds an all-ones mask at slide resolution -> one region = the whole slide; that case never
materialises a mask here (region_lab NULL).  There is no CPU fallback: every map stays in HBM."""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .postproc import _workspace, get_inst_info_dict, inst_table_device, mask_lumen_by_gland, postproc_device


def load_mask(path):
    """-> uint8 [mh, mw] of 0 / 1.  PIL's 'L' conversion uses the same ITU-R 601 luma as cv2.COLOR_BGR2GRAY; masks are black /
    white images, for which `> 0` cannot differ."""
    from PIL import Image

    m = np.array(Image.open(path).convert("RGB").convert("L"))
    return (m > 0).astype(np.uint8)


def select_patches(mask, out_boxes_yx, slide_hw):
    """mask: uint8 [mh, mw]; out_boxes_yx: int [P, 2, 2] ((y0, x0), (y1, x1)) output boxes at slide resolution.
    A patch runs when its output box, scaled to the mask with np.ceil, covers at least one mask pixel (tiatoolbox 1.3.1
    SemanticSegmentor.filter_coordinates; un-vendored, restated).  Summed-area table instead of one slice per patch."""
    mh, mw = mask.shape
    sat = np.zeros((mh + 1, mw + 1), np.int64)
    sat[1:, 1:] = np.cumsum(np.cumsum(mask > 0, axis=0, dtype=np.int64), axis=1)
    b = np.asarray(out_boxes_yx, np.float64)
    sy, sx = mh / slide_hw[0], mw / slide_hw[1]
    y0 = np.clip(np.ceil(b[:, 0, 0] * sy).astype(np.int64), 0, mh)
    y1 = np.clip(np.ceil(b[:, 1, 0] * sy).astype(np.int64), 0, mh)
    x0 = np.clip(np.ceil(b[:, 0, 1] * sx).astype(np.int64), 0, mw)
    x1 = np.clip(np.ceil(b[:, 1, 1] * sx).astype(np.int64), 0, mw)
    y1, x1 = np.maximum(y1, y0), np.maximum(x1, x0)
    return (sat[y1, x1] - sat[y0, x1] - sat[y1, x0] + sat[y0, x0]) > 0


class TissueRegions(object):
    """Connected components of the slide mask on the GPU.  .lab: CUDA int32 [mh, mw]; .boxes: [[rmin, rmax, cmin, cmax], ...] in
    mask coordinates, region k has label k + 1.  An empty mask gives the reference's single whole-mask region (no pixel of
    which carries its label, infer/wsi.py:389-390,745: every probability is then multiplied by 0)."""

    def __init__(self, mask_dev):
        assert mask_dev.is_cuda and mask_dev.dtype == torch.uint8 and mask_dev.dim() == 2 and mask_dev.stride(1) == 1
        mh, mw = int(mask_dev.shape[0]), int(mask_dev.shape[1])
        dev = mask_dev.device
        self.mask = mask_dev
        self.lab = torch.empty((mh, mw), dtype=torch.int32, device=dev)
        n = torch.zeros(1, dtype=torch.int32, device=dev)
        ws = _workspace(dev, mh, mw)
        st = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().cerb_label_mask(mask_dev.data_ptr(), mask_dev.stride(0), mh, mw, self.lab.data_ptr(), n.data_ptr(), ws.data_ptr(),
                                                  ws.numel(), C.c_void_p(st)))
        self.n = int(n.item())
        if self.n > 0:
            tab = inst_table_device(self.lab, None, self.n).cpu().numpy()
            self.boxes = [[int(t[3]), int(t[4]), int(t[5]), int(t[6])] for t in tab]
        else:
            self.boxes = [[0, mh, 0, mw]]


def pclass_tissue_map(pclass, mask_dev=None):
    """pclass: CUDA float32 [H, W] class canvas (row stride free) -> CUDA float32 [cvRound(H/4), cvRound(W/4)]"""
    assert pclass.is_cuda and pclass.dtype == torch.float32 and pclass.dim() == 2 and pclass.stride(1) == 1
    h, w = int(pclass.shape[0]), int(pclass.shape[1])
    out = torch.empty((int(round(h * 0.25)), int(round(w * 0.25))), dtype=torch.float32, device=pclass.device)
    mp, ms, mh, mw = None, 0, 0, 0
    if mask_dev is not None:
        assert mask_dev.is_cuda and mask_dev.dtype == torch.uint8 and mask_dev.stride(1) == 1
        mp, ms, mh, mw = mask_dev.data_ptr(), mask_dev.stride(0), int(mask_dev.shape[0]), int(mask_dev.shape[1])
    st = torch.cuda.current_stream(pclass.device).cuda_stream
    with torch.cuda.device(pclass.device):
        _lib.check(_lib.lib().cerb_pclass_tissue_map(pclass.data_ptr(), pclass.stride(0), h, w, mp, ms, mh, mw, out.data_ptr(), C.c_void_p(st)))
    return out


def half_inst_region(inst, region_lab=None, region_id=0):
    """x0.5 cv2-bilinear resize of an INST window (H, W, >=2) after keeping one region's mask pixels.  region_lab: CUDA int32
    window of the mask's label map covering the same area (any resolution) or None."""
    assert inst.is_cuda and inst.dtype == torch.float32 and inst.dim() == 3 and inst.stride(2) == 1
    L = _lib.lib()
    h, w = int(inst.shape[0]), int(inst.shape[1])
    out = torch.empty((L.cerb_half_size(h), L.cerb_half_size(w), 2), dtype=torch.float32, device=inst.device)
    lp, ls, mh, mw = None, 0, 0, 0
    if region_lab is not None:
        assert region_lab.is_cuda and region_lab.dtype == torch.int32 and region_lab.stride(1) == 1
        lp, ls, mh, mw = region_lab.data_ptr(), region_lab.stride(0), int(region_lab.shape[0]), int(region_lab.shape[1])
    st = torch.cuda.current_stream(inst.device).cuda_stream
    with torch.cuda.device(inst.device):
        _lib.check(L.cerb_downsample2_inst_region(inst.data_ptr(), inst.stride(0), inst.stride(1), h, w, lp, ls, mh, mw, int(region_id), out.data_ptr(),
                                                  C.c_void_p(st)))
    return out


def postprocess_regions(canv, slide_hw, regions=None, with_info=True):
    """Gland / lumen label maps and instance dictionaries, one tissue region at a time (infer/wsi.py:730-835).
    canv: head key -> CUDA canvas at slide resolution; regions: TissueRegions or None (no mask = the whole slide).
    -> list of {'topleft': [cmin, rmin], 'inst': {'Gland': int32 CUDA map at x0.5 of the region crop, 'Lumen': ...},
                'info': {'Gland': {id: {...}}, 'Lumen': {...}}}
    The dictionaries carry the reference's coordinates, including its box arithmetic: `inst_info["box"] += [cmin, rmin]` adds the
    x offset to the row pair and the y offset to the column pair (infer/wsi.py:739,813) -- kept, a drop-in must return the same."""
    H, W = int(slide_hw[0]), int(slide_hw[1])
    tissues = [t for t in ("Gland", "Lumen") if t + "-INST" in canv]
    out = []
    if regions is None:
        todo = [(None, 0, H, 0, W, None)]
    else:
        ratio = regions.lab.shape[0] / H
        todo = []
        for k, (r0, r1, c0, c1) in enumerate(regions.boxes):
            win = regions.lab[r0:r1, c0:c1]
            todo.append((k + 1, int(round(r0 / ratio)), int(round(r1 / ratio)), int(round(c0 / ratio)), int(round(c1 / ratio)), win))
    for rid, rmin, rmax, cmin, cmax, win in todo:
        inst, tmaps = OrderedDict(), {}
        for t in tissues:
            crop = canv[t + "-INST"][rmin:rmax, cmin:cmax]
            if crop.shape[0] < 1 or crop.shape[1] < 1:
                continue
            half = half_inst_region(crop, win, rid if rid is not None else 0)
            inst[t], _ = postproc_device(half, t, 0.5)
            tm = canv.get(t + "-TYPE")
            if tm is not None and with_info:
                sub = tm[rmin:rmax, cmin:cmax][::2, ::2][: half.shape[0], : half.shape[1]].contiguous()
                if win is not None:  # class ids outside the region's own mask pixels are 0 (infer/wsi.py:776)
                    ys = torch.clamp((torch.arange(sub.shape[0], device=sub.device, dtype=torch.float64) * 2 * (1.0 / (crop.shape[0] / win.shape[0]))).floor().long(), max=win.shape[0] - 1)
                    xs = torch.clamp((torch.arange(sub.shape[1], device=sub.device, dtype=torch.float64) * 2 * (1.0 / (crop.shape[1] / win.shape[1]))).floor().long(), max=win.shape[1] - 1)
                    sub = sub * (win[ys][:, xs] == rid).to(sub.dtype)
                tmaps[t] = sub
        if "Lumen" in inst and "Gland" in inst:
            mask_lumen_by_gland(inst["Lumen"], inst["Gland"])
        rec = {"topleft": [cmin, rmin], "inst": inst, "info": OrderedDict()}
        if with_info:
            shift = np.array([cmin, rmin])
            for t, lab in inst.items():
                d = get_inst_info_dict(lab, tmaps.get(t), 0.5)
                for v in d.values():
                    b = v["box"] + shift
                    v["box"] = np.array([b[0][1], b[0][0], b[1][1], b[1][0]])
                    v["contour"] = v["contour"] + shift
                    v["centroid"] = v["centroid"] + shift
                rec["info"][t] = d
        out.append(rec)
    return out
