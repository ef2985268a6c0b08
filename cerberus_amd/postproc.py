"""Mirror of the reference's instance post-processing (loader/postproc.py:268-407) on top of the HIP kernels.

    PostProcInstErodedContourMap.post_process(raw_map, idx_dict, tissue_mode, ds_factor=1.0) -> (inst_map, type_map)

keeps the reference's signature, assertions and return dtypes (int32 for the nuclei watershed branch, float64
otherwise) when called with numpy arrays.  The device-resident entry points used by the tile / WSI drivers are
`postproc_device` (one INST map -> int32 label map, all on the GPU) and `mask_lumen_by_gland`.

Handles are not picklable, so this runs in the main process (the reference's nr_post_proc_workers=0 path,
infer/tile.py:413-416).  There is no CPU fallback.
"""
import copy
import ctypes as C

import numpy as np
import torch

from . import _lib
from .inst_info import info_from_table  # noqa: F401  (numpy-only module: the .dat writer process imports it without torch)

_ws_cache = {}


def _workspace(device, h, w, nbytes=None):
    """The labelling workspace of an h x w map (cerb_pp_workspace_bytes: 96 B / px), or -- nbytes given -- at least that many bytes of it."""
    need = int(_lib.lib().cerb_pp_workspace_bytes(int(h), int(w))) if nbytes is None else int(nbytes)
    key = (device.index if device.index is not None else torch.cuda.current_device())
    t = _ws_cache.get(key)
    if t is None or t.numel() < need:
        # growing: the old block goes back to the allocator BEFORE the new one is asked for -- held through the assignment, a 80 GiB workspace
        # and its 88 GiB successor were both alive at the peak (a 98304^2 slide walked in sub-bands died there with 27 GiB free)
        _ws_cache.pop(key, None)
        t = None
        _ws_cache[key] = t = torch.empty(need, dtype=torch.uint8, device=device)
    return t


def postproc_device(inst, tissue_mode, ds_factor=1.0, out=None, exact_ties=True):
    """inst: CUDA float32 tensor (H,W,2) or a strided (H,W,>=2) window of a canvas (channel 0 inner, 1 contour).
    Returns (labels int32 CUDA (H,W), info) with info = {'n_inst': 0-d CUDA int32 (-1: empty nuclei map),
    'n_ambiguous': 0-d CUDA int32 (nuclei only)}.  Nothing is synchronised for nuclei.

    exact_ties (nuclei): when the floods count a region whose labels depend on the order in which skimage's heap releases
    equal-valued markers (n_ambiguous > 0), re-flood the map through the on-device replay of that heap so that the result is
    skimage's (loader/postproc.py:378) in every case.  The WSI band drivers pass False: the reference floods 4096^2 tiles there
    (infer/wsi.py:143-149), so its tie order belongs to its tiling and a slide-sized replay would buy nothing."""
    if not torch.cuda.is_available():
        raise _lib.CerberusHipError("cerberus_amd needs a ROCm GPU; there is no CPU fallback")
    L = _lib.lib()
    assert inst.is_cuda and inst.dtype == torch.float32 and inst.dim() == 3 and inst.shape[2] >= 2
    assert inst.stride(2) == 1, "inner/contour channels must be adjacent"
    h, w = int(inst.shape[0]), int(inst.shape[1])
    dev = inst.device
    labels = out if out is not None else torch.empty((h, w), dtype=torch.int32, device=dev)
    assert labels.is_contiguous() and labels.dtype == torch.int32 and tuple(labels.shape) == (h, w)
    meta = torch.zeros(2, dtype=torch.int32, device=dev)
    ws = _workspace(dev, h, w)
    stream = torch.cuda.current_stream(dev).cuda_stream
    t = tissue_mode.upper()
    with torch.cuda.device(dev):
        if t == "NUCLEI":
            _lib.check(L.cerb_postproc_nuclei(inst.data_ptr(), h, w, inst.stride(0), inst.stride(1), labels.data_ptr(), meta.data_ptr(),
                                              meta.data_ptr() + 4, 1 if exact_ties else 0, ws.data_ptr(), ws.numel(), C.c_void_p(stream)))
        elif t in ("GLAND", "LUMEN"):
            fn = L.cerb_postproc_gland if t == "GLAND" else L.cerb_postproc_lumen
            _lib.check(fn(inst.data_ptr(), h, w, inst.stride(0), inst.stride(1), C.c_float(ds_factor), labels.data_ptr(), meta.data_ptr(),
                          ws.data_ptr(), ws.numel(), C.c_void_p(stream)))
        else:
            raise AssertionError(tissue_mode)
    return labels, {"n_inst": meta[0], "n_ambiguous": meta[1]}


def mask_lumen_by_gland(lumen, gland):
    """Lumen *= (Gland > 0)   (infer/tile.py:187-191) in place on the GPU."""
    assert lumen.is_cuda and gland.is_cuda and lumen.dtype == torch.int32 and gland.dtype == torch.int32
    assert lumen.is_contiguous() and gland.is_contiguous() and lumen.shape == gland.shape
    stream = torch.cuda.current_stream(lumen.device).cuda_stream
    with torch.cuda.device(lumen.device):
        _lib.check(_lib.lib().cerb_mask_lumen_by_gland(lumen.data_ptr(), gland.data_ptr(), lumen.numel(), C.c_void_p(stream)))
    return lumen


class PostProcInstErodedContourMap(object):
    """Drop-in for the reference class of the same name (loader/postproc.py:268)."""

    last_info = None

    @classmethod
    def post_process(cls, raw_map, idx_dict, tissue_mode, ds_factor=1.0):
        assert tissue_mode.upper() in ("LUMEN", "GLAND", "NUCLEI")
        tissue_ch = f"{tissue_mode}-INST"
        idx_dict = copy.deepcopy(idx_dict)
        assert tissue_ch in list(idx_dict.keys())
        is_np = isinstance(raw_map, np.ndarray)
        dev_map = torch.from_numpy(np.ascontiguousarray(raw_map, dtype=np.float32)).cuda() if is_np else raw_map
        inst_fg = dev_map[..., idx_dict[tissue_ch][0]: idx_dict[tissue_ch][1]]
        assert inst_fg.shape[-1] == 2
        labels, info = postproc_device(inst_fg, tissue_mode, ds_factor)
        cls.last_info = info
        type_ch = tissue_mode + "-" + "TYPE"
        if type_ch in list(idx_dict.keys()):
            type_map = raw_map[..., idx_dict[type_ch][0]: idx_dict[type_ch][1]]
            type_map = np.squeeze(type_map) if is_np else torch.squeeze(type_map)
        else:
            type_map = None
        if not is_np:
            return labels, type_map
        inst_map = labels.cpu().numpy()
        # reference dtypes: int32 out of skimage.watershed, float64 zeros / canvases otherwise (postproc.py:290,331,380)
        if tissue_mode.upper() != "NUCLEI" or int(info["n_inst"].item()) < 0:
            inst_map = inst_map.astype(np.float64)
        return inst_map, type_map



def inst_table_device(inst_map, type_map=None, n_inst=None):
    """Per-instance reductions on the GPU.  inst_map: CUDA int32 (H,W); type_map: CUDA uint8 (H,W) or None.
    Returns a CUDA int64 tensor [n_inst, 16] (layout: include/cerberus_hip.h, cerb_inst_table)."""
    assert inst_map.is_cuda and inst_map.dtype == torch.int32 and inst_map.dim() == 2 and inst_map.stride(1) == 1
    if n_inst is None:
        n_inst = int(inst_map.max().item()) if inst_map.numel() else 0
    table = torch.empty((max(n_inst, 0), 16), dtype=torch.int64, device=inst_map.device)
    if n_inst <= 0:
        return table
    tp, ts = None, 0
    if type_map is not None:
        assert type_map.is_cuda and type_map.dtype == torch.uint8 and type_map.shape == inst_map.shape and type_map.stride(1) == 1
        tp, ts = type_map.data_ptr(), type_map.stride(0)
    stream = torch.cuda.current_stream(inst_map.device).cuda_stream
    with torch.cuda.device(inst_map.device):
        _lib.check(_lib.lib().cerb_inst_table(inst_map.data_ptr(), inst_map.stride(0), tp, ts, int(inst_map.shape[0]), int(inst_map.shape[1]),
                                              int(n_inst), table.data_ptr(), C.c_void_p(stream)))
    return table


def inst_contours_device(inst_map, table):
    """Outer border of every instance (cerb_inst_contour_count / _points: Suzuki-Abe border following with
    CHAIN_APPROX_SIMPLE, one GPU thread per instance).  inst_map: CUDA int32 (H,W); table: cerb_inst_table output (CUDA).
    Returns (counts int32 [n] on the host, points int32 [total, 2] (x, y) on the host, offsets int64 [n] on the host)."""
    n = int(table.shape[0])
    if n == 0:
        return np.zeros(0, np.int32), np.zeros((0, 2), np.int32), np.zeros(0, np.int64)
    L = _lib.lib()
    h, w = int(inst_map.shape[0]), int(inst_map.shape[1])
    stream = torch.cuda.current_stream(inst_map.device).cuda_stream
    counts = torch.empty(n, dtype=torch.int32, device=inst_map.device)
    # cerb_inst_contour_start: one union-find label per pixel (int32; int64 for maps of 2^31 pixels and more) -- not the 96 B / px of a labelling
    # call: a 40000^2 slide map needs 6.4 GB here, not 143 GiB
    ws = _workspace(inst_map.device, h, w, nbytes=int(L.cerb_inst_contour_start_workspace_bytes(h, w)) + 256)
    table = table.clone()  # column 7 becomes the start pixel of the border findContours lists first (several-piece instances)
    start = torch.empty(n, dtype=torch.int64, device=inst_map.device)
    with torch.cuda.device(inst_map.device):
        _lib.check(L.cerb_inst_contour_start(inst_map.data_ptr(), inst_map.stride(0), h, w, n, start.data_ptr(), ws.data_ptr(), ws.numel(), C.c_void_p(stream)))
        table[:, 7] = start
        _lib.check(L.cerb_inst_contour_count(inst_map.data_ptr(), inst_map.stride(0), h, w, n, table.data_ptr(), counts.data_ptr(), C.c_void_p(stream)))
        incl = torch.cumsum(counts.to(torch.int64), 0)
        offsets = (incl - counts).contiguous()
        total = int(incl[-1].item())
        points = torch.empty((max(total, 1), 2), dtype=torch.int32, device=inst_map.device)
        _lib.check(L.cerb_inst_contour_points(inst_map.data_ptr(), inst_map.stride(0), h, w, n, table.data_ptr(), offsets.data_ptr(), points.data_ptr(),
                                              C.c_void_p(stream)))
    return counts.cpu().numpy(), points[:total].cpu().numpy(), offsets.cpu().numpy()


def get_inst_info_dict(inst_map, type_map=None, ds_factor=1.0, flat_box=False):
    """Mirror of the reference's get_inst_info_dict (loader/postproc.py:12-98):
    dict id -> {'box': [[rmin,cmin],[rmax,cmax]], 'centroid': [x, y], 'contour': int32 (K,2) of (x, y), 'type', 'type_prob'}.

    inst_map / type_map may be CUDA tensors (label map int32, class map uint8) or numpy arrays.  The per-instance sums
    (cerb_inst_table) and the border following (cerb_inst_contour_*) run on the GPU; only the table and the compact point
    list are copied to the host.  Instances whose contour has fewer than 3 points are skipped (postproc.py:34-35).
    OpenCV is not installed in this image: contours are restated from Suzuki-Abe + OpenCV's conventions, not pinned."""
    from collections import OrderedDict

    if isinstance(inst_map, np.ndarray):
        inst_map = torch.from_numpy(np.ascontiguousarray(inst_map).astype(np.int32)).cuda()
    if type_map is not None and isinstance(type_map, np.ndarray):
        type_map = torch.from_numpy(np.ascontiguousarray(type_map).astype(np.uint8)).cuda()
    inst_map = inst_map.contiguous()
    tab_dev = inst_table_device(inst_map, None if type_map is None else type_map.contiguous())
    cnts, pts, offs = inst_contours_device(inst_map, tab_dev)
    return info_from_table(tab_dev.cpu().numpy(), cnts, pts, offs, type_map is not None, ds_factor, flat_box)
